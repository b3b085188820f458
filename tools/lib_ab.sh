#!/bin/bash
# A/B of two builds of libddimx on one box: per-op timings, the bench line and single-stream kernel averages.
# usage: tools/lib_ab.sh OUTDIR   (expects ddim_audio_amd/libddimx_base.so next to libddimx.so: build the commit to compare against
# -- git stash / git worktree, python -m ddim_audio_amd.build -- and copy its libddimx.so to that name; .so files are git-ignored but
# travel to the GPU box with the snapshot)
out=$1; mkdir -p $out
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for tag in base new base new; do
  if [ $tag = base ]; then export DDIMX_LIB=$R/ddim_audio_amd/libddimx_base.so; else export DDIMX_LIB=$R/ddim_audio_amd/libddimx.so; fi
  for l in 0 1 2 3 5; do python tools/conv_time.py $l 8 2 >> $out/ops_$tag.txt 2>&1; done
  python tools/downup_time.py up 1 8 >> $out/ops_$tag.txt 2>&1
  python tools/downup_time.py up 2 8 >> $out/ops_$tag.txt 2>&1
  python bench.py --no-cpu-baseline --no-roofline --no-train-leg 2>&1 | tail -n1 | cut -c100-190 >> $out/bench_$tag.txt
  DDIMX_FORK_MASK=0 python bench.py --no-cpu-baseline --no-roofline --no-train-leg 2>&1 | tail -n1 | cut -c100-190 >> $out/bench1s_$tag.txt
done
cd /tmp && export TMPDIR=/tmp
for tag in base new; do
  if [ $tag = base ]; then export DDIMX_LIB=$R/ddim_audio_amd/libddimx_base.so; else export DDIMX_LIB=$R/ddim_audio_amd/libddimx.so; fi
  rm -rf /tmp/pf_$tag; DDIMX_FORK_MASK=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_$tag -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-train-leg > /tmp/pf_$tag.log 2>&1
  cp $(find /tmp/pf_$tag -name "*kernel_stats.csv" | head -n1) $R/$out/kernel_stats_$tag.csv
done
cd $R; tail -n 20 $out/ops_*.txt $out/bench*.txt
