#!/bin/bash
# A/B of the FNet paths on one box: DDIMX_FNET_DENSE=0 (six launches per layer) vs 1 (fnet_dense.hip), forked and single-stream
# step, then single-stream kernel averages of both.  usage: tools/fnet_ab.sh OUTDIR
out=$1; mkdir -p $out
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
B="python bench.py --no-cpu-baseline --no-roofline --no-train-leg --no-extra-legs"
for round in 1 2; do
  for d in 0 1; do
    DDIMX_FNET_DENSE=$d $B 2>>$out/err.txt | tail -n1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('dense=$d fork   ', j['value'], j['ms_per_step'])" >> $out/ab.txt
    DDIMX_FNET_DENSE=$d DDIMX_FORK_MASK=0 $B 2>>$out/err.txt | tail -n1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('dense=$d nofork ', j['value'], j['ms_per_step'])" >> $out/ab.txt
  done
done
DDIMX_FNET_DENSE=1 $B --batch 32 --steps 200 2>>$out/err.txt | tail -n1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('dense=1 fork b32', j['value'], j['ms_per_step'])" >> $out/ab.txt
DDIMX_FNET_DENSE=1 $B --fnet-dtype f32 2>>$out/err.txt | tail -n1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('dense=1 fork mixed', j['value'], j['ms_per_step'])" >> $out/ab.txt
DDIMX_FNET_DENSE=0 $B --fnet-dtype f32 2>>$out/err.txt | tail -n1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('dense=0 fork mixed', j['value'], j['ms_per_step'])" >> $out/ab.txt
cd /tmp && export TMPDIR=/tmp
for d in 0 1; do
  rm -rf /tmp/pf_$d; DDIMX_FNET_DENSE=$d DDIMX_FORK_MASK=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_$d -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-train-leg --no-extra-legs > /tmp/pf_$d.log 2>&1
  cp $(find /tmp/pf_$d -name "*kernel_stats.csv" | head -n1) $R/$out/kernel_stats_dense$d.csv
done
cd $R; cat $out/ab.txt
