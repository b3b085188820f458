#!/bin/bash
# Same-box A/B of two builds of the library on the headline sampling loop (B = 8) and at batch 32: libddimx_base.so vs libddimx.so.
#   usage: tools/nt_ab.sh OUTDIR
R=$(cd "$(dirname "$0")/.." && pwd)
out=$1; mkdir -p $out
for tag in base new base new base new; do
  if [ $tag = base ]; then export DDIMX_LIB=$R/ddim_audio_amd/libddimx_base.so; else export DDIMX_LIB=$R/ddim_audio_amd/libddimx.so; fi
  python3 $R/bench.py --steps 600 --no-cpu-baseline --no-roofline --no-train-leg --no-extra-legs 2>/dev/null | tail -n1 | cut -c90-150 | sed "s/^/$tag b8 /" >> $out/ab.txt || exit 1
done
for tag in base new base new; do
  if [ $tag = base ]; then export DDIMX_LIB=$R/ddim_audio_amd/libddimx_base.so; else export DDIMX_LIB=$R/ddim_audio_amd/libddimx.so; fi
  python3 $R/bench.py --batch 32 --steps 60 --no-cpu-baseline --no-roofline --no-train-leg --no-extra-legs 2>/dev/null | tail -n1 | cut -c90-150 | sed "s/^/$tag b32 /" >> $out/ab.txt || exit 1
  DDIMX_FORK_MASK=0 python3 $R/bench.py --steps 300 --no-cpu-baseline --no-roofline --no-train-leg --no-extra-legs 2>/dev/null | tail -n1 | cut -c90-150 | sed "s/^/$tag b8-one-stream /" >> $out/ab.txt || exit 1
done
cat $out/ab.txt
