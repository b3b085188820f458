#!/bin/bash
# per-op conv timings for several builds of the library on one box: tools/ops_ab.sh OUT lib1.so lib2.so ...
out=$1; shift; mkdir -p $(dirname $out); R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for rep in 1 2; do for lib in "$@"; do
  export DDIMX_LIB=$R/ddim_audio_amd/$lib
  for l in 0 1 2 3 4 5; do python tools/conv_time.py $l 8 2 2>&1 | sed "s#$R/ddim_audio_amd/##" | cut -c1-60 >> $out; done
  echo "$lib $(python tools/downup_time.py up 1 8 2>&1)" >> $out
  echo "$lib $(python tools/downup_time.py up 2 8 2>&1)" >> $out
  echo "$lib $(python tools/downup_time.py down 2 8 2>&1)" >> $out
done; done
cat $out
