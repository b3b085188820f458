#!/bin/bash
# Kernel traces of the training step with the weight gradients on the backward's own stream (0) and on the second stream (1: behind each block's last data-gradient conv, 2: as soon as `du` exists):
# tools/trace_overlap.py then says how much really overlaps and what the sharing costs each kernel.   usage: wgside_trace.sh <outdir>
R=$(cd "$(dirname "$0")/.." && pwd)
out=$1; mkdir -p $R/$out
cd /tmp && export TMPDIR=/tmp
for v in 0 2 1; do
  rm -rf /tmp/pws$v
  DDIMX_WGRAD_SIDE=$v rocprofv3 --kernel-trace --output-format csv -d /tmp/pws$v -- python3 $R/tools/train_bench.py 32 1024 3 bf16 > /tmp/pws$v.log 2>&1 || exit 1
  f=$(find /tmp/pws$v -name "*kernel_trace.csv" | head -n1)
  python3 $R/tools/trace_overlap.py $f wgrad_mfma gn_bwd_apply gn_bwd_stats conv_mfma edge_wgrad > $R/$out/overlap_side$v.txt
  tail -n 1 /tmp/pws$v.log >> $R/$out/overlap_side$v.txt
done
