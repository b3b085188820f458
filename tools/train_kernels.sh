#!/bin/bash
# Kernel averages of the training step at B samples per GPU (rocprofv3 --kernel-trace --stats) -> OUT.csv.  usage: tools/train_kernels.sh OUT.csv [B]
out=$1; B=${2:-32}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pf_tk
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_tk -- python3 $R/tools/train_bench.py $B 1024 4 bf16 > /tmp/pf_tk.log 2>&1
tail -n 2 /tmp/pf_tk.log
mkdir -p $(dirname $R/$out); cp $(find /tmp/pf_tk -name "*kernel_stats.csv" | head -n1) $R/$out
