#!/bin/bash
# Kernel averages of the single-stream step (rocprofv3 --kernel-trace --stats) -> OUT.csv.   usage: tools/step_kernels.sh OUT.csv [env...]
out=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}

for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pf_sk
DDIMX_FORK_MASK=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_sk -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-train-leg --no-extra-legs > /tmp/pf_sk.log 2>&1
mkdir -p $(dirname $R/$out); cp $(find /tmp/pf_sk -name "*kernel_stats.csv" | head -n1) $R/$out
