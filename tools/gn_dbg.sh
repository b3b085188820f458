#!/bin/bash
# per-kernel averages (single stream) with the launch-free GroupNorm switched off for resid (1), the convs (2), both (3)
out=$1; mkdir -p $out; R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for d in 0 1 2 3; do
  DDIMX_GN_DBG=$d DDIMX_FORK_MASK=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pg_$d -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-train-leg > /tmp/pg_$d.log 2>&1
  cp $(find /tmp/pg_$d -name "*kernel_stats.csv" | head -n1) $R/$out/kernel_stats_dbg$d.csv
  tail -n 3 /tmp/pg_$d.log | grep -o '"value": [0-9.]*' 
done
