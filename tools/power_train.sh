#!/bin/bash
# Board power / shader clock while the training step loops (tools/train_bench.py, B = 32, T = 1024, bf16, replayed from a hipGraph):
# is the training step at the board's power cap?  If it is, time = energy / cap and overlapping kernels on two streams is zero sum.
#   tools/power_train.sh [env...]        e.g. tools/power_train.sh DDIMX_WGRAD_SIDE=0
R=$(cd "$(dirname "$0")/.." && pwd)
for kv in "$@"; do export "$kv"; done
python3 $R/tools/train_bench.py 32 1024 150 bf16 graph > /tmp/power_train.json 2>/dev/null &
pid=$!
sleep 14   # import + warm-up + the eager leg's start
for i in $(seq 1 14); do
  /opt/rocm/bin/rocm-smi --showpower --showclocks -d 0 2>/dev/null | grep -E "Power|sclk" | sed 's/.*: //' | tr '\n' ' '; echo
  sleep 0.6
done
wait $pid
tail -n1 /tmp/power_train.json | cut -c1-120
