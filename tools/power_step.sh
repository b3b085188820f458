#!/bin/bash
# Board power / shader clock while the sampling loop runs (bench.py, 3000 steps): is the whole step at the board's power cap?
#   tools/power_step.sh [env...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for kv in "$@"; do export "$kv"; done
python3 $R/bench.py --steps 3000 --warmup 20 --no-cpu-baseline --no-roofline --no-train-leg --no-extra-legs > /tmp/power_step.json 2>/dev/null &
pid=$!
sleep 9
for i in 1 2 3 4 5 6; do
  /opt/rocm/bin/rocm-smi --showpower --showclocks -d 0 2>/dev/null | grep -E "Power|sclk" | sed 's/.*: //' | tr '\n' ' '; echo
  sleep 0.7
done
wait $pid
tail -n1 /tmp/power_step.json | cut -c100-190
