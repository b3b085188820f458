#!/bin/bash
# A/B of library knobs at the step level: tools/step_ab.sh "ENV=VAL ENV=VAL" "ENV=VAL" ...   ("-" = defaults); 3 interleaved rounds
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for r in 1 2 3; do
  for s in "$@"; do
    v=$( ( [ "$s" != "-" ] && export $s; python3 "$ROOT/bench.py" --steps 300 --no-cpu-baseline --no-train-leg --no-extra-legs --no-roofline 2>/dev/null ) | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f' % d['value'])")
    echo "round $r [$s] $v sample-fwd/s"
  done
done
