#!/bin/bash
# rocprofv3 --kernel-trace --stats of the sampling step (bench.py, no side legs); prints the per-kernel table.
#   tools/step_prof.sh OUTDIR [bench args...]      env (DDIMX_*) passes through
OUT=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$ROOT/bench.py" --steps 100 --no-cpu-baseline --no-train-leg --no-extra-legs --no-roofline "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"
f=$(ls "$OUT"/*/*kernel_stats.csv | head -1)
cp "$f" "$OUT/kernel_stats.csv"
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms")
for r in rows[:28]:
    n = r["Name"]
    n = n[:110]
    print(f'{float(r["TotalDurationNs"])/1e3:10.1f} us  {int(r["Calls"]):6d} calls  avg {float(r["AverageNs"])/1e3:8.2f} us  {100*float(r["TotalDurationNs"])/tot:5.1f}%  {n}')
PY
