#!/bin/bash
# Kernel timeline of the step (rocprofv3 --kernel-trace, no counters): OUTDIR/kernel_trace_{fork,nofork}.csv
out=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for kv in "$@"; do export "$kv"; done
mkdir -p $R/$out
cd /tmp && export TMPDIR=/tmp
for m in fork nofork; do
  rm -rf /tmp/pf_tr
  if [ $m = nofork ]; then export DDIMX_FORK_MASK=0; else unset DDIMX_FORK_MASK; fi
  rocprofv3 --kernel-trace --output-format csv -d /tmp/pf_tr -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-train-leg --no-extra-legs > /tmp/pf_tr.log 2>&1
  f=$(find /tmp/pf_tr -name "*kernel_trace.csv" | head -n1)
  python3 - "$f" "$R/$out/kernel_trace_$m.csv" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-900:]  # the last steps
t0 = int(rows[0]["Start_Timestamp"])
with open(sys.argv[2], "w") as o:
    o.write("start_us,dur_us,queue,stream,wg,grid,name\n")
    for r in rows:
        o.write("%.2f,%.2f,%s,%s,%s,%s,%s\n" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                r.get("Queue_Id", ""), r.get("Stream_Id", ""), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "")), r.get("Grid_Size", r.get("Grid_Size_X", "")), r["Kernel_Name"][:70].replace(",", ";")))
P
done
