"""How much of a rocprofv3 kernel trace runs on more than one queue at a time.

usage: python tools/trace_overlap.py <..._kernel_trace.csv> [name-substring ...]

Prints the traced span, the sum of kernel durations, the time with >= 1 / >= 2 kernels in flight, the busy time per queue, and --
for every name substring given -- the average duration of the matching kernels (to compare a kernel alone with the same kernel
sharing the chip).  Used for the weight-gradient branch of the backward and the forked sampling step.
"""
import csv
import sys
from collections import defaultdict


def main():
    path, pats = sys.argv[1], sys.argv[2:]
    rows = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), r["Kernel_Name"]))
    rows.sort()
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    ev = []
    for s, e, _, _ in rows:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    depth, last, ge1, ge2 = 0, t0, 0, 0
    for t, d in ev:
        if depth >= 1:
            ge1 += t - last
        if depth >= 2:
            ge2 += t - last
        depth += d
        last = t
    per_q = defaultdict(int)
    for s, e, q, _ in rows:
        per_q[q] += e - s
    tot = sum(e - s for s, e, _, _ in rows)
    ms = 1e-6
    print(f"{len(rows)} kernels, span {(t1 - t0) * ms:.2f} ms, sum of durations {tot * ms:.2f} ms, "
          f">=1 in flight {ge1 * ms:.2f} ms, >=2 in flight {ge2 * ms:.2f} ms")
    for q, v in sorted(per_q.items(), key=lambda kv: -kv[1]):
        print(f"  queue {q}: {v * ms:.2f} ms busy")
    for p in pats:
        d = [e - s for s, e, _, n in rows if p in n]
        if d:
            print(f"  '{p}': {len(d)} launches, avg {sum(d) / len(d) * 1e-3:.1f} us, total {sum(d) * ms:.2f} ms")


if __name__ == "__main__":
    main()
