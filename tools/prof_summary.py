"""Summarise a rocprofv3 *_kernel_stats.csv: per-step time per kernel (test/measurement tooling)."""
import csv
import re
import sys

path, steps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(csv.DictReader(open(path)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6:.2f} ms = {tot / 1e6 / steps:.3f} ms/step")
for r in rows[:40]:
    n = r["Name"].replace("ddimx::", "")
    m = re.search(r"ConvCfgI(DF16b|f)Li(\d+)ELi(\d+)ELi(\d+)ELi(\d)ELi(\d+)ELi(\d+)", n)
    if m:
        n = f"conv_mfma<{'bf16' if m.group(1) == 'DF16b' else 'f32'},cin={m.group(2)},nout={m.group(3)},nb={m.group(4)},mode={m.group(5)},tile={m.group(6)}x{m.group(7)}>"
    print(f"{float(r['TotalDurationNs']) / 1e6 / steps:8.3f} ms/step {int(r['Calls']) / steps:7.1f} calls/step avg {float(r['AverageNs']) / 1e3:8.1f} us "
          f"{float(r['Percentage']):5.1f}%  {n[:100]}")
