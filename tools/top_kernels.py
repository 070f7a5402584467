#!/usr/bin/env python3
"""usage: python tools/top_kernels.py <rocprofv3 kernel_stats.csv> [steps=4] [n=45]: per-step time of the top kernels."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n = int(sys.argv[3]) if len(sys.argv) > 3 else 45
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:n]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print(f"{float(r['TotalDurationNs']) / steps / 1e6:8.3f} ms/step {int(r['Calls']) / steps:6.1f} calls/step {float(r['AverageNs']) / 1e3:9.1f} us  {name[:120]}")
print(f"total {tot / steps / 1e6:.3f} ms/step")
