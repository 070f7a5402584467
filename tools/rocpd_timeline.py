#!/usr/bin/env python3
"""usage: python tools/rocpd_timeline.py <rocprofv3 *_results.db> [anchor-kernel-substring] [steps]
Per-kernel totals and the time-ordered kernel sequence of the LAST step (between the last two launches of the anchor kernel) of a
`rocprofv3 --kernel-trace` run whose output is the rocpd SQLite database (the default format of this ROCm)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
anchor = sys.argv[2] if len(sys.argv) > 2 else "coattn_scores"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 110
clean = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "")
rows = db.execute("select name, count(*), sum(end - start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
for name, calls, ns in rows[:40]:
    print(f"{ns / 1e3 / steps:8.2f} us/step {calls / steps:6.2f} calls/step {ns / calls / 1e3:8.1f} us  {clean(name)[:100]}")
print(f"total {tot / 1e3 / steps:.1f} us/step over {steps} steps\n")
seq = db.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x, lds_size, vgpr_count from kernels order by start").fetchall()
idx = [i for i, r in enumerate(seq) if anchor in r[0]]
if len(idx) >= 2:
    a, b = idx[-2], idx[-1]
    t0 = seq[a][1]
    for r in seq[a:b]:
        print(f"{(r[1] - t0) / 1e3:8.1f} +{(r[2] - r[1]) / 1e3:6.1f} us  grid {r[3] // max(r[6], 1)}x{r[4]}x{r[5]} wg{r[6]} lds{r[7]} v{r[8]}  {clean(r[0])[:80]}")
    print(f"step: {(seq[b][1] - t0) / 1e3:.1f} us from anchor to anchor")
