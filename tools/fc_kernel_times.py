import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, end-start, grid_x/workgroup_x, grid_y from kernels where name like '%fc_%' order by start").fetchall()
from collections import defaultdict
d = defaultdict(list)
for n, t, gx, gy in rows:
    key = (n.replace('(anonymous namespace)::','').replace('void ','')[:40], gx, gy)
    d[key].append(t / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v2 = v[len(v)//2:]
    print(f"{k[0]:42s} grid {k[1]}x{k[2]:<4d} calls {len(v):4d}  median {sorted(v2)[len(v2)//2]:8.1f} us")
