# kernel-trace of a short bench run: GPU busy union per step, idle gaps, and which kernels bound the gaps
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/tg && rocprofv3 --kernel-trace --output-format csv -d /tmp/tg -o t -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/tg/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows))
# last 4 steps: find adam_kernel ends as step boundaries
adam = [e for e in ev if 'adam_kernel' in e[2]]
bounds = [adam[i][1] for i in range(1, len(adam), 2)]   # two adam launches per step
if len(bounds) >= 5:
    t0, t1 = bounds[-5], bounds[-1]
    nsteps = 4
else:
    t0, t1 = ev[0][0], ev[-1][1]; nsteps = 6
sel = [e for e in ev if e[0] >= t0 and e[1] <= t1]
busy = 0; cur_s, cur_e = sel[0][0], sel[0][1]; gaps = []
last_name = sel[0][2]
for s, e, n in sel[1:]:
    if s > cur_e:
        gaps.append((s - cur_e, last_name[:50], n[:50]))
        busy += cur_e - cur_s; cur_s, cur_e = s, e; last_name = n
    elif e > cur_e:
        cur_e = e; last_name = n
busy += cur_e - cur_s
tot = t1 - t0
print(f"steps {nsteps}: wall {tot/1e6/nsteps:.2f} ms/step, GPU busy (union) {busy/1e6/nsteps:.2f} ms/step, idle {(tot-busy)/1e6/nsteps:.2f} ms/step in {len(gaps)/nsteps:.0f} gaps/step")
gaps.sort(reverse=True)
for g in gaps[:12]: print(f"  gap {g[0]/1e3:8.1f} us after {g[1]} before {g[2]}")
import collections
agg = collections.Counter()
for g in gaps: agg[(g[1][:40], g[2][:40])] += g[0]
print("largest gap classes (us per step):")
for k, v in agg.most_common(10): print(f"  {v/1e3/nsteps:8.1f}  {k}")
PY
