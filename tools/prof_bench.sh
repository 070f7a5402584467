# usage: bash tools/prof_bench.sh <tag> [bench.py args]  -> gpurun_out/<tag>_kernel_stats.csv + printed top kernels
# rocprofv3 kernel trace of bench.py (3 timed + 1 warm-up step = 4 steps in the trace)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag="$1"; shift
rm -rf /tmp/pb_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb_$tag -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $R/gpurun_out/${tag}_bench.jsonl 2> /tmp/pb_$tag.err || { tail -5 /tmp/pb_$tag.err; exit 1; }
cp /tmp/pb_$tag/p_kernel_stats.csv $R/gpurun_out/${tag}_kernel_stats.csv
PB_TIMELINE=$R/gpurun_out/${tag}_timeline.txt python3 - /tmp/pb_$tag/p_kernel_trace.csv <<'PY'
import csv, sys, collections
# longest individual dispatches of the generic kernels (which GEMM / reduce calls carry the time)
rows = list(csv.DictReader(open(sys.argv[1])))
sel = [r for r in rows if any(k in r['Kernel_Name'] for k in ('gemm_f32', 'splitk_reduce', 'colsum', 'wgrad_reduce', 'fc_'))]
sel.sort(key=lambda r: int(r['End_Timestamp']) - int(r['Start_Timestamp']), reverse=True)
# per-queue (stream) busy time: which stream is the critical path
hdr = rows[0].keys()
qk = 'Queue_Id' if 'Queue_Id' in hdr else ('Stream_Id' if 'Stream_Id' in hdr else None)
if qk:
    t0 = min(int(r['Start_Timestamp']) for r in rows); t1 = max(int(r['End_Timestamp']) for r in rows)
    half = (t0 + t1) // 2                      # second half of the run = steady-state steps
    per = collections.defaultdict(lambda: [0, 0, None, None])
    for r in rows:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        if s < half: continue
        q = per[r[qk]]
        q[0] += e - s; q[1] += 1
        q[2] = s if q[2] is None else min(q[2], s); q[3] = e if q[3] is None else max(q[3], e)
    print(f"per-{qk} busy time in the second half of the trace ({(t1 - half)/1e6:.2f} ms):")
    for k, v in sorted(per.items(), key=lambda kv: -kv[1][0]):
        print(f"  {qk} {k}: busy {v[0]/1e6:7.2f} ms in {v[1]:5d} dispatches, span {(v[3]-v[2])/1e6:7.2f} ms")
    # top kernels of the second-busiest queue (the text stream in the full model)
    order = sorted(per.items(), key=lambda kv: -kv[1][0])
    if len(order) > 1:
        q2 = order[1][0]
        agg = collections.defaultdict(lambda: [0, 0])
        for r in rows:
            if r[qk] == q2 and int(r['Start_Timestamp']) >= half:
                k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:70]
                agg[k][0] += int(r['End_Timestamp']) - int(r['Start_Timestamp']); agg[k][1] += 1
        print(f"top kernels on {qk} {q2} (per step = /4):")
        for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:16]:
            print(f"   {k:70s} n={v[1]:4d} avg_us={v[0]/v[1]/1e3:8.1f} per_step_ms={v[0]/4e6:6.3f}")
    # idle gaps > 40 us on the busiest queue inside the last step of the trace (between which kernels does it wait?)
    mainq = max(per.items(), key=lambda kv: kv[1][0])[0]
    mq = sorted((r for r in rows if r[qk] == mainq), key=lambda r: int(r['Start_Timestamp']))
    adam = [i for i, r in enumerate(mq) if 'adam_kernel' in r['Kernel_Name']]
    if len(adam) >= 2:
        lo = max(i for i in adam if i < adam[-1] - 20) if any(i < adam[-1] - 20 for i in adam) else 0
        seg = mq[lo:adam[-1] + 1]
        print(f"queue {mainq}: last step = {len(seg)} dispatches, {(int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp']))/1e6:.2f} ms; gaps > 40 us:")
        sh = lambda r: r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:44]
        tot = 0
        for a, b in zip(seg, seg[1:]):
            gap = int(b['Start_Timestamp']) - int(a['End_Timestamp'])
            if gap > 0: tot += gap
            if gap > 40000:
                print(f"   {gap/1e3:8.1f} us  after {sh(a)}  before {sh(b)}")
        print(f"   total idle between kernels on that queue: {tot/1e6:.2f} ms")
# timeline of the last step over all queues: start (us from the step's first dispatch), duration, queue, kernel
import os
if qk and len(adam) >= 2:
    t_lo, t_hi = int(seg[0]['Start_Timestamp']) - 2000000, int(seg[-1]['End_Timestamp'])
    prev_adam_end = int(mq[lo]['End_Timestamp']) if lo else t_lo
    tl = sorted((r for r in rows if prev_adam_end <= int(r['Start_Timestamp']) <= t_hi), key=lambda r: int(r['Start_Timestamp']))
    with open(os.environ.get('PB_TIMELINE', '/tmp/timeline.txt'), 'w') as f:
        for r in tl:
            s0, e0 = int(r['Start_Timestamp']), int(r['End_Timestamp'])
            n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:80]
            f.write(f"{(s0 - prev_adam_end)/1e3:9.1f} {(e0 - s0)/1e3:8.1f} q{r[qk]} {n}\n")
print("longest generic-kernel dispatches:")
for r in sel[:24]:
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:60]
    print(f"  {n:60s} {(int(r['End_Timestamp']) - int(r['Start_Timestamp']))/1e3:8.1f} us grid={r.get('Grid_Size_X','?')}x{r.get('Grid_Size_Y','?')}x{r.get('Grid_Size_Z','?')} wg={r.get('Workgroup_Size_X','?')}")
PY
python3 - "$R/gpurun_out/${tag}_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"total kernel time over 4 steps: {tot/1e6:.2f} ms ({tot/4e6:.2f} ms/step)")
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:28]:
    n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    print(f"{n[:86]:86s} n={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f} per_step_ms={float(r['TotalDurationNs'])/4e6:7.3f} {float(r['Percentage']):5.1f}%")
PY
