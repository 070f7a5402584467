# usage: bash tools/prof_bench.sh <tag> [bench.py args]  -> gpurun_out/<tag>_kernel_stats.csv + printed top kernels
# rocprofv3 kernel trace of bench.py (3 timed + 1 warm-up step = 4 steps in the trace)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag="$1"; shift
rm -rf /tmp/pb_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb_$tag -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $R/gpurun_out/${tag}_bench.jsonl 2> /tmp/pb_$tag.err || { tail -5 /tmp/pb_$tag.err; exit 1; }
cp /tmp/pb_$tag/p_kernel_stats.csv $R/gpurun_out/${tag}_kernel_stats.csv
python3 - /tmp/pb_$tag/p_kernel_trace.csv <<'PY'
import csv, sys, collections
# longest individual dispatches of the generic kernels (which GEMM / reduce calls carry the time)
rows = list(csv.DictReader(open(sys.argv[1])))
sel = [r for r in rows if any(k in r['Kernel_Name'] for k in ('gemm_f32', 'splitk_reduce', 'colsum', 'wgrad_reduce'))]
sel.sort(key=lambda r: int(r['End_Timestamp']) - int(r['Start_Timestamp']), reverse=True)
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
