# usage: bash tools/prof_conv.sh "<bench_conv args>" [ENV=VAL ...]   -> per-kernel average durations (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
args="$1"; shift
for kv in "$@"; do export "$kv"; done
rm -rf /tmp/pc && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc -o p -- python3 $R/tools/bench_conv.py $args > /dev/null 2>&1
python3 - <<'PY'
import csv
for r in csv.DictReader(open('/tmp/pc/p_kernel_stats.csv')):
    n=r['Name']
    if any(k in n for k in ('wino','igemm','wgrad','pack')):
        print(f"{n[:72]:72s} n={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:9.1f} min_us={float(r['MinNs'])/1e3:9.1f}")
PY
