# usage: bash tools/prof_conv_bf16.sh "<bench_conv_bf16 args>" [ENV=VAL ...]  -> per-kernel average durations
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
args="$1"; shift
for kv in "$@"; do export "$kv"; done
rm -rf /tmp/pcb && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pcb -o p -- python3 $R/tools/bench_conv_bf16.py $args > /tmp/pcb.out 2>&1
cat /tmp/pcb.out | grep -v Warning | tail -12
python3 - <<'PY'
import csv
for r in csv.DictReader(open('/tmp/pcb/p_kernel_stats.csv')):
    n=r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    if any(k in n for k in ('conv','wgrad','pack','guard','bias_grad')):
        print(f"{n[:72]:72s} n={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:9.1f} min_us={float(r['MinNs'])/1e3:9.1f}")
PY
