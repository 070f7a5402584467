# HBM-side traffic of the conv kernels (rocprofv3 --pmc, FETCH_SIZE and WRITE_SIZE in separate passes as the guide
# prescribes), summed per kernel family over one bench run of 2 steps.  Writes gpurun_out/pmc_traffic.json.
# usage: bash tools/pmc_traffic.sh [extra bench.py args, e.g. --eval: forward only, so that every conv dispatch in the
# trace is a forward launch (forward, dgrad and the wgrad transforms share kernel names)]
# `--eval` alone measures the INFERENCE forward (4x4 Winograd tile under no_grad); UMPR_WINO_F4=0 with --eval measures the
# algorithm of the training forward (2x2 tile).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
EXTRA="$@"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pt_$c
  rocprofv3 --pmc $c --output-format csv -d /tmp/pt_$c -o p -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline $EXTRA > /tmp/pt_$c.out 2> /tmp/pt_$c.err || { echo "pass $c failed"; tail -5 /tmp/pt_$c.err; exit 1; }
done
python3 - <<'PY'
import csv, glob, json, os, collections
R = os.environ['GRAFT_REPO_ROOT']
fam = lambda n: ('b16_conv_family' if any(k in n for k in ('conv_bf16_kernel', 'conv1_bf16_fwd'))
                 else 'igemm_family' if any(k in n for k in ('conv3x3_igemm', 'wino_', 'wino4_', 'pack_weights_kernel', 'conv3x3_fwd', 'conv3x3_dgrad'))
                 else 'wgrad_family' if ('wgrad' in n) else None)
out = collections.defaultdict(lambda: collections.defaultdict(float))
per_kernel = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(int)
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    for f in glob.glob(f'/tmp/pt_{c}/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            n = r['Kernel_Name']
            short = n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
            if r['Counter_Name'] != c:
                continue
            per_kernel[short][c] += float(r['Counter_Value'])
            if c == 'FETCH_SIZE':
                calls[short] += 1
            if fam(n):
                out[fam(n)][c] += float(r['Counter_Value'])
res = {'steps': 2, 'unit': 'KB (rocprofv3 FETCH_SIZE / WRITE_SIZE raw sums over 2 steps)', 'families': out,
       'kernels': {k: dict(v, dispatches=calls[k]) for k, v in per_kernel.items() if any(s in k for s in ('conv', 'wino', 'pack', 'pool', 'cb8'))}}
json.dump(res, open(R + '/gpurun_out/pmc_traffic.json', 'w'), indent=1)
print(json.dumps(res['families'], indent=1))
PY
