# usage (on the GPU box): bash tools/final_record.sh [tag]  -> gpurun_out/<tag>_*: full -m gpu suite, smoke, every bench line of the round
set -e
export T=${1:-r03_final}
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/${T}_pytest_gpu.txt 2>&1 || { tail -20 gpurun_out/${T}_pytest_gpu.txt; exit 1; }
tail -2 gpurun_out/${T}_pytest_gpu.txt
# the default-mode per-tensor margins of this run (the child runs log elsewhere): kept next to the pass count
cp gpurun_out/parity.log gpurun_out/${T}_parity_fp32.log 2>/dev/null || true
cp gpurun_out/parity_bf16.log gpurun_out/${T}_parity_bf16.log 2>/dev/null || true
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${T}_smoke.txt 2>&1; tail -2 gpurun_out/${T}_smoke.txt
python bench.py > gpurun_out/${T}_bench_default.jsonl 2>/dev/null            # the driver's line: fp32 headline + other_configs
python bench.py --h2d --no-other-configs > gpurun_out/${T}_bench_f32_b64_h2d.jsonl 2>/dev/null
python bench.py --h2d --dtype bf16 --emb 300 > gpurun_out/${T}_bench_bf16_b64_e300_h2d.jsonl 2>/dev/null
python bench.py --review_net_only --batch 32 --steps 50 --warmup 10 > gpurun_out/${T}_bench_umpr_r_b32.jsonl 2>/dev/null
UMPR_REDUCE_AT_WORLD1=1 python bench.py --no-cpu-baseline --no-other-configs > gpurun_out/${T}_bench_rccl_world1.jsonl 2>/dev/null
UMPR_REDUCE_AT_WORLD1=1 python bench.py --dtype bf16 --emb 300 --no-cpu-baseline >> gpurun_out/${T}_bench_rccl_world1.jsonl 2>/dev/null
python tools/conv_error.py > gpurun_out/${T}_conv_error.txt 2>/dev/null
UMPR_WINO_F4=1 python tools/conv_error.py --layers 5,8 >> gpurun_out/${T}_conv_error.txt 2>/dev/null
UMPR_CONV_WINO=0 python tools/conv_error.py --layers 5,8 >> gpurun_out/${T}_conv_error.txt 2>/dev/null
python tools/fix_counts.py > gpurun_out/${T}_fixup_list_sizes.txt 2>/dev/null
python - <<'P'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/%s_bench_*.jsonl" % os.environ["T"])):
    for l in open(f):
        l=l.strip()
        if not l.startswith("{"): continue
        d=json.loads(l); print(f.split("/")[-1], d["dtype"], d["config"]["workload"][:50], round(d["value"],1), round(d["ms_per_step"],3), round(d["roofline"]["frac"],3), (d.get("cpu_baseline") or {}).get("value"))
        for k,v in (d.get("other_configs") or {}).items(): print("   ", k, round(v.get("value",0),1), round(v.get("ms_per_step",0),3))
P
