# SQ counters of the bf16 conv kernels for one layer: bash tools/pmc_conv_bf16.sh <layer> <fwd|dgrad|wgrad> [ENV=VAL ...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
layer=$1; what=$2; shift; shift
for kv in "$@"; do export "$kv"; done
rm -rf $R/gpurun_out/pmc_b
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_b/$tag -o p -- python3 $R/tools/bench_conv_bf16.py --n 64 --layers $layer --reps 1 --only $what > /dev/null 2>&1 || echo fail $tag
done
python3 - <<'PY'
import csv, glob, collections, os
R=os.environ['GRAFT_REPO_ROOT']
agg=collections.defaultdict(lambda: collections.defaultdict(float))
cnt=collections.defaultdict(int)
dur=collections.defaultdict(list)
for f in glob.glob(R+'/gpurun_out/pmc_b/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','')[:50]
        if 'conv_bf16' in k or 'wgrad_bf16' in k or 'conv1_bf16' in k:
            agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
            if r['Counter_Name'] in ('SQ_WAVE_CYCLES','GRBM_GUI_ACTIVE'): cnt[(k,r['Counter_Name'])]+=1
for f in glob.glob(R+'/gpurun_out/pmc_b/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','')[:50]
        if 'conv_bf16' in k or 'wgrad_bf16' in k or 'conv1_bf16' in k:
            dur[k].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in agg.items():
    n=max(cnt[(k,'SQ_WAVE_CYCLES')],1)
    print(k, 'dispatches', n, 'median us', sorted(dur[k])[len(dur[k])//2] if dur[k] else None)
    for c,x in sorted(v.items()): print(f'   {c:32s} {x/n:.4e} per dispatch')
    if 'GRBM_GUI_ACTIVE' in v and dur[k]:
        print(f"   clock ~ {v['GRBM_GUI_ACTIVE']/max(cnt[(k,'GRBM_GUI_ACTIVE')],1)/8/(sorted(dur[k])[len(dur[k])//2]*1e-6)/1e9:.2f} GHz")
PY
rm -rf $R/gpurun_out/pmc_b
