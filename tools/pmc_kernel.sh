# SQ counters of one kernel (name substring) inside a bench.py run:
#   bash tools/pmc_kernel.sh <kernel-substring> [bench.py args]     -> printed per-dispatch averages
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
pat=$1; shift
rm -rf $R/gpurun_out/pmc_k
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR SQ_WAVES SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_k/$tag -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2>&1 || echo fail $tag
done
PAT="$pat" python3 - <<'PY'
import csv, glob, collections, os
R=os.environ['GRAFT_REPO_ROOT']; pat=os.environ['PAT']
agg=collections.defaultdict(lambda: collections.defaultdict(float))
cnt=collections.defaultdict(int)
dur=collections.defaultdict(list)
for f in glob.glob(R+'/gpurun_out/pmc_k/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','')[:50]
        if pat in k:
            agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
            cnt[(k,r['Counter_Name'])]+=1
for f in glob.glob(R+'/gpurun_out/pmc_k/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','')[:50]
        if pat in k: dur[k].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in agg.items():
    print(k, 'median us', sorted(dur[k])[len(dur[k])//2] if dur[k] else None)
    for c,x in sorted(v.items()): print(f'   {c:32s} {x/max(cnt[(k,c)],1):.4e} per dispatch')
PY
rm -rf $R/gpurun_out/pmc_k
