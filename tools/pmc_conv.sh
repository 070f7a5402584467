cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_w/$tag -o p -- python3 $R/tools/bench_conv.py --n 64 --layers 5 --reps 1 --only fwd > /dev/null 2>&1 || echo fail $tag
done
python3 - <<'PY'
import csv, glob, collections, os
R=os.environ['GRAFT_REPO_ROOT']
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(R+'/gpurun_out/pmc_w/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:60]
        if 'wino' in k or 'igemm' in k:
            agg[(k, r['Grid_Size'])][r['Counter_Name']]+=float(r['Counter_Value'])
with open(R+'/gpurun_out/pmc_w/summary.txt','w') as o:
    for k,v in agg.items():
        o.write(str(k)+'\n')
        for c,x in sorted(v.items()): o.write(f'   {c:32s} {x:.4e}\n')
PY
rm -rf $R/gpurun_out/pmc_w/SQ_*
cat $R/gpurun_out/pmc_w/summary.txt
