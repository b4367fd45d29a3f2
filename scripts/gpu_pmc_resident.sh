# usage (GPU box): bash scripts/gpu_pmc_resident.sh  -- PMC counters of one resident epoch (32 steps) of the default flow at batch 32
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp; cd /tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d /root/repo/gpurun_out/pmc_resident/s$i -o p -- python3 /root/repo/scripts/resident_stamp.py > /root/repo/gpurun_out/pmc_resident.s$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob('/root/repo/gpurun_out/pmc_resident/s*/*counter_collection.csv'):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].split('(')[0][-60:]
        agg[k][row['Counter_Name']] += float(row['Counter_Value']); cnt[(k,row['Counter_Name'])] += 1
for k in agg:
    if 'resident' not in k: continue
    print(k)
    for c,v in sorted(agg[k].items()): print('   %-28s %16.0f  (per launch of 32 steps, %d launches)' % (c, v/cnt[(k,c)], cnt[(k,c)]))
PY
