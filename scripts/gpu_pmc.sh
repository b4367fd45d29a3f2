# usage: bash scripts/gpu_pmc.sh <tag> <config> <ops>   -- PMC counters for the kernels of one config
set -x
TAG=${1:-pmc}; CFG=${2:-c2}; export OPS=${3:-train}
cd /root/repo; mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ|TCC|TCP|GRBM|TA|TD)_[A-Z0-9_]+" | sort -u > /root/repo/gpurun_out/counters_list.txt
wc -l /root/repo/gpurun_out/counters_list.txt
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d /root/repo/gpurun_out/pmc_$TAG/s$i -o p -- python3 /root/repo/scripts/bench_kernels.py $CFG > /root/repo/gpurun_out/pmc_$TAG.s$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob('/root/repo/gpurun_out/pmc_$TAG/s*/*counter_collection.csv'):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].split('(')[0][-60:]
        agg[k][row['Counter_Name']] += float(row['Counter_Value']); 
        cnt[(k,row['Counter_Name'])] += 1
for k in agg:
    if 'rnvp' not in k: continue
    print(k)
    for c,v in sorted(agg[k].items()): print('   %-28s %16.0f  (per launch, %d launches)' % (c, v/cnt[(k,c)], cnt[(k,c)]))
PY
