# usage (GPU box): bash scripts/gpu_pmc_any.sh <tag> <kernel substring> <python script> [args...]   -- PMC counter sets (separate passes) for one kernel
TAG=$1; KM=$2; shift; shift
cd /root/repo; mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d /root/repo/gpurun_out/pmc_$TAG/s$i -o p -- python3 "$@" > /root/repo/gpurun_out/pmc_$TAG.s$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(float); cnt = collections.Counter(); durs = []
for f in glob.glob('/root/repo/gpurun_out/pmc_$TAG/s*/*counter_collection.csv'):
    seen = set()
    for row in csv.DictReader(open(f)):
        if '$KM' not in row['Kernel_Name']: continue
        agg[row['Counter_Name']] += float(row['Counter_Value']); cnt[row['Counter_Name']] += 1
        if row['Dispatch_Id'] not in seen and row.get('End_Timestamp'):
            seen.add(row['Dispatch_Id']); durs.append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3)
print('$KM: %d launches, avg %.1f us under the counters' % (len(durs), sum(durs) / max(1, len(durs))))
for c, v in sorted(agg.items()): print('   %-28s %16.0f  (per launch, %d launches)' % (c, v / cnt[c], cnt[c]))
PY
