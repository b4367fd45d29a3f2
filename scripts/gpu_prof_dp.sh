# kernel timeline of the data-parallel bench step on one rank (BENCH_FORCE_DIST=1)
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp; cd /tmp
BENCH_FORCE_DIST=1 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_dp -o dp -- python3 /root/repo/bench.py --steps 16 --warmup 2 --no-cpu-baseline > /root/repo/gpurun_out/prof_dp.log 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open('/root/repo/gpurun_out/prof_dp/dp_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last 3 steps: find train kernels
idx=[i for i,r in enumerate(rows) if 'k_mfma_train' in r['Kernel_Name']]
a=idx[-3]; b=idx[-1]
t0=int(rows[a]['Start_Timestamp'])
for r in rows[a:b+1]:
    n=r['Kernel_Name']; n=n.split('(')[0][-46:]
    print('%9.1f us  +%7.1f us  q%s  %s'%((int(r['Start_Timestamp'])-t0)/1e3,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, r.get('Queue_Id','?'), n))
PY
