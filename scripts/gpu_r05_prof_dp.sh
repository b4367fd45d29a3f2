# round 5: kernel timeline of one rank's data-parallel step at 8192 rows (BENCH_FORCE_DIST=1 --global-batch 8192), C3 and C2
cd /root/repo; O=/root/repo/gpurun_out/r05; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
for wl in c3 c2; do
export BENCH_FORCE_DIST=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_dp_$wl -o dp -- python3 /root/repo/bench.py --steps 2 --warmup 1 --global-batch 8192 --workload $wl --no-cpu-baseline --no-api-level > $O/prof_dp_$wl.log 2>&1
python3 - $wl <<'PY'
import csv, sys, glob
wl = sys.argv[1]
f = glob.glob('/root/repo/gpurun_out/r05/prof_dp_%s/**/dp_kernel_trace.csv' % wl, recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'k_mfma_train' in r['Kernel_Name']]
a=idx[-40]; b=idx[-37]
t0=int(rows[a]['Start_Timestamp'])
print("== %s: three consecutive rank steps" % wl)
for r in rows[a:b+1]:
    n=r['Kernel_Name']; n=n.split('(')[0][-60:]
    print('%9.1f us  +%7.1f us  %s'%((int(r['Start_Timestamp'])-t0)/1e3,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, n))
PY
done > $O/prof_dp_timeline.txt 2>&1
cat $O/prof_dp_timeline.txt
