cd /root/repo; export TMPDIR=/tmp; O=/root/repo/gpurun_out/r05/kst; rm -rf $O; mkdir -p $O; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o p -- python3 /root/repo/bench.py --no-cpu-baseline --no-api-level > $O.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('/root/repo/gpurun_out/r05/kst/**/p_kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r['Name']
    if 'k_mfma_train' in n or 'k_train_finish' in n: print(n[n.find('k_'):n.find('>')+1][:44], r['Calls'], round(float(r['AverageNs'])/1e3, 2))
PY
