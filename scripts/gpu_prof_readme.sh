cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp; cd /tmp
EPOCHS=5 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_readme -o rd -- python3 /root/repo/scripts/readme_example.py > /root/repo/gpurun_out/prof_readme.log 2>&1
tail -2 /root/repo/gpurun_out/prof_readme.log
python3 - <<PY
import csv
for r in list(csv.DictReader(open('/root/repo/gpurun_out/prof_readme/rd_kernel_stats.csv')))[:8]:
    print(r['Name'][:60].ljust(62), r['Calls'], '%.1f us'%(float(r['AverageNs'])/1e3), r['Percentage'])
PY
