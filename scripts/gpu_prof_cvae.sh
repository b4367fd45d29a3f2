# rocprofv3 kernel summary of the CVAE config-5 run (scripts/cvae_c5.py) -> gpurun_out/prof_cvae
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_cvae -o cvae -- python3 /root/repo/scripts/cvae_c5.py > /root/repo/gpurun_out/prof_cvae.log 2>&1
tail -5 /root/repo/gpurun_out/prof_cvae.log | grep -v rocprofv3
python3 - <<PY
import csv
for r in list(csv.DictReader(open('/root/repo/gpurun_out/prof_cvae/cvae_kernel_stats.csv')))[:10]:
    print(r['Name'][:80].ljust(82), r['Calls'], '%.1f us'%(float(r['AverageNs'])/1e3), r['Percentage'])
PY
