# per-kernel average durations (rocprofv3 --kernel-trace --stats) of scripts/bench_kernels.py for library variants, one box:
#   CFGS=c2 OPS=train NT=65536 bash scripts/gpu_kstats.sh "" _x      -> gpurun_out/kstats/<variant>/
cd /root/repo; export TMPDIR=/tmp; OUT=/root/repo/gpurun_out/kstats; mkdir -p $OUT; cd /tmp
export ITERS=${ITERS:-30} OPS=${OPS:-train}
for v in "$@"; do
  d=$OUT/v$v; rm -rf $d
  export RNVP_HIP_LIB=/root/repo/probaforms_amd/csrc/librnvp_hip$v.so
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -o ks -- python3 /root/repo/scripts/bench_kernels.py ${CFGS:-c2} > $d.log 2>&1
  echo "variant [$v]"
  python3 - $d <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/ks_kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r['Name'].split('(')[0].replace('rnvp::', '').replace('(anonymous namespace)::', '').replace('void ', '')
    if float(r['Percentage']) > 0.5:
        print('   %-70s calls %5s avg %9.2f us  min %9.2f  max %9.2f  %5.1f %%' % (n[-70:], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3, float(r['Percentage'])))
PY
done
