# usage (GPU box): bash scripts/gpu_lmm_profile.sh <tag> [hidden]   -- per-kernel split of the any-shape (lmm) path
TAG=${1:-r04}; H=${2:-128,128}
cd /root/repo; mkdir -p gpurun_out
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/${TAG}_prof_lmm -o p -- python3 /root/repo/scripts/lmm_profile.py $H > /root/repo/gpurun_out/${TAG}_prof_lmm.log 2>&1
cp $(find /root/repo/gpurun_out/${TAG}_prof_lmm -name "*kernel_stats.csv" | head -1) /root/repo/gpurun_out/${TAG}_lmm_kernel_stats.csv
python3 - <<PY
import csv
for r in list(csv.DictReader(open("/root/repo/gpurun_out/${TAG}_lmm_kernel_stats.csv")))[:8]:
    n = r["Name"]; i = n.find("k_")
    print("%-50s calls %5s avg %9.1f us total %8.2f ms" % (n[i:i+50] if i >= 0 else n[:50], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
