# usage (GPU box): bash scripts/gpu_step_profile.sh <tag> ["CASES"]  -- per-kernel split of the fit-epoch step (rocprofv3 --kernel-trace --stats)
TAG=${1:-r06}
cd /root/repo; mkdir -p gpurun_out
[ -n "$2" ] && export CASES="$2"
python scripts/step_profile.py 2>&1 | tee gpurun_out/${TAG}_steps.txt
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/${TAG}_prof_steps -o p -- python3 /root/repo/scripts/step_profile.py > /root/repo/gpurun_out/${TAG}_prof_steps.log 2>&1
cp $(find /root/repo/gpurun_out/${TAG}_prof_steps -name "*kernel_stats.csv" | head -1) /root/repo/gpurun_out/${TAG}_steps_kernel_stats.csv
python3 - <<PY
import csv
for r in list(csv.DictReader(open("/root/repo/gpurun_out/${TAG}_steps_kernel_stats.csv")))[:14]:
    print("%-90s calls %6s avg %10.1f us  total %8.2f ms" % (r["Name"].split("(")[0][-90:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
