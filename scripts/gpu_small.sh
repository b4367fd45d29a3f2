# usage (GPU box): bash scripts/gpu_small.sh <tag>  -- small-batch training steps: us per fused step and the per-kernel split
TAG=${1:-r03}
cd /root/repo; mkdir -p gpurun_out
python scripts/small_step_latency.py 2>&1 | tee gpurun_out/${TAG}_small_steps.txt
export TMPDIR=/tmp; cd /tmp
SHAPES="16,4,128,8,32" rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/${TAG}_prof_small -o p -- python3 /root/repo/scripts/small_step_latency.py > /root/repo/gpurun_out/${TAG}_prof_small.log 2>&1
cp $(find /root/repo/gpurun_out/${TAG}_prof_small -name "*kernel_stats.csv" | head -1) /root/repo/gpurun_out/${TAG}_small_c2b32_kernel_stats.csv
head -6 /root/repo/gpurun_out/${TAG}_small_c2b32_kernel_stats.csv | cut -c1-200
