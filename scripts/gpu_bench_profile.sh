# usage: bash scripts/gpu_bench_profile.sh <tag>      (runs on the GPU box via gpurun)
set -x
TAG=${1:-r1}
cd /root/repo
mkdir -p gpurun_out
python bench.py --steps 32 --warmup 4 > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
tail -3 gpurun_out/bench_$TAG.err
cat gpurun_out/bench_$TAG.json
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_$TAG -o $TAG -- python3 /root/repo/bench.py --steps 16 --warmup 2 --no-cpu-baseline > /root/repo/gpurun_out/prof_$TAG.log 2>&1
tail -2 /root/repo/gpurun_out/prof_$TAG.log
find /root/repo/gpurun_out/prof_$TAG -name "*stats*" | head; 
f=$(find /root/repo/gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1); head -12 "$f"
