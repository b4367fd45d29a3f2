# round 5: phase-3 transposed reads requested ahead of the input-gradient products (RNVP_BWD_READS_FIRST, product) against
# the compiler's placement (_rl); one box
cd /root/repo
O=gpurun_out/r05; mkdir -p $O
export ITERS=40 OPS=train
{ CFGS="c2 c3 c4" bash scripts/gpu_ab.sh "" _rl
  echo "NT=16960"; NT=16960 CFGS="c2 c3" bash scripts/gpu_ab.sh "" _rl
  echo "NT=8192"; NT=8192 CFGS="c2 c3" bash scripts/gpu_ab.sh "" _rl
} > $O/reads_first_ab.txt 2>&1
timeout 600 python -m pytest tests/test_bench_sizes_gpu.py tests/test_dist_gpu.py -x -q 2>&1 | tail -5 >> $O/reads_first_ab.txt
cat $O/reads_first_ab.txt
