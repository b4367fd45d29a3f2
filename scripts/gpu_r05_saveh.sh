# round 5: hidden activations saved by the forward phase and read back by the backward (RNVP_SAVE_H, product) against the
# recompute (_nosh); net-split launches of C2; one box
cd /root/repo
O=gpurun_out/r05; mkdir -p $O
export ITERS=40 OPS=train
{ CFGS="c2" bash scripts/gpu_ab.sh "" _nosh
  echo "NT=16960"; NT=16960 CFGS="c2" bash scripts/gpu_ab.sh "" _nosh
  echo "NT=32768"; NT=32768 CFGS="c2" bash scripts/gpu_ab.sh "" _nosh
} > $O/saveh_ab.txt 2>&1
timeout 600 python -m pytest tests/test_bench_sizes_gpu.py tests/test_hip_kernels.py -x -q 2>&1 | tail -5 >> $O/saveh_ab.txt
cat $O/saveh_ab.txt
