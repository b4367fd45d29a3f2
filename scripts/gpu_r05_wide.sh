# round 5: the barrier-free flush for the eight-wave wide form (RNVP_WIDE_TFLUSH, product) against its two-barrier flush (_nw); one box
cd /root/repo
O=gpurun_out/r05; mkdir -p $O
export ITERS=40 OPS=train
{ CFGS="c3" bash scripts/gpu_ab.sh "" _nw
  echo "NT=40000"; NT=40000 CFGS="c3" bash scripts/gpu_ab.sh "" _nw
  echo "NT=131072"; NT=131072 CFGS="c3" bash scripts/gpu_ab.sh "" _nw
  echo "NT=262144"; NT=262144 CFGS="c2 c3" bash scripts/gpu_ab.sh "" _nw
  echo "c2"; CFGS="c2" bash scripts/gpu_ab.sh "" _nw
} > $O/wide_tflush_ab.txt 2>&1
timeout 600 python -m pytest tests/test_bench_sizes_gpu.py tests/test_hip_kernels.py -x -q 2>&1 | tail -5 >> $O/wide_tflush_ab.txt
cat $O/wide_tflush_ab.txt
