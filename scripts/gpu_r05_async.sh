# round 5: barrier-free synchronisation of the net-split training kernel, one box: product (t-wave flush RNVP_NS_TFLUSH + pairwise
# exchange RNVP_NS_PAIRSYNC) / _tf (t-wave flush only) / _sync (round 4: workgroup barriers); parity of the new paths
cd /root/repo
O=gpurun_out/r05; mkdir -p $O
export ITERS=40 OPS=train
{ CFGS="c2" bash scripts/gpu_ab.sh "" _tf _sync
  echo "NT=16960"; NT=16960 CFGS="c2 c3" bash scripts/gpu_ab.sh "" _tf _sync
  echo "NT=32768"; NT=32768 CFGS="c2 c3" bash scripts/gpu_ab.sh "" _tf _sync
} > $O/ns_sync_ab.txt 2>&1
timeout 600 python -m pytest tests/test_bench_sizes_gpu.py tests/test_hip_kernels.py -x -q 2>&1 | tail -5 >> $O/ns_sync_ab.txt
cat $O/ns_sync_ab.txt
