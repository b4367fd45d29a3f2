# round 5: the tile-split training kernel keeps its hidden activations (RNVP_TS_SAVE_H, product) against the recompute (_nts); one box
cd /root/repo
O=gpurun_out/r05; mkdir -p $O
export ITERS=60 OPS=train
{ for nt in 32 1024 4096 8192; do echo "NT=$nt"; NT=$nt CFGS="c2 c3 c4" bash scripts/gpu_ab.sh "" _nts; done
} > $O/ts_saveh_ab.txt 2>&1
timeout 900 python -m pytest tests/test_bench_sizes_gpu.py tests/test_hip_kernels.py tests/test_api_gpu.py tests/test_autograd_gpu.py -x -q 2>&1 | tail -5 >> $O/ts_saveh_ab.txt
cat $O/ts_saveh_ab.txt
