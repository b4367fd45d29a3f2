# round 6: the tile-split training kernel with the wave number in an SGPR (static VALU count of the C2 kernel 1689 -> 1506)
cd /root/repo
for NTv in 32 1024 8192; do echo "rows $NTv"; OPS=train ITERS=400 WARM_S=0.1 NT=$NTv CFGS="c2 c3" bash scripts/gpu_ab.sh "" _tsrfl | grep -v "^variant.*rep 2" ; done
RNVP_HIP_LIB=/root/repo/probaforms_amd/csrc/librnvp_hip_tsrfl.so python -m pytest tests/test_bench_sizes_gpu.py tests/test_hip_kernels.py -x -q 2>&1 | grep -E "passed|failed|rror|ERROR" | tail -3
