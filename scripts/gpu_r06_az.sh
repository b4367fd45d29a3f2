# round 6: CVAE step kernel with the row tiles' dependent MFMA chains interleaved (encoder / decoder forward GEMM1, backward g_h)
cd /root/repo
for rep in 1 2 3; do for v in "" _cil; do RNVP_HIP_LIB=/root/repo/probaforms_amd/csrc/librnvp_hip$v.so python scripts/cvae_kernel_time.py 2>&1 | tail -1; done; done
RNVP_HIP_LIB=/root/repo/probaforms_amd/csrc/librnvp_hip_cil.so python -m pytest tests/test_cvae_gpu.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -2
