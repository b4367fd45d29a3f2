# round 6: CVAE step kernel compiled for one wave per SIMD + vgpr-form (product) against the two-wave register budget ([_cw2])
cd /root/repo; O=gpurun_out/r06s; mkdir -p $O
{ echo "cvae_train_step 65536 rows, us; [] CVAE_TRAIN_WPE=1 (product), [_cw2] CVAE_TRAIN_WPE=2 (round 5's budget), both with -amdgpu-mfma-vgpr-form on the translation unit"
for rep in 1 2 3 4; do for v in "" _cw2; do RNVP_HIP_LIB=/root/repo/probaforms_amd/csrc/librnvp_hip$v.so python scripts/cvae_kernel_time.py 2>&1 | grep "^lib"; done; done
for n in 32 1024 8192; do for v in "" _cw2; do N=$n RNVP_HIP_LIB=/root/repo/probaforms_amd/csrc/librnvp_hip$v.so python scripts/cvae_kernel_time.py 2>&1 | grep "^lib"; done; done; } > $O/cvae_wpe.txt 2>&1; cat $O/cvae_wpe.txt
python -m pytest tests/test_cvae_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed" | tail -1
