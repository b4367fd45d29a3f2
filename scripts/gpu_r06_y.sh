cd /root/repo; O=gpurun_out/r06y; mkdir -p $O
{ echo "cvae_train_step 65536 rows, us, 200 warm-up + 200 timed steps; [] 4 waves x 4 row tiles (product), [_cv8] 8 waves x 2 row tiles FT 3 with the one-wave register budget, [_cv8w2] the same with the two-wave budget"
for rep in 1 2 3; do for v in "" _cv8 _cv8w2; do RNVP_HIP_LIB=/root/repo/probaforms_amd/csrc/librnvp_hip$v.so python scripts/cvae_kernel_time.py 2>&1 | grep "^lib"; done; done; } > $O/cvae_waves_warm.txt 2>&1; cat $O/cvae_waves_warm.txt
