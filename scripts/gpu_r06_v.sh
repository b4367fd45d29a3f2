cd /root/repo; O=gpurun_out/r06v; mkdir -p $O
{ echo "cvae_train_step 65536 rows (CVAE_TRAIN_WPE=1 + vgpr-form in all), us, 200 warm-up steps; [] default scheduler, [_ci] iterative-ilp, [_cm] max-ilp, [_co] iterative-maxocc"
for rep in 1 2 3; do for v in "" _ci _cm _co; do RNVP_HIP_LIB=/root/repo/probaforms_amd/csrc/librnvp_hip$v.so python scripts/cvae_kernel_time.py 2>&1 | grep "^lib"; done; done; } > $O/cvae_sched.txt 2>&1; cat $O/cvae_sched.txt
