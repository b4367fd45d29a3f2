cd /root/repo; O=gpurun_out/r06t; mkdir -p $O
{ echo "cvae_train_step 65536 rows, us; [] product, [_u2] forward hidden-tile loops unrolled by 2, [_u4] by 4"
for rep in 1 2 3; do for v in "" _u2 _u4; do RNVP_HIP_LIB=/root/repo/probaforms_amd/csrc/librnvp_hip$v.so python scripts/cvae_kernel_time.py 2>&1 | grep "^lib"; done; done; } > $O/cvae_unroll.txt 2>&1; cat $O/cvae_unroll.txt
