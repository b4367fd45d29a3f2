# round 6: -mllvm -amdgpu-mfma-vgpr-form on the whole library ([_vf]); [_w1] adds -DRNVP_WPE=1 (register budget of one wave per SIMD where the kernels carry the attribute)
cd /root/repo; O=gpurun_out/r06p; mkdir -p $O
{ echo "[] product; [_vf] whole library with -mllvm -amdgpu-mfma-vgpr-form; [_w1] the same + -DRNVP_WPE=1"
  NT=65536 N=1048576 OPS=train,fwd,inv CFGS="c2 c3 c4" ITERS=10 bash scripts/gpu_ab.sh "" _vf _w1
for rep in 1 2; do for v in "" _vf _w1; do export RNVP_HIP_LIB=/root/repo/probaforms_amd/csrc/librnvp_hip$v.so
  echo "== variant [$v] rep $rep"
  python scripts/cvae_kernel_time.py 2>&1 | grep "^lib"
  python scripts/lmm64_time.py 128,128 2>&1 | grep -v "^$" | tail -3 | cut -c1-170
  python scripts/lmm16_time.py 2>&1 | tail -4 | cut -c1-200
  SHAPES="2,1,10,8,32;16,4,128,8,32;16,4,128,8,8192;32,8,256,12,8192" python scripts/small_step_latency.py 2>&1 | tail -4
  python scripts/resident_time.py 2>&1 | tail -4 | cut -c1-150
  python scripts/cvae_resident_time.py 2>&1 | tail -3 | cut -c1-170
done; done; } > $O/vgpr_form.txt 2>&1; cat $O/vgpr_form.txt
