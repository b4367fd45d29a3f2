# round 6: the other scheduling strategies on the OTHER translation units (CVAE step, any-shape kernels, tile-split steps, resident epochs)
cd /root/repo; O=gpurun_out/r06o; mkdir -p $O
{ echo "[] product (iterative-ilp on nf4 / nf8 / bx3 / resident, default elsewhere); [_smo] whole library iterative-maxocc, [_smr] iterative-minreg, [_smc] max-memory-clause, [_smi] max-ilp; [_w1] -DRNVP_WPE=1 -mllvm -amdgpu-mfma-vgpr-form (one wave per SIMD register budget: only the CVAE line is meaningful), [_w1a] -DRNVP_WPE=1 alone"
for rep in 1 2; do for v in "" _smo _smr _smc _smi _w1 _w1a; do export RNVP_HIP_LIB=/root/repo/probaforms_amd/csrc/librnvp_hip$v.so
  echo "== variant [$v] rep $rep"
  python scripts/cvae_kernel_time.py 2>&1 | grep "^lib"
  python scripts/lmm64_time.py 128,128 2>&1 | grep -v "^$" | tail -3 | cut -c1-170
  python scripts/lmm16_time.py 2>&1 | tail -4 | cut -c1-200
  SHAPES="16,4,128,8,32;16,4,128,8,8192" python scripts/small_step_latency.py 2>&1 | tail -2
  python scripts/resident_time.py 2>&1 | tail -4 | cut -c1-150
done; done; } > $O/sched_other2.txt 2>&1; cat $O/sched_other2.txt
