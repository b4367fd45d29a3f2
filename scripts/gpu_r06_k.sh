# round 6: iterative-ilp scheduling on the OTHER translation units: CVAE step, any-shape kernels, tile-split steps (the per-rank batches), resident epochs
cd /root/repo; O=gpurun_out/r06k; mkdir -p $O
{ echo "[] product, [_sii] whole library under -mllvm -amdgpu-sched-strategy=iterative-ilp"
for rep in 1 2; do for v in "" _sii; do export RNVP_HIP_LIB=/root/repo/probaforms_amd/csrc/librnvp_hip$v.so
  echo "== variant [$v] rep $rep"
  python scripts/cvae_kernel_time.py 2>&1 | grep "^lib"
  python scripts/lmm64_time.py 128,128 2>&1 | grep -v "^$" | tail -3
  SHAPES="16,4,128,8,32;16,4,128,8,8192;32,8,256,12,8192;64,16,128,8,4096" python scripts/small_step_latency.py 2>&1 | tail -4
  python scripts/resident_time.py 2>&1 | tail -4
done; done; } > $O/sched_other.txt 2>&1; cat $O/sched_other.txt
