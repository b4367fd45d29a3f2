# round 6: CVAE C5 step kernel -- eight waves x two row tiles per workgroup (two waves per SIMD at the same 256 rows per workgroup) against the
# shipped four waves x four row tiles; and a soak of the barrier-free flush with the release / acquire counters
cd /root/repo; O=gpurun_out/r06h; mkdir -p $O
{ echo "cvae_train_step, 65536 rows (scripts/cvae_kernel_time.py), three repetitions per variant: [] 4 waves x 4 row tiles, FT 8 (shipped); [_cv8] 8 waves x 2 row tiles, 3 tiles per flush; [_cv8f2] the same, 2 tiles per flush"
  for rep in 1 2 3; do for v in "" _cv8 _cv8f2; do RNVP_HIP_LIB=/root/repo/probaforms_amd/csrc/librnvp_hip$v.so python scripts/cvae_kernel_time.py 2>&1 | grep "^lib"; done; done; } > $O/cvae_waves_ab.txt 2>&1; cat $O/cvae_waves_ab.txt
for v in _cv8 _cv8f2; do RNVP_HIP_LIB=/root/repo/probaforms_amd/csrc/librnvp_hip$v.so python -m pytest tests/test_cvae_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed" | tail -1; done > $O/cvae_waves_parity.txt 2>&1; cat $O/cvae_waves_parity.txt
python scripts/tflush_soak.py 1000 > $O/tflush_soak.txt 2>&1; tail -12 $O/tflush_soak.txt
