cd /root/repo
export OPS=train
for v in "" _a1 _a2 _a4 _a8 _a16 _a32 _a64; do
  echo "variant [$v]"; RNVP_HIP_LIB=/root/repo/probaforms_amd/csrc/librnvp_hip$v.so python scripts/bench_kernels.py ${CFGS:-c2} 2>&1 | grep -E "^\{|rror" | cut -c1-200
done
