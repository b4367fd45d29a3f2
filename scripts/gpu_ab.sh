cd /root/repo
for v in "" _r2; do for N in 65536 1048576; do
  echo "variant [$v] N=$N"; N=$N OPS=${OPS:-fwd,inv} RNVP_HIP_LIB=/root/repo/probaforms_amd/csrc/librnvp_hip$v.so python scripts/bench_kernels.py ${CFGS:-c2 c3 c4} 2>&1 | grep -E "^\{|rror" | cut -c1-260
done; done
