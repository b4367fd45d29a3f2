# A/B of library variants inside ONE gpurun call (boxes differ by a few percent): bash scripts/gpu_ab.sh "" _x
cd /root/repo
for rep in 1 2; do for v in "$@"; do
  echo "variant [$v] rep $rep"; N=${N:-1048576} OPS=${OPS:-fwd,inv,train} RNVP_HIP_LIB=/root/repo/probaforms_amd/csrc/librnvp_hip$v.so python scripts/bench_kernels.py ${CFGS:-c2 c3 c4} 2>&1 | grep -E "^\{|rror" | python3 -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print('  ', j['config'], ' '.join('%s=%.4f' % (k, v) for k, v in j.items() if k.endswith('_ms')))"
done; done
