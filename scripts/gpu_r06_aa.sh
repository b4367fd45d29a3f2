# round 6: C2 flows on the barrier-free split-bf16 kernel with 2 / 3 / 4 row tiles per wave against the f32 kernels, warm clocks
cd /root/repo; O=gpurun_out/r06aa; mkdir -p $O
{ echo "scripts/bench_kernels.py c2 (1M rows, ITERS=30, >= 50 ms warm-up), ms; PREC unset = f32 kernels (auto); PREC=bx3 on [] 3 row tiles per wave, [_d2] 2, [_d4] 4"
for rep in 1 2 3; do
  echo "f32 (auto)"; N=1048576 OPS=fwd,inv ITERS=30 python scripts/bench_kernels.py c2 2>&1 | grep "^{" | python3 -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print('  ', ' '.join('%s=%.4f' % (k, v) for k, v in j.items() if k.endswith('_ms')))"
  for v in _d2 "" _d4; do echo "bx3 [$v]"; RNVP_HIP_LIB=/root/repo/probaforms_amd/csrc/librnvp_hip$v.so PREC=bx3 N=1048576 OPS=fwd,inv ITERS=30 python scripts/bench_kernels.py c2 2>&1 | grep "^{" | python3 -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print('  ', ' '.join('%s=%.4f' % (k, v) for k, v in j.items() if k.endswith('_ms')))"; done; done; } > $O/c2_bx3_rows.txt 2>&1; cat $O/c2_bx3_rows.txt
