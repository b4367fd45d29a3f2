# round 6: the C2 kernels under warm clocks (scripts/bench_kernels.py now warms up for >= 50 ms): compiler knobs once more, and the split-bf16 flows
cd /root/repo; O=gpurun_out/r06z; mkdir -p $O
{ echo "scripts/bench_kernels.py c2, warm clocks, ms; [] product, [_vf] vgpr-form everywhere, [_sii] iterative-ilp everywhere, [_smo] iterative-maxocc everywhere"
  NT=65536 N=1048576 OPS=train,fwd,inv CFGS="c2" ITERS=30 bash scripts/gpu_ab.sh "" _vf _sii _smo
  echo "== C2 flows: auto (f32 kernels) against PREC=bx3 (barrier-free split-bf16 form)"
  for rep in 1 2; do for prec in "" bx3; do echo "PREC=[$prec]"; PREC=$prec N=1048576 OPS=fwd,inv ITERS=30 python scripts/bench_kernels.py c2 2>&1 | grep "^{" | python3 -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print('  ', ' '.join('%s=%.4f' % (k, v) for k, v in j.items() if k.endswith('_ms')))"; done; done; } > $O/c2_warm.txt 2>&1; cat $O/c2_warm.txt
