# round 6: d <= 16 flows, f32 kernels (what auto runs) against the barrier-free split-bf16 form, now that rnvp_bx3.hip is compiled under iterative-ilp
cd /root/repo; O=gpurun_out/r06n; mkdir -p $O
{ echo "scripts/bench_kernels.py c2, 1M rows, ms; PREC unset = auto (f32 flow kernels), PREC=bx3 = k_flow_bx3 direct (3 row tiles, 4 waves)"
for rep in 1 2 3; do for prec in "" bx3; do echo "PREC=[$prec] rep $rep"; PREC=$prec N=1048576 OPS=fwd,inv python scripts/bench_kernels.py c2 2>&1 | grep "^{" | python3 -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print('  ', ' '.join('%s=%.4f' % (k, v) for k, v in j.items() if k.endswith('_ms')))"; done; done; } > $O/c2_flow_prec.txt 2>&1; cat $O/c2_flow_prec.txt
