# round 6: where does split-bf16 GEMM1 pay for d <= 16?  hidden width sweep, warm clocks, f32 against bx3 (forward / inverse on 1M rows, loss+grad call on 65536)
cd /root/repo
for h in 16 32 64 96 128 256; do for p in f32 bx3; do
  echo -n "hidden $h precision $p: "; PREC=$p ITERS=60 WARM_S=0.1 N=1048576 NT=65536 python scripts/bench_kernels.py 8,16,4,$h 2>&1 | grep -E "^\{" | python3 -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print(' '.join('%s=%.4f' % (k, v) for k, v in j.items() if k.endswith('_ms')))"
done; done
