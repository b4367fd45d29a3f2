# one line per box: the C2 flow kernels (1M rows, warm) on f32 against split-bf16 -- boxes of the pool differ (round 6: 0.94-1.07 ms for the same kernel)
cd /root/repo
for p in f32 bx3; do echo -n "box_probe $(hostname) precision $p: "; PREC=$p OPS=fwd,inv ITERS=30 WARM_S=0.1 N=1048576 python scripts/bench_kernels.py c2 2>&1 | grep -E "^\{" | python3 -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print(' '.join('%s=%.4f' % (k, v) for k, v in j.items() if k.endswith('_ms')))"
done
