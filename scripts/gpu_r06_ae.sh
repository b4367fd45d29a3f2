# round 6: the C2 training call with the forward phase's GEMM1 on f32 against split-bf16, WARM clocks (>= 0.3 s of warm-up, 200 timed calls)
cd /root/repo
for rep in 1 2 3; do for p in f32 bx3; do
  echo -n "rep $rep precision $p: "; PREC=$p OPS=train ITERS=200 WARM_S=0.3 NT=65536 python scripts/bench_kernels.py c2 2>&1 | grep -E "^\{" | python3 -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print(' '.join('%s=%.4f' % (k, v) for k, v in j.items() if k.endswith('_ms')))"
done; done
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-api-level > gpurun_out/r06_bench_ae.json 2>/dev/null
python -c "
import json; j=json.load(open('gpurun_out/r06_bench_ae.json')); print(j['value'], j['ms_per_step']); print(j['secondary_configs']['c2_precision_ab'])"
