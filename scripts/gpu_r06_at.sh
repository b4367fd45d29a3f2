cd /root/repo
OPS=train ITERS=300 WARM_S=0.2 NT=65536 CFGS="c2" bash scripts/gpu_ab.sh "" _inpl _inpl2
for rep in 1 2; do for v in "" _inpl _inpl2; do echo -n "bench20 [$v] "; RNVP_HIP_LIB=/root/repo/probaforms_amd/csrc/librnvp_hip$v.so python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-api-level 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f M rows/s  %.3f ms/step  frac %.4f' % (j['value'] / 1e6, j['ms_per_step'], j['roofline']['frac']))"
done; done
