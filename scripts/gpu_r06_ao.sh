cd /root/repo
bash scripts/box_probe.sh
python -m pytest tests/test_hip_kernels.py tests/test_bench_sizes_gpu.py tests/test_prior_torch.py tests/test_api_gpu.py -x -q 2>&1 | grep -E "passed|failed|rror|ERROR" | tail -3
OPS=fwd,inv CFGS="c2 c2_nocond" bash scripts/gpu_ab.sh _old ""
for rep in 1 2; do for v in _old ""; do echo -n "bench20 [$v] "; RNVP_HIP_LIB=/root/repo/probaforms_amd/csrc/librnvp_hip$v.so python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-api-level 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f M rows/s  %.3f ms/step  frac %.4f sample %.4f ms' % (j['value'] / 1e6, j['ms_per_step'], j['roofline']['frac'], [v for k, v in j['roofline_kernels'].items() if k.startswith('sample (')][0]['ms_per_launch']))"
done; done
