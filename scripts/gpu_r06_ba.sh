cd /root/repo
bash scripts/box_probe.sh
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-api-level 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench20 %.1f M rows/s  %.3f ms/step  frac %.4f sample %.4f ms traffic %s' % (j['value'] / 1e6, j['ms_per_step'], j['roofline']['frac'], [v for k, v in j['roofline_kernels'].items() if k.startswith('sample (')][0]['ms_per_launch'], j['roofline']['traffic']))"
