cd /root/repo; O=gpurun_out/r06ab; mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|rror|FAILED" | tail -15 > $O/gpu_tests.txt; cat $O/gpu_tests.txt
python bench.py --steps 20 --no-cpu-baseline > $O/bench20.json 2> $O/bench20.err; python3 -c "
import json; j = json.load(open('$O/bench20.json'))
print('bench20', j['value'], j['ms_per_step'], j['roofline']['frac'], j['logprob_mae'], j['dtype'][:120])
for k, v in j['roofline_kernels'].items(): print('  ', k[:70], round(v['frac'], 4), round(v['ms_per_launch'], 4), v.get('frac_mixed_bound'))
print(j['api_level'])"
