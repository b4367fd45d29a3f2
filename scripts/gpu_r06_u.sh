cd /root/repo; O=gpurun_out/r06u; mkdir -p $O
python bench.py --steps 20 --no-cpu-baseline > $O/bench20.json 2> $O/bench20.err; python3 -c "
import json; j = json.load(open('$O/bench20.json')); sc = j['secondary_configs']
print('bench20', j['value'], j['ms_per_step'], j['roofline']['frac'])
for k in ('c3', 'c4'):
    for n, v in sc[k].items():
        if isinstance(v, dict) and 'kernel_ms' in v: print(k, n, round(v['kernel_ms'], 4), round(v['roofline_frac_f32_mfma'], 3))
print('cvae', sc['cvae_c5']['ms_per_step'], sc['cvae_c5']['kernel_ms'], sc['cvae_c5']['roofline_frac_f32_mfma'], 'c2b32', sc['c2_batch32']['us_per_step'], sc['c2_batch1024']['us_per_step'])"
