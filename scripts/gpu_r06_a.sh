# round 6, first GPU call: the pruned tree's tests + bench, the A/Bs that decide items 1 and 2 of VERDICT r05
cd /root/repo; O=gpurun_out/r06a; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/gpu_tests.txt; cat $O/gpu_tests.txt
python bench.py --steps 20 > $O/bench20.json 2> $O/bench20.err; python3 -c "
import json; j = json.load(open('$O/bench20.json')); print('bench20', j['value'], j['ms_per_step'], j['timed_blocks'], j['roofline']['frac'], j.get('logprob_mae'))"
# A/B in one box: round 5's library against the pruned one (full and ragged batch), the ragged-launch candidates, write-through partials
{ echo "rnvp_loss_grad, ms (scripts/bench_kernels.py); [_r5] round 5's library, [] pruned product, [_wt] write-through partial stores"
  for nt in 65536 16960; do echo "== NT=$nt"; NT=$nt OPS=train CFGS=c2 ITERS=20 bash scripts/gpu_ab.sh _r5 "" _wt; done
  echo "== ragged 16960 rows: [] R=2 net split on 133 workgroups, [_fr1] R=1 row-parallel on 265 workgroups (2 per CU), [_fr1ns] R=1 net split, 256 workgroups, 9 with two groups"
  NT=16960 OPS=train CFGS=c2 ITERS=20 bash scripts/gpu_ab.sh "" _fr1 _fr1ns; } > $O/ab_train.txt 2>&1; cat $O/ab_train.txt
# the whole step with write-through partials (the gap behind the training kernel is what it is after)
for v in "" _wt; do RNVP_HIP_LIB=/root/repo/probaforms_amd/csrc/librnvp_hip$v.so python bench.py --steps 20 --no-cpu-baseline --no-api-level > $O/bench_plain$v.json 2>/dev/null
python3 -c "
import json; j = json.load(open('$O/bench_plain$v.json')); print('bench plain [$v]', j['value'], j['ms_per_step'], j['roofline']['frac'])"; done
# micro benchmarks
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -Wno-unused-result scripts/micro/finish_read.hip -o scripts/micro/finish_read && scripts/micro/finish_read > $O/finish_read.txt 2>&1; cat $O/finish_read.txt
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -Wno-unused-result scripts/micro/unit_mix.hip -o scripts/micro/unit_mix && timeout 300 scripts/micro/unit_mix > $O/unit_mix.txt 2>&1; sed -n 1,24p $O/unit_mix.txt
# kernel timeline of the fit epoch: durations and the gaps between consecutive launches
export TMPDIR=/tmp; cd /tmp
CASES="16,4,128,8,1000000,65536" REPS=3 rocprofv3 --kernel-trace --output-format csv -d /root/repo/$O/trace -o ep -- python3 /root/repo/scripts/step_profile.py > /root/repo/$O/trace.log 2>&1
cd /root/repo
python3 - <<'PY' > gpurun_out/r06a/epoch_timeline.txt 2>&1
import csv, glob
f = glob.glob('/root/repo/gpurun_out/r06a/trace/**/ep_kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_mfma_train' in r['Kernel_Name']]
a = idx[-20]
prev_end = None
gaps = {}
for r in rows[a:a + 36]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp']); n = r['Kernel_Name'].split('(')[0][-50:]
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print('gap %6.2f us   dur %8.2f us   %s' % (gap, (e - s) / 1e3, n)); prev_end = e
# averages over the whole trace
import collections
d = collections.defaultdict(list); g = collections.defaultdict(list); prev = None
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp']); n = r['Kernel_Name'].split('(')[0].split('::')[-1][:40]
    d[n].append((e - s) / 1e3)
    if prev is not None and (s - prev[1]) < 50000: g[prev[0] + ' -> ' + n].append((s - prev[1]) / 1e3)
    prev = (n, e)
for k, v in d.items(): print('dur  %-44s n %5d avg %8.2f us' % (k, len(v), sum(v) / len(v)))
for k, v in g.items(): print('gap  %-80s n %5d avg %6.2f us' % (k, len(v), sum(v) / len(v)))
PY
cat gpurun_out/r06a/epoch_timeline.txt | tail -30
