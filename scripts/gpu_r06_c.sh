# round 6, third GPU call: pipelined LDS-fed bx3 unit (V14); transposition-tile swizzle A/B + parity; LDS conflict attribution by ablation
cd /root/repo; O=gpurun_out/r06c; mkdir -p $O
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 scripts/micro/unit_mix.hip -o scripts/micro/unit_mix 2>/dev/null && timeout 300 scripts/micro/unit_mix > $O/unit_mix.txt 2>&1; sed -n 12,20p $O/unit_mix.txt
{ echo "rnvp_loss_grad, ms; [] shipped, [_tt] RNVP_TT_SWIZZLE=1 (conflict-free transposition tiles), [_r5] round 5's library"
  for nt in 65536 16960; do echo "== NT=$nt"; NT=$nt OPS=train CFGS=c2 ITERS=20 bash scripts/gpu_ab.sh "" _tt _r5; done; } > $O/ab_tt.txt 2>&1; cat $O/ab_tt.txt
RNVP_HIP_LIB=/root/repo/probaforms_amd/csrc/librnvp_hip_tt.so python -m pytest tests/test_bench_sizes_gpu.py tests/test_hip_kernels.py -m gpu -x -q 2>&1 | tail -3 > $O/tt_parity.txt; cat $O/tt_parity.txt
export TMPDIR=/tmp; cd /tmp
for v in "" _tt _a1 _a2 _a4; do
  RNVP_HIP_LIB=/root/repo/probaforms_amd/csrc/librnvp_hip$v.so NT=65536 N=1048576 OPS=train ITERS=5 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_LDS_ADDR_CONFLICT --output-format csv -d /root/repo/$O/pmc_lds$v -o p -- python3 /root/repo/scripts/bench_kernels.py c2 > /root/repo/$O/pmc_lds$v.log 2>&1
done
cd /root/repo
python3 - <<'PY' > gpurun_out/r06c/lds_conflicts.txt 2>&1
import csv, glob, collections
print("C2 training kernel k_mfma_train<2,1,4,1,0,true>, 65536 rows: LDS counters per launch (rocprofv3 --pmc, scripts/gpu_r06_c.sh)")
print("variant: '' product; _tt RNVP_TT_SWIZZLE=1; _a1 RNVP_ABLATE=1 (no (h, g_pre) transposition tiles: WRONG results, counters only); _a2 no slot writes; _a4 no flush")
for v in ["", "_tt", "_a1", "_a2", "_a4"]:
    agg = collections.defaultdict(list)
    for f in glob.glob('/root/repo/gpurun_out/r06c/pmc_lds%s/**/*counter_collection.csv' % v, recursive=True):
        for row in csv.DictReader(open(f)):
            if 'k_mfma_train<2, 1, 4, 1, 0' in row['Kernel_Name']: agg[row['Counter_Name']].append(float(row['Counter_Value']))
    print("[%-3s] " % v + "  ".join("%s %.3fM" % (k.replace('SQ_', ''), sum(x) / len(x) / 1e6) for k, x in sorted(agg.items())))
PY
cat gpurun_out/r06c/lds_conflicts.txt
