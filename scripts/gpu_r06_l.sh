# round 6: per-file iterative-ilp adopted -- full GPU suite, C3 / C4 / C2 kernel times, the c3 / c4 bench lines
cd /root/repo; O=gpurun_out/r06l; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3 > $O/gpu_tests.txt; cat $O/gpu_tests.txt
NT=65536 N=1048576 OPS=train,fwd,inv CFGS="c2 c3 c4" ITERS=10 bash scripts/gpu_ab.sh "" > $O/kernels.txt 2>&1; cat $O/kernels.txt
python bench.py --workload c3 --no-api-level > $O/bench_c3.json 2> $O/bench_c3.err; echo "rc=$?"
python bench.py --workload c4 --no-api-level > $O/bench_c4.json 2> $O/bench_c4.err; echo "rc=$?"
python bench.py --steps 20 > $O/bench20.json 2> $O/bench20.err; echo "rc=$?"
python3 - <<'PY'
import json
for f in ("bench20", "bench_c3", "bench_c4"):
    j = json.load(open("gpurun_out/r06l/%s.json" % f)); print(f, "value %.1f M rows/s, ms/step %.3f, frac %.3f" % (j["value"] / 1e6, j["ms_per_step"], j["roofline"]["frac"]), j.get("logprob_mae"))
    if "secondary_configs" in j:
        sc = j["secondary_configs"]; print("   c3 train", sc["c3"]["train_step_65536_rows"]["kernel_ms"], sc["c3"]["train_step_65536_rows"]["roofline_frac_f32_mfma"], "c1 defaults", sc.get("c1_defaults_batch32", {}).get("us_per_step"), "dp8 c3", sc["dp8_rank_steps"]["c3"]["rank_step_8192_rows"]["us_per_step"], sc["dp8_rank_steps"]["c3"]["eight_rank_steps_over_one_gpu_step"])
PY
