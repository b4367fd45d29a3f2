# round 6, second call of the freeze: smoke + the bench lines (they read roofline.traffic from profiles/r06_traffic_pmc*.json)
cd /root/repo
O=gpurun_out/r06; mkdir -p $O
python __graft_entry__.py smoke 2>&1 | tail -2
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python bench.py --steps 20 --warmup 3 > $O/bench_steps20.json 2> $O/bench_steps20.err; echo "rc=$?"
python bench.py --workload c3 --no-api-level > $O/bench_c3.json 2> $O/bench_c3.err; echo "rc=$?"
python bench.py --workload c4 --no-api-level > $O/bench_c4.json 2> $O/bench_c4.err; echo "rc=$?"
BENCH_FORCE_DIST=1 python bench.py --steps 5 --warmup 2 --global-batch 8192 --no-cpu-baseline --no-api-level > $O/bench_dp_rank8192_c2.json 2> $O/bench_dp_c2.err; echo "rc=$?"
BENCH_FORCE_DIST=1 python bench.py --steps 3 --warmup 1 --global-batch 8192 --workload c3 --no-cpu-baseline --no-api-level > $O/bench_dp_rank8192_c3.json 2> $O/bench_dp_c3.err; echo "rc=$?"
python - <<'PY'
import json
for f in ("bench", "bench_steps20", "bench_c3", "bench_c4", "bench_dp_rank8192_c2", "bench_dp_rank8192_c3"):
    try:
        j = json.load(open("gpurun_out/r06/%s.json" % f))
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f, "value %.1f M rows/s, ms/step %.3f, blocks %s (%.2f s), frac %.3f traffic %s" % (j["value"] / 1e6, j["ms_per_step"], j.get("timed_blocks"), j.get("timed_blocks", 1) * j["block_seconds"]["median"], j["roofline"]["frac"], j["roofline"].get("traffic")))
    print("   logp mae", j.get("logprob_mae"), "|", j.get("batch_regime"))
    if "api_level" in j: print("   api:", {k: (round(v / 1e6, 1) if isinstance(v, float) else v) for k, v in j["api_level"].items() if k != "note"})
    if "cpu_baseline" in j: print("   cpu:", j["cpu_baseline"]["value"], j["cpu_baseline"]["cores"])
    if "strong_batch_8gpu_projection" in j: print("   proj:", j["strong_batch_8gpu_projection"]["eight_rank_steps_over_one_gpu_step"])
    if "secondary_configs" in j:
        sc = j["secondary_configs"]
        for k in ("c2", "c3"):
            d = sc["dp8_rank_steps"][k]; print("   dp8", k, round(d["rank_step_8192_rows"]["us_per_step"], 1), round(d["rank_step_8192_rows_exchange_in_4_chunks"]["us_per_step"], 1), round(d["one_gpu_step_65536_rows"]["us_per_step"], 1), round(d["eight_rank_steps_over_one_gpu_step"], 2))
        print("   cvae_c5:", sc["cvae_c5"]["ms_per_step"], sc["cvae_c5"]["roofline_frac_f32_mfma"], "| c2_batch32", sc["c2_batch32"]["us_per_step"], "| c3 train", sc["c3"]["train_step_65536_rows"]["kernel_ms"], sc["c3"]["train_step_65536_rows"]["roofline_frac_f32_mfma"], "| h128x128", sc["hidden_128x128"]["ms_per_step"])
    for k, v in j["roofline_kernels"].items():
        print("   ", k[:60], {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if a in ("frac", "ms_per_launch", "prior_draw_ms", "draw_plus_inverse_ms", "rows_per_s")})
PY
