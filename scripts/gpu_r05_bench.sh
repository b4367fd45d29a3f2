# round 5: bench.py as the driver runs it (20 steps), the default (100 steps), and the per-rank regime of the 8-GPU configuration on one GPU
cd /root/repo
O=gpurun_out/r05; mkdir -p $O
python bench.py --steps 20 --warmup 3 > $O/bench_steps20.json 2> $O/bench_steps20.err; echo "rc=$?"; tail -3 $O/bench_steps20.err
BENCH_FORCE_DIST=1 python bench.py --steps 5 --warmup 2 --global-batch 8192 --no-cpu-baseline --no-api-level > $O/bench_dp_rank8192_c2.json 2> $O/bench_dp_c2.err; echo "rc=$?"; tail -3 $O/bench_dp_c2.err
BENCH_FORCE_DIST=1 python bench.py --steps 3 --warmup 1 --global-batch 8192 --workload c3 --no-cpu-baseline --no-api-level > $O/bench_dp_rank8192_c3.json 2> $O/bench_dp_c3.err; echo "rc=$?"; tail -3 $O/bench_dp_c3.err
python - <<'PY'
import json
for f in ("bench_steps20", "bench_dp_rank8192_c2", "bench_dp_rank8192_c3"):
    try:
        j = json.load(open("gpurun_out/r05/%s.json" % f))
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f, "value %.1f M rows/s, ms/step %.3f, blocks %s, frac %.3f" % (j["value"] / 1e6, j["ms_per_step"], j.get("timed_blocks"), j["roofline"]["frac"]), j["config"].get("rank_batch"), j["config"].get("global_batch"))
    print("   kernel:", j["roofline"]["kernel"][:200])
    if "api_level" in j: print("   api:", {k: (round(v / 1e6, 1) if isinstance(v, float) else v) for k, v in j["api_level"].items() if k != "note"})
    if "secondary_configs" in j:
        sc = j["secondary_configs"]
        print("   dp8:", json.dumps(sc.get("dp8_rank_steps"))[:1500])
        print("   cvae_c5:", sc["cvae_c5"]["ms_per_step"], sc["cvae_c5"]["roofline_frac_f32_mfma"])
    for k, v in j["roofline_kernels"].items():
        print("   ", k[:60], {a: b for a, b in v.items() if a in ("frac", "ms_per_launch", "prior_draw_ms", "draw_plus_inverse_ms", "rows_per_s")})
PY
