# round 6: chunked data-parallel exchange -- parity on one rank (RCCL side stream, caller's exchange) and on two gloo ranks; the one-rank cost of chunking
cd /root/repo; O=gpurun_out/r06e; mkdir -p $O
python -m pytest tests/test_dist_gpu.py -m gpu -x -q 2>&1 | tail -8 > $O/dist_tests.txt; cat $O/dist_tests.txt
python - <<'PY' > $O/dp_chunks.txt 2>&1
import sys, json, torch
sys.path.insert(0, "/root/repo")
import bench
bench.N_ROWS, bench.D, bench.CDIM, bench.LAYERS, bench.HIDDEN = 1_000_000, 16, 4, 8, (128,)
r = bench.rank_steps_of_8(torch.device("cuda", 0))
for k in ("c2", "c3"):
    for n in ("rank_step_8192_rows", "rank_step_8192_rows_exchange_in_4_chunks", "one_gpu_step_65536_rows"):
        print(k, n, "%.1f us per step, training kernel %.1f us" % (r[k][n]["us_per_step"], r[k][n]["kernel_us"]))
    print(k, "eight_rank_steps_over_one_gpu_step %.3f" % r[k]["eight_rank_steps_over_one_gpu_step"])
PY
cat $O/dp_chunks.txt
