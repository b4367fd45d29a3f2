cd /root/repo
python -m pytest tests/test_bench_sizes_gpu.py -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -8
FUZZ_WIDE=0 python scripts/autograd_fuzz.py 2>&1 | tail -3
python scripts/lmm64_fuzz.py 2>&1 | tail -2
python scripts/resident_fuzz.py 2>&1 | tail -2
python scripts/cvae_lmm_fuzz.py 2>&1 | tail -2
