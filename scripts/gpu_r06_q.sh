# round 6: per-file vgpr-form adopted (rnvp_lmm, rnvp_resident*) -- full GPU suite + the affected timings
cd /root/repo; O=gpurun_out/r06q; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3 > $O/gpu_tests.txt; cat $O/gpu_tests.txt
{ python scripts/lmm16_time.py 2>&1 | tail -4 | cut -c1-200; python scripts/resident_time.py 2>&1 | tail -4 | cut -c1-150; python scripts/lmm64_time.py 128,128 2>&1 | tail -3 | cut -c1-170; } > $O/times.txt 2>&1; cat $O/times.txt
