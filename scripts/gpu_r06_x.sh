cd /root/repo; O=gpurun_out/r06x; mkdir -p $O
python -m pytest tests/test_api_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -2
python scripts/api_sample_prior.py > $O/api_sample_prior.txt 2>&1; grep -v amdgpu.ids $O/api_sample_prior.txt
