set -x
cd /root/repo
make -C oracle -s 2>&1 | tail -2
python -m pytest tests/test_hip_kernels.py -m gpu -x -q 2>&1 | tail -40
