set -x
cd /root/repo
make -C oracle -s 2>&1 | tail -2
python -m pytest tests -m gpu -x -q 2>&1 | tail -40
python __graft_entry__.py smoke 2>&1 | tail -5
python scripts/bench_kernels.py 2>&1 | grep -E "^\{|rror"
