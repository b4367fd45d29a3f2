# round 6: the seam's shape hole closed -- autograd tests incl. the wide nets on the VALU kernel
cd /root/repo; O=gpurun_out/r06d; mkdir -p $O
python -m pytest tests/test_autograd_gpu.py -m gpu -x -q 2>&1 | tail -15 > $O/autograd.txt; cat $O/autograd.txt
