# freeze on the final sources (round 6, auto -> bx3 C2 flows)
cd /root/repo
bash scripts/gpu_freeze.sh r06 2>&1 | tail -80
