# round 6: the LDS-staged split-bf16 flow kernels' row tiles per wave under WARM clocks (rounds 2-3 chose R = 4 for C3 / 2 for C4 from cold timings)
cd /root/repo
OPS=fwd,inv CFGS="c3" bash scripts/gpu_ab.sh "" _r43 _r45
OPS=fwd,inv CFGS="c4" bash scripts/gpu_ab.sh "" _r81 _r83
