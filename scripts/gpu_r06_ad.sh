# round 6: the barrier-free split-bf16 flow kernel (C2) at ONE wave per SIMD (512 registers: no scratch) with 3..6 row tiles per wave,
# against the product (two waves per SIMD, 3 row tiles, 57 spilled registers)
cd /root/repo
OPS=fwd,inv CFGS=c2 bash scripts/gpu_ab.sh "" _w1r3 _w1r4 _w1r5 _w1r6
