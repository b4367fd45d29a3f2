# round 6, second GPU call: unit_mix with the round-6 variants; RNVP_TRAIN_FT_NF2 = 2 (one-tile flush windows) against the shipped 4
cd /root/repo; O=gpurun_out/r06b; mkdir -p $O
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 scripts/micro/unit_mix.hip -o scripts/micro/unit_mix 2>/dev/null && timeout 300 scripts/micro/unit_mix > $O/unit_mix.txt 2>&1; sed -n 1,26p $O/unit_mix.txt
{ echo "rnvp_loss_grad, ms; [] shipped (FT = 4: flush windows of two hidden tiles), [_ft2] RNVP_TRAIN_FT_NF2 = 2 (windows of one tile; frees 24.5 KB of LDS)"
  for nt in 65536 16960; do echo "== NT=$nt"; NT=$nt OPS=train CFGS=c2 ITERS=20 bash scripts/gpu_ab.sh "" _ft2; done; } > $O/ab_ft2.txt 2>&1; cat $O/ab_ft2.txt
