# round 6: the C2 f32 flow kernels' counters with the same warm-up as the split-bf16 ones (r06_flow_pmc_c2.json): clock and cycles side by side
cd /root/repo
PREC=f32 N=1048576 bash scripts/gpu_pmc.sh r06flowc2f32 c2 fwd,inv > gpurun_out/r06_flow_pmc_c2_f32.log 2>&1
python scripts/make_train_pmc.py r06 r06flowc2f32 "k_mfma_flow<2, 1, 4" r06_flow_pmc_c2_f32.json "C2 forward / inverse, 1M rows, f32 kernels (precision='f32'; auto ran these for C2 in rounds 4-5)"
