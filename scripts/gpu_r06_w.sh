# round 6: the fuzzers over the translation units whose compile flags changed (rnvp_lmm: vgpr-form; rnvp_resident*: iterative-ilp + vgpr-form;
# rnvp_bx3 / nf4 / nf8: iterative-ilp are covered by the fixed-shape suite) -- random shapes against the independent paths
cd /root/repo; O=gpurun_out/r06w; mkdir -p $O
{ echo "== scripts/resident_fuzz.py (CASES=300, SEED=6): resident epochs against the batch-by-batch path"; SEED=6 CASES=300 python scripts/resident_fuzz.py 2>&1 | tail -4
  echo "== scripts/cvae_lmm_fuzz.py (CASES=300, SEED=6): CVAE on the any-shape MFMA kernels against one thread per row"; SEED=6 CASES=300 python scripts/cvae_lmm_fuzz.py 2>&1 | tail -3
  echo "== scripts/autograd_fuzz.py 300 6: rnvp_backward_cond / rnvp_inverse_backward (16-row MFMA kernel) against float64 torch autograd"; python scripts/autograd_fuzz.py 300 6 2>&1 | tail -3
  echo "== scripts/lmm64_fuzz.py 6 200: 64-row training form against the 16-row form"; python scripts/lmm64_fuzz.py 6 200 2>&1 | tail -3; } > $O/fuzz.txt 2>&1; cat $O/fuzz.txt | cut -c1-250
