# round 6: random WIDE shapes through the differentiable seam (the VALU kernel's new seeds) against float64 torch autograd
cd /root/repo; O=gpurun_out/r06g; mkdir -p $O
FUZZ_WIDE=1 python scripts/autograd_fuzz.py 120 7 > $O/autograd_fuzz_wide.txt 2>&1; tail -12 $O/autograd_fuzz_wide.txt
python - <<'PY' >> gpurun_out/r06g/autograd_fuzz_wide.txt 2>&1
import sys; sys.path.insert(0, "/root/repo")
from probaforms_amd import _hip
print("kernel family per shape (rnvp_backward_cond_workspace_bytes > 0 everywhere; lmm16 serves the shape iff its train workspace query for family LMM16 is non-zero)")
PY
