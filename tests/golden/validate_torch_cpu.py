"""Build container only (imports the reference from /root/reference): checks that oracle/torch_cpu.py -- the eager-PyTorch
CPU loop bench.py times as `cpu_baseline` on the GPU box -- computes what the reference computes and runs at the
reference's speed on the same cores (within +-10 %).  Usage: python tests/golden/validate_torch_cpu.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True
from probaforms.models import RealNVP as RefRealNVP          # noqa: E402  (the reference itself)
assert "/root/reference" in sys.modules["probaforms"].__file__
from oracle.torch_cpu import EagerFlow, timed_fit_and_sample   # noqa: E402

L, d, c, hidden, bs = 8, 16, 4, (128,), 65536
n = int(os.environ.get("N", 196608))
threads = int(os.environ.get("THREADS", os.cpu_count() or 1))
torch.set_num_threads(threads)
rng = np.random.default_rng(0)
X = rng.standard_normal((n, d)).astype(np.float32); C = rng.standard_normal((n, c)).astype(np.float32)

# --- outputs: the reference's weights loaded into the eager loop -> same log-prob and samples
torch.manual_seed(0)
ref = RefRealNVP(n_layers=L, hidden=hidden, batch_size=bs, n_epochs=1, lr=1e-3)
ref.fit(X[:bs], C[:bs])                                           # builds the model (+ one step)
flat = torch.cat([p.detach().reshape(-1) for p in ref.nf.parameters()]).numpy()
mine = EagerFlow(L, d, c, hidden); mine.load_flat(flat)
xt, ct = torch.from_numpy(X[:4096]), torch.from_numpy(C[:4096])
with torch.no_grad():
    lp_mine = mine.log_prob_rows(xt, ct)[0]
    lp_ref = float(ref.nf.log_prob(xt, ct))
    torch.manual_seed(3); s_mine = mine.sample(ct)
    torch.manual_seed(3); s_ref = ref.nf.sample(ct)
print("mean log-prob: eager %.6f  reference %.6f" % (float(lp_mine.mean()), lp_ref))
print("sample max |diff| %.2e" % float((s_mine - s_ref).abs().max()))
assert abs(float(lp_mine.mean()) - lp_ref) < 1e-5 and float((s_mine - s_ref).abs().max()) < 1e-4

# --- speed: one epoch + sampling n rows, same cores
best = None
for rep in range(3):
    t0 = time.perf_counter(); ref.fit(X, C); t_fit = time.perf_counter() - t0
    t0 = time.perf_counter(); ref.sample(C); t_s = time.perf_counter() - t0
    r = dict(fit=n / t_fit, sample=n / t_s, combined=2 * n / (t_fit + t_s))
    best = r if best is None or r["combined"] > best["combined"] else best
print("reference  (%d threads, n=%d): fit %.1f k rows/s  sample %.1f k rows/s  combined %.1f k rows/s" %
      (threads, n, best["fit"] / 1e3, best["sample"] / 1e3, best["combined"] / 1e3))
mb = None
for rep in range(3):
    r = timed_fit_and_sample(L, d, c, hidden, X, C, bs, threads)
    mb = r if mb is None or r["combined_rows_per_s"] > mb["combined_rows_per_s"] else mb
print("eager loop (%d threads, n=%d): fit %.1f k rows/s  sample %.1f k rows/s  combined %.1f k rows/s" %
      (threads, n, mb["fit_rows_per_s"] / 1e3, mb["sample_rows_per_s"] / 1e3, mb["combined_rows_per_s"] / 1e3))
print("ratio eager / reference: fit %.2f  sample %.2f  combined %.2f" %
      (mb["fit_rows_per_s"] / best["fit"], mb["sample_rows_per_s"] / best["sample"], mb["combined_rows_per_s"] / best["combined"]))
