"""CVAE at the reference's defaults (latent 2, hidden (10,), batch 32) on 1000 moon rows: wall time per optimizer step"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probaforms_amd.models import CVAE
f = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "moons_fit.npz"))
X, C = f["X"], f["C"]
for noise in ("host", "device"):
    torch.manual_seed(0)
    m = CVAE(n_epochs=20, noise_rng=noise)
    m.fit(X, C); torch.cuda.synchronize()
    t0 = time.perf_counter(); m.fit(X, C); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    steps = 20 * 32
    print("noise_rng=%s: fit 20 epochs %.1f ms = %.1f us per step; last loss %.4f" % (noise, dt * 1e3, dt / steps * 1e6, float(m.loss_history[-1])))
