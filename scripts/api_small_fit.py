"""API-level wall time of the reference's own test sizes (tests/test_models.py: n=100, d=5, cond 3 or none, default models)
and of the README example (1000 rows of 2-d moons, n_epochs=100)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probaforms_amd.models import RealNVP, CVAE
rng = np.random.default_rng(0)
def timed(label, make, X, C, reps=3):
    ts = []
    for _ in range(reps):
        m = make(); torch.cuda.synchronize(); t0 = time.perf_counter(); m.fit(X, C); torch.cuda.synchronize(); t1 = time.perf_counter()
        xs = m.sample(C if C is not None else len(X)); t2 = time.perf_counter(); ts.append((t1 - t0, t2 - t1))
    steps = len(m.loss_history) if isinstance(m, RealNVP) else m.n_epochs * ((len(X) + m.batch_size - 1) // m.batch_size)
    print("%-55s fit %.2f ms (%d steps, %.1f us/step wall), sample %.2f ms   [best of %d]" % (
        label, min(t[0] for t in ts) * 1e3, steps, min(t[0] for t in ts) / steps * 1e6, min(t[1] for t in ts) * 1e3, reps), flush=True)
X = rng.normal(size=(100, 5)); C = rng.normal(size=(100, 3))
timed("RealNVP() n=100 d=5 cond=3 (reference test)", RealNVP, X, C)
timed("RealNVP() n=100 d=5 no cond (reference test)", RealNVP, X, None)
timed("CVAE() n=100 d=5 cond=3 (reference test)", CVAE, X, C)
timed("CVAE() n=100 d=5 no cond (reference test)", CVAE, X, None)
f = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "moons_fit.npz"))
timed("README: RealNVP(lr=0.01, n_epochs=100), 1000 x 2-d moons", lambda: RealNVP(lr=0.01, n_epochs=100), f["X"], f["C"])
timed("CVAE(n_epochs=100) on the same rows", lambda: CVAE(n_epochs=100), f["X"], f["C"])
