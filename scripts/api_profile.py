"""cProfile of the host side of one small fit (the reference's test size and the README example), after a warm-up fit"""
import os, sys, cProfile, pstats, io
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probaforms_amd.models import RealNVP, CVAE
rng = np.random.default_rng(0)
X = rng.normal(size=(100, 5)); C = rng.normal(size=(100, 3))
f = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "moons_fit.npz"))
for label, make, x, c in [("RealNVP() n=100", RealNVP, X, C), ("README RealNVP", lambda: RealNVP(lr=0.01, n_epochs=100), f["X"], f["C"]), ("CVAE() n=100", CVAE, X, C)]:
    make().fit(x, c); torch.cuda.synchronize()
    m = make(); pr = cProfile.Profile(); pr.enable(); m.fit(x, c); torch.cuda.synchronize(); pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22)
    print("=====", label); print("\n".join(l[:150] for l in s.getvalue().splitlines()[:40]))
