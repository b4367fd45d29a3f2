"""README.md:45-65 of the reference, on the build: make_moons n=1000, defaults (L=8, h=(10,), bs=32),
lr=0.01, n_epochs=100 -> 3200 optimizer steps.  Prints wall time for fit and sample."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probaforms_amd.models import RealNVP
f = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "moons_fit.npz"))
X, C = f["X"], f["C"]
torch.manual_seed(0)
m = RealNVP(lr=0.01, n_epochs=int(os.environ.get("EPOCHS", 100)))
t0 = time.perf_counter(); m.fit(X, C); torch.cuda.synchronize(); t1 = time.perf_counter()
xs = m.sample(C); t2 = time.perf_counter()
print("fit %.3f s (%d steps, %.1f us/step, %.0f row-visits/s)  final loss %.4f  sample %.4f s" %
      (t1 - t0, len(m.loss_history), (t1 - t0) / len(m.loss_history) * 1e6, 1000 * m.n_epochs / (t1 - t0),
       float(m.loss_history[-1]), t2 - t1))
t0 = time.perf_counter(); m.fit(X, C); torch.cuda.synchronize(); t1 = time.perf_counter()
print("second fit %.3f s (%.1f us/step)" % (t1 - t0, (t1 - t0) / (32 * m.n_epochs) * 1e6))
