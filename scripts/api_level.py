"""API-level (numpy in -> numpy out) timing of the C2 workload through RealNVP.fit / .sample,
with a breakdown of the host-side pieces (API level vs device resident)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from probaforms_amd.models import RealNVP
from probaforms_amd import _engine

n = int(os.environ.get("N", 1_000_000))
X, C = bench.make_data(n, 16, 4, 0)
torch.manual_seed(0)
m = RealNVP(n_layers=8, hidden=(128,), batch_size=65536, n_epochs=1, lr=1e-3)
t0 = time.perf_counter(); m.fit(X, C); torch.cuda.synchronize(); t1 = time.perf_counter()
print("first fit (1 epoch, incl. init/alloc) %.1f ms" % ((t1 - t0) * 1e3))
m.n_epochs = int(os.environ.get("EPOCHS", 4))
t0 = time.perf_counter(); m.fit(X, C); torch.cuda.synchronize(); t1 = time.perf_counter()
print("fit %d epochs: %.1f ms -> %.2f M rows/s API level" % (m.n_epochs, (t1 - t0) * 1e3, n * m.n_epochs / (t1 - t0) / 1e6))
t0 = time.perf_counter(); p = _engine.loader_permutation(n); t1 = time.perf_counter()
print("  loader_permutation(%d): %.1f ms" % (n, (t1 - t0) * 1e3))
t0 = time.perf_counter(); Xd = torch.tensor(X, dtype=torch.float32, device="cuda"); torch.cuda.synchronize(); t1 = time.perf_counter()
print("  H2D X (%.0f MB): %.1f ms" % (X.nbytes / 1e6, (t1 - t0) * 1e3))
t0 = time.perf_counter(); xs = m.sample(C); t1 = time.perf_counter()
print("sample(%d): %.1f ms -> %.2f M rows/s API level" % (n, (t1 - t0) * 1e3, n / (t1 - t0) / 1e6))
t0 = time.perf_counter(); z = torch.randn(n, 16); t1 = time.perf_counter()
print("  host randn: %.1f ms" % ((t1 - t0) * 1e3))
t0 = time.perf_counter(); h = torch.empty(n, 16, device="cuda").cpu(); t1 = time.perf_counter()
print("  D2H: %.1f ms" % ((t1 - t0) * 1e3))
