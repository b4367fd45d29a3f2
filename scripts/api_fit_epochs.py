"""API-level RealNVP.fit over several epochs of the C2 arrays: wall time per epoch against the device time of one
epoch (the gap is host work between epochs: permutation upload, loss read-back)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from probaforms_amd.models import RealNVP

n = int(os.environ.get("N", 1_000_000))
X, C = bench.make_data(n, 16, 4, 0)
torch.manual_seed(0)
m = RealNVP(n_layers=8, hidden=(128,), batch_size=65536, n_epochs=1, lr=1e-3)
m.fit(X, C); torch.cuda.synchronize()
for ep in (1, 4, 16, 32):
    m.n_epochs = ep
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); m.fit(X, C); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print("fit %2d epochs: %7.1f ms = %6.2f ms/epoch -> %6.1f M rows/s" % (ep, best * 1e3, best * 1e3 / ep, n * ep / best / 1e6))

# where the host time of one fit() goes
from probaforms_amd import _engine
acc = {"perm_get": 0.0, "upload": 0.0, "launch": 0.0, "to_dev": 0.0}
_get = _engine.PermutationPrefetcher.get
def get(self, e):
    t0 = time.perf_counter(); r = _get(self, e); acc["perm_get"] += time.perf_counter() - t0; return r
_engine.PermutationPrefetcher.get = get
_fe = _engine.FlowEngine.fit_epoch
def fe(self, *a, **k):
    t0 = time.perf_counter(); r = _fe(self, *a, **k); acc["launch"] += time.perf_counter() - t0; return r
_engine.FlowEngine.fit_epoch = fe
import probaforms_amd.models.realnvp as R
_td = R._to_device_f32
def td(*a, **k):
    t0 = time.perf_counter(); r = _td(*a, **k); torch.cuda.synchronize(); acc["to_dev"] += time.perf_counter() - t0; return r
R._to_device_f32 = td
_cpu = torch.Tensor.cpu
acc["losses_cpu"] = 0.0
def cpu(self, *a, **k):
    t0 = time.perf_counter(); r = _cpu(self, *a, **k); acc["losses_cpu"] += time.perf_counter() - t0; return r
torch.Tensor.cpu = cpu
_to = torch.Tensor.to
def to(self, *a, **k):
    t0 = time.perf_counter(); r = _to(self, *a, **k)
    if self.dtype == torch.int64 and self.numel() == n: acc["upload"] += time.perf_counter() - t0
    return r
torch.Tensor.to = to
m.n_epochs = 32
t0 = time.perf_counter(); m.fit(X, C); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("instrumented fit 32 epochs: %.1f ms; host seconds in: %s" % (dt * 1e3, {k: round(v * 1e3, 1) for k, v in acc.items()}))
