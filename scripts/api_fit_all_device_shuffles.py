import sys, time, os, numpy as np, torch
sys.path.insert(0, "/root/repo")
import bench
from probaforms_amd import _engine
from probaforms_amd.models import RealNVP
n = 1_000_000
Xh, Ch = bench.make_data(n, 16, 4, 0)
real = _engine.effective_cpus
for label, fake in (("first two epochs on the device", None), ("every epoch on the device", 2), ("first two epochs on the device", None), ("every epoch on the device", 2)):
    _engine.effective_cpus = (lambda: fake) if fake else real
    torch.manual_seed(0)
    m = RealNVP(n_layers=8, hidden=(128,), batch_size=65536, n_epochs=1, lr=1e-3)
    m.fit(Xh, Ch); m.n_epochs = 10; torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); m.fit(Xh, Ch); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("%s: 10-epoch fit of 1M rows, ms:" % label, " ".join("%.1f" % t for t in ts))
