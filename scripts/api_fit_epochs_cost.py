import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
import bench
from probaforms_amd.models import RealNVP
n = 1_000_000
Xh, Ch = bench.make_data(n, 16, 4, 0)
res = {}
for ep in (1, 5, 10, 20):
    torch.manual_seed(0)
    m = RealNVP(n_layers=8, hidden=(128,), batch_size=65536, n_epochs=ep, lr=1e-3, prior_rng="device")
    m.fit(Xh, Ch); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); m.fit(Xh, Ch); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    res[ep] = min(ts)
    print("n_epochs=%d: fit %.2f ms (%.2f ms per epoch)" % (ep, res[ep] * 1e3, res[ep] * 1e3 / ep))
print("per extra epoch: %.2f ms; fixed: %.2f ms" % ((res[20] - res[10]) / 10 * 1e3, (res[10] - 10 * (res[20] - res[10]) / 10) * 1e3))
