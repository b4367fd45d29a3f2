"""API-level RealNVP.sample(C) (numpy in -> numpy out) on the C2 shape with the reference-exact prior stream drawn on the host
against the same stream drawn on the device (HostStreamOnDevice), and the counter-based device prior; plus the draw alone."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from probaforms_amd.models import RealNVP
from probaforms_amd.models.nflow import HostStreamOnDevice as H

n = int(os.environ.get("N", 1_000_000))
X, C = bench.make_data(n, 16, 4, 0)
torch.manual_seed(0)
m = RealNVP(n_layers=8, hidden=(128,), batch_size=65536, n_epochs=1, lr=1e-3)
m.fit(X[:131072], C[:131072])
print("device draw validated against torch.randn on this host:", H.usable("cuda"), "| CPU capability", torch.backends.cpu.get_cpu_capability())

def timed(label, reps=5):
    ts = []
    for _ in range(reps):
        torch.manual_seed(1); torch.cuda.synchronize()
        t0 = time.perf_counter(); xs = m.sample(C); ts.append(time.perf_counter() - t0)
    print("%-44s best %.2f ms median %.2f ms -> %.1f M rows/s" % (label, min(ts) * 1e3, sorted(ts)[len(ts) // 2] * 1e3, n / min(ts) / 1e6))
    return xs

a = timed("sample(%d), host prior drawn on the DEVICE" % n)
saved = dict(H._ok); H._ok = {k: False for k in list(saved) + [0, None]}
b = timed("sample(%d), host prior drawn on the HOST" % n)
H._ok = saved
print("identical bytes:", np.array_equal(a.view(np.uint32), b.view(np.uint32)))
m.prior.host_rng = False
timed("sample(%d), counter-based device prior" % n)
# the draw alone
z = torch.empty(n * 16, device="cuda")
hs = H("cuda")
for _ in range(2):
    hs.begin(); hs.draw(z); hs.end()
torch.cuda.synchronize(); t0 = time.perf_counter(); hs.begin(); hs.draw(z); hs.end(); torch.cuda.synchronize(); t1 = time.perf_counter()
print("device draw of %d normals incl. state round trip: %.2f ms (%.2f G/s)" % (z.numel(), (t1 - t0) * 1e3, z.numel() / (t1 - t0) / 1e9))
t0 = time.perf_counter(); torch.randn(n * 16); t1 = time.perf_counter()
print("torch.randn on the host: %.2f ms" % ((t1 - t0) * 1e3))
