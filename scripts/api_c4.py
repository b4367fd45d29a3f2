"""API-level timing of the C4 workload (inverse only, d=64, c=16, L=8, h=128) through RealNVP.sample,
with the host-side pieces timed separately."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probaforms_amd.models import RealNVP

n = int(os.environ.get("N", 16_000_000)); d, c = 64, 16
def T(label, fn, sync=True):
    t0 = time.perf_counter(); r = fn()
    if sync: torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%-44s %8.1f ms" % (label, dt * 1e3), flush=True); return r
rng = np.random.default_rng(0)
Xs = rng.standard_normal((4096, d)).astype(np.float32); Cs = rng.standard_normal((4096, c)).astype(np.float32)
torch.manual_seed(0)
m = RealNVP(n_layers=8, hidden=(128,), batch_size=4096, n_epochs=1, lr=1e-3); m.fit(Xs, Cs)
C = rng.standard_normal((n, c), dtype=np.float32)
print("n = %d, out %.2f GB" % (n, n * d * 4 / 1e9))
T("warm sample(65536)", lambda: m.sample(C[:65536]))
for rep in range(2):
    x = T("RealNVP.sample(C) host prior", lambda: m.sample(C))
m2 = RealNVP(n_layers=8, hidden=(128,), batch_size=4096, n_epochs=1, lr=1e-3, prior_rng='device'); m2.fit(Xs, Cs)
for rep in range(2):
    x = T("RealNVP.sample(C) device prior", lambda: m2.sample(C))
del x
if os.environ.get("PIECES", "1") == "1":
    z = T("host randn(n, d)", lambda: torch.randn(n, d), sync=False)
    zp = T("pinned alloc (n, d)", lambda: torch.empty(n, d, pin_memory=True), sync=False)
    T("host randn into pinned", lambda: torch.randn(n, d, out=zp), sync=False)
    zd = T("H2D pageable", lambda: z.to("cuda"))
    T("H2D pinned", lambda: zd.copy_(zp, non_blocking=True))
    T("D2H pageable (.cpu())", lambda: zd.cpu())
    T("D2H pinned", lambda: zp.copy_(zd, non_blocking=True))
    eng = m.nf.engine(); Cd = torch.from_numpy(C).cuda()
    T("inverse kernel", lambda: eng.inverse(zd, Cd, out=zd))
    T("inverse kernel again", lambda: eng.inverse(zd, Cd, out=zd))
