"""dev aid: CVAE MFMA step vs generic kernels vs the float64 oracle on random shapes (GPU box)."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probaforms_amd import _hip
from oracle import CvaeOracle, CvaeShape

def dev(a): return None if a is None else torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).cuda()

o64 = CvaeOracle(64)
rng = np.random.default_rng(0)
shapes = [(5, 3, 2, 10), (5, 0, 2, 10), (16, 4, 2, 128), (16, 4, 4, 128), (1, 0, 1, 1), (13, 1, 3, 37), (16, 2, 4, 200), (7, 4, 1, 16), (2, 2, 2, 300)]
for (d, c, lat, h) in shapes:
    for n in (1, 50, 300, 5000):
        shp = _hip.CvaeShape.make(d, c, lat, (h,), "tanh")
        P = _hip.cvae_param_count(shp)
        p = (rng.standard_normal(P) * 0.3).astype(np.float32)
        X = rng.standard_normal((n, d)).astype(np.float32); Cc = rng.standard_normal((n, c)).astype(np.float32) if c else None
        eps = rng.standard_normal((n, lat)).astype(np.float32)
        lo, go = o64.loss_grad(CvaeShape.make(d, c, lat, (h,), "tanh"), p, X, Cc, eps, 0.3)
        ws = torch.empty(_hip.cvae_workspace_bytes(shp, n), dtype=torch.uint8, device="cuda")
        res = {}
        for path in ("mfma", "generic"):
            shp.family = 1 if path == "generic" else 0
            g = torch.full((P,), float("nan"), device="cuda"); l = torch.empty(1, device="cuda")
            _hip.cvae_loss_grad(shp, dev(p), dev(X), dev(Cc), None, dev(eps), n, 1.0 / n, 0.3, g, l, ws)
            torch.cuda.synchronize()
            gg = g.cpu().numpy()
            res[path] = (abs(float(l) - lo) / max(1, abs(lo)), np.abs(gg - go).max() / np.abs(go).max())
        print((d, c, lat, h), n, "path", _hip.cvae_kernel_path(shp), "mfma: loss %.2e grad %.2e | generic: loss %.2e grad %.2e" % (res["mfma"] + res["generic"]), flush=True)
# timing at config 5
d, c, lat, h, n = 16, 4, 2, 128, 65536
shp = _hip.CvaeShape.make(d, c, lat, (h,), "tanh"); P = _hip.cvae_param_count(shp)
p = dev((rng.standard_normal(P) * 0.1).astype(np.float32)); X = torch.randn(n, d, device="cuda"); Cc = torch.randn(n, c, device="cuda"); eps = torch.randn(n, lat, device="cuda")
ws = torch.empty(_hip.cvae_workspace_bytes(shp, n), dtype=torch.uint8, device="cuda"); g = torch.empty(P, device="cuda"); l = torch.empty(1, device="cuda")
for path in ("mfma", "generic"):
    shp.family = 1 if path == "generic" else 0
    for _ in range(5): _hip.cvae_loss_grad(shp, p, X, Cc, None, eps, n, 1.0 / n, 0.001, g, l, ws)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(50): _hip.cvae_loss_grad(shp, p, X, Cc, None, eps, n, 1.0 / n, 0.001, g, l, ws)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 50
    print(path, "loss_grad 65536 rows: %.3f ms  %.1f M rows/s" % (dt * 1e3, n / dt / 1e6))
