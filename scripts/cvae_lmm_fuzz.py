"""fuzz: random CVAE shapes, cvae_loss_grad on the any-shape MFMA kernels (family lmm) against the one-thread-per-row kernels
(family generic) on the same inputs; encoder / decoder alone too.  Exit code 1 on a mismatch."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probaforms_amd import _hip
rng = np.random.default_rng(int(os.environ.get("SEED", 0)))
N = int(os.environ.get("CASES", 200))
def dev(a): return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).cuda()
bad = 0; tried = 0; worst = (0.0, None)
while tried < N:
    d = int(rng.integers(1, 41)); c = int(rng.integers(0, 21)); lat = int(rng.integers(1, 13))
    hidden = tuple(int(v) for v in rng.integers(1, 101, size=int(rng.integers(1, 4)))); act = str(rng.choice(["tanh", "relu"]))
    n = int(rng.integers(1, 300))
    sl = _hip.CvaeShape.make(d, c, lat, hidden, act, family="lmm"); sg = _hip.CvaeShape.make(d, c, lat, hidden, act, family="generic")
    if _hip.cvae_kernel_path(sl) != _hip.PATH_LMM: continue
    P = _hip.cvae_param_count(sl)
    p = dev(rng.standard_normal(P) * 0.2); x = dev(rng.standard_normal((n, d))); cc = dev(rng.standard_normal((n, c))) if c else None
    eps = dev(rng.standard_normal((n, lat))); idx = torch.from_numpy(rng.permutation(n).astype(np.int64)).cuda() if rng.integers(0, 2) else None
    out = {}
    try:
        for name, sh in (("lmm", sl), ("gen", sg)):
            ws = torch.empty(_hip.cvae_workspace_bytes(sh, n), dtype=torch.uint8, device="cuda")
            g = torch.full((P + 1,), float("nan"), device="cuda")
            _hip.cvae_loss_grad(sh, p, x, cc, idx, eps, n, 1.0 / n, 0.3, g[:P], g[P:], ws)
            mu = torch.empty(n, lat, device="cuda"); ls = torch.empty_like(mu); xr = torch.empty(n, d, device="cuda")
            _hip.cvae_encode(sh, p, x, cc, n, mu, ls, ws); _hip.cvae_decode(sh, p, eps, cc, n, xr, ws)
            out[name] = [t.cpu().numpy().astype(np.float64) for t in (g, mu, ls, xr)]
    except RuntimeError as e:
        if "generic" in str(e) or "EUNSUPPORTED" in str(e) or "status -2" in str(e): continue      # the VALU kernels cannot hold this shape
        raise
    tried += 1
    errs = [np.abs(a - b).max() / max(1e-30, np.abs(b).max()) for a, b in zip(out["lmm"], out["gen"])]
    ok = all(np.isfinite(a).all() for a in out["lmm"]) and max(errs) < 2e-5
    if max(errs) > worst[0]: worst = (max(errs), "d=%d c=%d lat=%d hidden=%s %s n=%d" % (d, c, lat, hidden, act, n))
    if not ok:
        bad += 1; print("MISMATCH d=%d c=%d lat=%d hidden=%s %s n=%d gather=%s errs %s" % (d, c, lat, hidden, act, n, idx is not None, errs), flush=True)
print("%d cases compared, %d mismatches; worst relative difference %.2e on %s" % (tried, bad, worst[0], worst[1]))
sys.exit(1 if bad else 0)
