import sys, os, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/tests/golden")
from conftest import load_case
from probaforms_amd import _hip
from oracle import Oracle, Shape
o64 = Oracle(64); o32 = Oracle(32)
for name in ("c2", "c3", "c4"):
    cs = load_case(name)
    rng = np.random.default_rng(0)
    n = 4096
    X = rng.standard_normal((n, cs["d"])).astype(np.float32); C = rng.standard_normal((n, cs["c"])).astype(np.float32)
    s = Shape.make(cs["L"], cs["d"], cs["c"], cs["hidden"], cs["act"])
    z64, lp64, _ = o64.log_prob(s, cs["params"], X, C, cs["masks"])
    z32, lp32, _ = o32.log_prob(s, cs["params"], X, C, cs["masks"])
    for prec in ("f32", "bx3"):
        shape = _hip.RnvpShape.make(cs["L"], cs["d"], cs["c"], cs["hidden"], cs["act"], alt_masks=1, precision=prec)
        p = torch.from_numpy(cs["params"]).cuda(); x = torch.from_numpy(X).cuda(); c = torch.from_numpy(C).cuda()
        z = torch.empty_like(x); lp = torch.empty(n, device="cuda")
        ws = torch.empty(_hip.workspace_bytes(shape, 0, n), dtype=torch.uint8, device="cuda")
        _hip.forward_logprob(shape, p, None, x, c, None, n, z, None, lp, None, ws)
        lpn = lp.cpu().numpy(); zn = z.cpu().numpy()
        print(name, prec, "logp MAE vs f64 %.2e (oracle f32: %.2e)  z MAE vs f64 %.2e (oracle f32 %.2e)  max|logp| %.0f" %
              (np.abs(lpn - lp64).mean(), np.abs(lp32 - lp64).mean(), np.abs(zn - z64).mean(), np.abs(z32 - z64).mean(), np.abs(lp64).max()))
