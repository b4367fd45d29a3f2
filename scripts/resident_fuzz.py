"""fuzz: random resident-eligible flow and CVAE shapes, one epoch through the resident kernel and through the batch-by-batch path
(family pinned), same data: per-batch losses and final parameters must agree to rounding.  Prints the worst cases; exit code 1 on
a mismatch."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probaforms_amd import _hip
rng = np.random.default_rng(int(os.environ.get("SEED", 0)))
N = int(os.environ.get("CASES", 150))
bad = 0; worst = (0.0, None); tried = 0; skipped = 0
def dev(a, dt=torch.float32): return torch.as_tensor(np.ascontiguousarray(a)).to(dt).cuda()
while tried < N:
    kind = rng.choice(["flow1", "flow1", "deep", "cvae"])
    d = int(rng.integers(1, 17)); c = int(rng.integers(0, min(16, 31 - d) + 1)); L = int(rng.integers(1, 9))
    act = str(rng.choice(["tanh", "relu"])); batch = int(rng.choice([1, 3, 16, 17, 32, 33, 64, 100, 128])); n = int(batch * rng.integers(1, 4) + rng.integers(0, batch))
    n = max(n, 1)
    if kind == "flow1": hidden = (int(rng.integers(1, 33)),)
    elif kind == "deep": hidden = tuple(int(v) for v in rng.integers(1, 33, size=int(rng.integers(2, 4))))
    else: hidden = (int(rng.integers(1, 33)),)
    x = dev(rng.standard_normal((n, d))); cc = dev(rng.standard_normal((n, c))) if c else None
    perm = torch.from_numpy(rng.permutation(n).astype(np.int64)).cuda()
    nb = (n + batch - 1) // batch
    adam = (2e-3, 0.9, 0.999, 1e-8, float(rng.choice([0.0, 0.1])))
    res = {}
    if kind == "cvae":
        lat = int(rng.integers(1, 9))
        if lat + c > 31: continue
        mk = lambda fam: _hip.CvaeShape.make(d, c, lat, hidden, act, family=fam)
        if not _hip.cvae_fit_epoch_resident(mk("auto"), batch): skipped += 1; continue
        P = _hip.cvae_param_count(mk("auto")); p0 = dev(rng.uniform(-1, 1, P) * 0.4); eps = dev(rng.standard_normal((n, lat)))
        for fam in ("auto", "generic"):
            sh = mk(fam); p = p0.clone(); m = torch.zeros(P, device="cuda"); v = torch.zeros(P, device="cuda")
            hist = torch.empty(nb, device="cuda"); g = torch.empty(P, device="cuda")
            ws = torch.empty(_hip.cvae_workspace_bytes(sh, batch), dtype=torch.uint8, device="cuda")
            _hip.cvae_fit_epoch(sh, p, x, cc, perm, eps, n, batch, 0.05, g, hist, m, v, *adam, 1, ws)
            res[fam] = (hist.cpu().numpy().astype(np.float64), p.cpu().numpy().astype(np.float64))
        a, b = res["auto"], res["generic"]; desc = "cvae d=%d c=%d lat=%d hidden=%s %s n=%d batch=%d" % (d, c, lat, hidden, act, n, batch)
    else:
        user = bool(rng.integers(0, 2)) or kind == "deep"
        masks = (rng.random((L, d)) < 0.5).astype(np.uint8) if user else ((np.arange(d)[None] + np.arange(L)[:, None]) % 2).astype(np.uint8)
        alt = 0 if user else 1
        mk = lambda fam: _hip.RnvpShape.make(L, d, c, hidden, act, alt_masks=alt, family=fam)
        if not _hip.fit_epoch_resident(mk("auto"), batch): skipped += 1; continue
        P = _hip.param_count(mk("auto")); p0 = dev(rng.uniform(-1, 1, P) * 0.3); mkd = dev(masks, torch.uint8)
        for fam in ("auto", "valu"):
            sh = mk(fam); p = p0.clone(); m = torch.zeros(P, device="cuda"); v = torch.zeros(P, device="cuda")
            hist = torch.empty(nb, device="cuda"); g = torch.empty(P, device="cuda")
            ws = torch.empty(max(16, _hip.workspace_bytes(sh, _hip.OP_TRAIN, batch)), dtype=torch.uint8, device="cuda")
            _hip.fit_epoch(sh, p, mkd, x, cc, perm, n, batch, g, hist, m, v, *adam, 1, ws)
            res[fam] = (hist.cpu().numpy().astype(np.float64), p.cpu().numpy().astype(np.float64))
        a, b = res["auto"], res["valu"]; desc = "flow L=%d d=%d c=%d hidden=%s %s n=%d batch=%d user_masks=%s" % (L, d, c, hidden, act, n, batch, user)
    tried += 1
    big = max(np.nanmax(np.abs(np.where(np.isfinite(a[0]), a[0], 0))), np.nanmax(np.abs(np.where(np.isfinite(b[0]), b[0], 0))))
    if big > 1e8 or not (np.isfinite(a[0]).all() and np.isfinite(b[0]).all()):
        # an exploding random model (relu: exp(s) with s in the tens; batch losses of 1e8 .. 1e38, overflow next door): a rounding of s
        # is a relative error of the loss, so the two summation orders part ways -- counted, not compared
        exploding = globals().get("exploding", 0) + 1; globals()["exploding"] = exploding
        continue
    if False:
        pass
    else:
        err = max(np.abs(a[0] - b[0]).max() / max(1.0, np.abs(b[0]).max()), np.abs(a[1] - b[1]).max())
        ok = np.abs(a[0] - b[0]).max() <= 5e-5 * max(1.0, np.abs(b[0]).max()) and np.abs(a[1] - b[1]).mean() < 5e-6 and np.abs(a[1] - b[1]).max() < 5e-4
        if err > worst[0]: worst = (err, desc)
    if not ok:
        bad += 1; print("MISMATCH", desc, "loss err", np.abs(a[0] - b[0]).max(), "param err max", np.abs(a[1] - b[1]).max(), flush=True)
print("%d cases compared (%d more were not resident-eligible, %d exploding models set aside), %d mismatches; worst agreement %.2e on %s"
      % (tried - globals().get("exploding", 0), skipped, globals().get("exploding", 0), bad, worst[0], worst[1]))
sys.exit(1 if bad else 0)
