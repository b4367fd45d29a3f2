"""soak: 100 epochs of the README example's shape through the resident fit and through the batch-by-batch loop (family pinned),
same data, same permutations: the loss curves must stay together (they agree step by step to rounding; rounding differences
grow slowly with the step count)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probaforms_amd import _hip
f = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "moons_fit.npz"))
X = torch.tensor(f["X"], dtype=torch.float32).cuda(); C = torch.tensor(f["C"], dtype=torch.float32).cuda()
n, d, c, L, batch, epochs = X.shape[0], X.shape[1], C.shape[1], 8, 32, 100
masks = torch.tensor(((np.arange(d)[None] + np.arange(L)[:, None]) % 2).astype(np.uint8)).cuda()
rng = np.random.default_rng(0)
res = {}
for fam in ("auto", "valu"):
    shape = _hip.RnvpShape.make(L, d, c, (10,), "tanh", alt_masks=1, family=fam)
    P = _hip.param_count(shape)
    p = torch.tensor((np.random.default_rng(1).uniform(-1, 1, P) * 0.3).astype(np.float32)).cuda()
    m = torch.zeros(P, device="cuda"); v = torch.zeros(P, device="cuda")
    ws = torch.empty(_hip.workspace_bytes(shape, _hip.OP_TRAIN, batch), dtype=torch.uint8, device="cuda")
    nb = (n + batch - 1) // batch
    hist = torch.empty(nb, device="cuda"); g = torch.empty(P, device="cuda"); curve = []
    prng = np.random.default_rng(7)
    for ep in range(epochs):
        perm = torch.from_numpy(prng.permutation(n).astype(np.int64)).cuda()
        _hip.fit_epoch(shape, p, masks, X, C, perm, n, batch, g, hist, m, v, 0.01, 0.9, 0.999, 1e-8, 0.0, 1 + ep * nb, ws)
        curve.append(float(hist.mean()))
    res[fam] = (np.array(curve), p.cpu().numpy(), _hip.fit_epoch_resident(shape, batch))
a, b = res["auto"], res["valu"]
print("resident:", a[2], "loop:", b[2])
for ep in (0, 1, 9, 49, 99):
    print("epoch %3d mean batch loss: resident %.5f  loop %.5f  (diff %.2e)" % (ep + 1, a[0][ep], b[0][ep], abs(a[0][ep] - b[0][ep])))
print("max |param diff| after %d steps: %.3e; all finite: %s" % (epochs * ((n + batch - 1) // batch), np.abs(a[1] - b[1]).max(), np.isfinite(a[1]).all()))
