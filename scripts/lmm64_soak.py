"""run-to-run bits of the 64-row any-shape training kernel: the same call repeated, every result compared with the first
(a race between the LDS-DMA of the next visit's activations and a wave still reading the region would show here)
python scripts/lmm64_soak.py [repeats]"""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from probaforms_amd import _hip
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(0)
for (L, d, c, hidden, n) in [(8, 16, 4, (128, 128), 65536), (8, 16, 4, (128, 128), 70001), (4, 6, 2, (12, 20), 30000), (3, 80, 20, (24,), 20000),
                             (8, 2, 0, (10, 20, 15), 65536), (6, 16, 4, (64, 64), 262144 + 777)]:
    masks = torch.as_tensor(rng.integers(0, 2, (L, d)).astype(np.uint8)).cuda()
    sh = _hip.RnvpShape.make(L, d, c, hidden, "tanh", alt_masks=0, family="lmm64")
    P = _hip.param_count(sh)
    p = torch.as_tensor((rng.uniform(-1, 1, P) * 0.1).astype(np.float32)).cuda()
    x = torch.randn(n, d, device="cuda"); cc = torch.randn(n, c, device="cuda") if c else None
    ws = torch.empty(_hip.workspace_bytes(sh, _hip.OP_TRAIN, n), dtype=torch.uint8, device="cuda")
    first = None; bad = 0
    for it in range(reps):
        g = torch.empty(P + 1, device="cuda")
        ws.random_(0, 255)                      # the workspace's previous contents must not matter
        _hip.loss_grad(sh, p, masks, x, cc, None, n, 1.0 / n, g[:P], g[P:], ws)
        if first is None: first = g.clone()
        elif not torch.equal(first, g): bad += 1
    assert _hip.last_dispatch(_hip.PROFILE_TRAIN)["kernel"] == "k_lmm_train64"
    print("L=%d d=%d c=%d hidden=%s rows=%d: %d repeats, %d differ from the first, all finite %s" % (L, d, c, hidden, n, reps, bad, bool(torch.isfinite(first).all())))
    assert bad == 0
