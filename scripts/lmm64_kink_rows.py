"""Why relu cases of scripts/lmm64_fuzz.py differ by more than rounding between the two any-shape training kernels: replays the fuzz
stream of a seed, takes its worst relu case and compares the kernels ROW BY ROW (one-row calls): the batch difference is the sum of
the differences of the few rows where a pre-activation lands on the other side of 0.   python scripts/lmm64_kink_rows.py <seed> <cases>"""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from probaforms_amd import _hip
# replay scripts/lmm64_fuzz.py's stream for a seed and stop at the worst relu case
seed = int(sys.argv[1]); cases = int(sys.argv[2])
rng = np.random.default_rng(seed)
worst = (0, None)
def run(sh, p, mk, x, cc, n, invB):
    P = p.numel(); g = torch.empty(P + 1, device="cuda")
    ws = torch.empty(max(_hip.workspace_bytes(sh, _hip.OP_TRAIN, n), 16), dtype=torch.uint8, device="cuda")
    _hip.loss_grad(sh, p, mk, x, cc, None, n, invB, g[:P], g[P:], ws)
    return g[:P].clone()
for it in range(cases):
    L = int(rng.integers(1, 7)); d = int(rng.choice([1, 2, 3, 5, 8, 13, 16, 17, 24, 32, 33, 40, 64, 70])); c = int(rng.choice([0, 0, 1, 3, 4, 7, 16, 19]))
    nh = int(rng.integers(1, 4))
    hidden = tuple(int(rng.choice([1, 3, 10, 16, 17, 31, 32, 48, 64, 65, 100, 128])) for _ in range(nh))
    act = "tanh" if rng.integers(0, 2) else "relu"
    n = int(rng.choice([1, 15, 63, 64, 65, 129, 500, 1000, 4097, 9000]))
    masks = rng.integers(0, 2, (L, d)).astype(np.uint8)
    sh64 = _hip.RnvpShape.make(L, d, c, hidden, act, alt_masks=0, family="lmm64"); sh16 = _hip.RnvpShape.make(L, d, c, hidden, act, alt_masks=0, family="lmm16")
    if _hip.kernel_path(sh16, masks, _hip.OP_TRAIN) != _hip.PATH_LMM: continue
    P = _hip.param_count(sh64)
    p = torch.as_tensor((rng.uniform(-1, 1, P) * min(0.3, 1.5 / np.sqrt(max(hidden) + d + c))).astype(np.float32)).cuda()
    x = torch.as_tensor(rng.standard_normal((n, d)).astype(np.float32)).cuda()
    cc = torch.as_tensor(rng.standard_normal((n, c)).astype(np.float32)).cuda() if c else None
    mk = torch.as_tensor(masks).cuda()
    g64 = run(sh64, p, mk, x, cc, n, 1.0 / n)
    if _hip.last_dispatch(_hip.PROFILE_TRAIN)["kernel"] != "k_lmm_train64": continue
    g16 = run(sh16, p, mk, x, cc, n, 1.0 / n)
    err = ((g64 - g16).abs().max() / (g16.abs().max() + 1e-12)).item()
    if act == "relu" and err > worst[0]: worst = (err, (L, d, c, hidden, n, sh64, sh16, p, mk, x, cc, g64, g16))
err, (L, d, c, hidden, n, sh64, sh16, p, mk, x, cc, g64, g16) = worst
print("worst relu case: L=%d d=%d c=%d hidden=%s n=%d err %.3e" % (L, d, c, hidden, n, err))
scale = g16.abs().max().item()
tot = torch.zeros_like(g64); bad = []
for r in range(n):
    xr = x[r:r + 1].contiguous(); cr = cc[r:r + 1].contiguous() if cc is not None else None
    a = run(sh64, p, mk, xr, cr, 1, 1.0 / n); b = run(sh16, p, mk, xr, cr, 1, 1.0 / n)
    dd = (a - b)
    if dd.abs().max().item() > 1e-7 * scale: bad.append((r, dd.abs().max().item() / scale, int((dd.abs() > 1e-7 * scale).sum())))
    tot += dd
print("rows whose single-row gradients differ between the two kernels: %d of %d: %s" % (len(bad), n, bad[:6]))
print("sum of the single-row differences vs the batch difference: max |.| / scale = %.3e (batch difference %.3e)" % ((tot - (g64 - g16)).abs().max().item() / scale, err))
