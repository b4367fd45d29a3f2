"""whole rnvp_loss_grad call and its chain kernel on the any-shape training kernels, 16-row and 64-row forms (ms):
python scripts/lmm64_time.py [h1,h2,..] [rows] [L] [d] [c]"""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from probaforms_amd import _hip
hidden = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "128,128").split(","))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
L = int(sys.argv[3]) if len(sys.argv) > 3 else 8
d = int(sys.argv[4]) if len(sys.argv) > 4 else 16
c = int(sys.argv[5]) if len(sys.argv) > 5 else 4
rng = np.random.default_rng(0)
masks = torch.as_tensor(((np.arange(d)[None] + np.arange(L)[:, None]) % 2).astype(np.uint8)).cuda()
x = torch.randn(n, d, device="cuda"); cc = torch.randn(n, c, device="cuda") if c else None
for fam in ("lmm16", "lmm64", "auto"):
    sh = _hip.RnvpShape.make(L, d, c, hidden, "tanh", alt_masks=0, family=fam)
    P = _hip.param_count(sh)
    p = torch.as_tensor((rng.uniform(-1, 1, P) * 0.1).astype(np.float32)).cuda()
    g = torch.empty(P + 1, device="cuda")
    ws = torch.empty(_hip.workspace_bytes(sh, _hip.OP_TRAIN, n), dtype=torch.uint8, device="cuda")
    for _ in range(3):
        _hip.loss_grad(sh, p, masks, x, cc, None, n, 1.0 / n, g[:P], g[P:], ws)
    torch.cuda.synchronize()
    _hip.profile_enable(64)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        _hip.loss_grad(sh, p, masks, x, cc, None, n, 1.0 / n, g[:P], g[P:], ws)
    e1.record(); torch.cuda.synchronize()
    cnt, ms = _hip.profile_read(_hip.PROFILE_TRAIN)
    _hip.profile_enable(0)
    disp = _hip.last_dispatch(_hip.PROFILE_TRAIN)
    flops = 0
    nin = d + c
    for h in hidden + (d,):
        flops += 2 * nin * h; nin = h
    useful = 3 * 2 * L * flops * n
    print("hidden=%s L=%d d=%d c=%d rows=%d family=%-5s: call %.3f ms, chain kernel %.3f ms (%d launches timed), %s, useful %.1f TF on the call = %.3f of the f32 MFMA peak, loss %.5f"
          % (hidden, L, d, c, n, fam, e0.elapsed_time(e1) / reps, ms / max(cnt, 1), cnt, disp["kernel"], useful / (e0.elapsed_time(e1) / reps * 1e-3) / 1e12,
             useful / (e0.elapsed_time(e1) / reps * 1e-3) / 157.3e12, g[P].item()))
