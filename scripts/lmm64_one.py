"""a few rnvp_loss_grad calls on the 64-row any-shape kernel (for profilers): python scripts/lmm64_one.py [h1,h2] [rows] [family]"""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from probaforms_amd import _hip
hidden = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "128,128").split(","))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
fam = sys.argv[3] if len(sys.argv) > 3 else "lmm64"
L, d, c = 8, 16, 4
rng = np.random.default_rng(0)
masks = torch.as_tensor(((np.arange(d)[None] + np.arange(L)[:, None]) % 2).astype(np.uint8)).cuda()
x = torch.randn(n, d, device="cuda"); cc = torch.randn(n, c, device="cuda")
sh = _hip.RnvpShape.make(L, d, c, hidden, "tanh", alt_masks=0, family=fam)
P = _hip.param_count(sh)
p = torch.as_tensor((rng.uniform(-1, 1, P) * 0.1).astype(np.float32)).cuda()
g = torch.empty(P + 1, device="cuda")
ws = torch.empty(_hip.workspace_bytes(sh, _hip.OP_TRAIN, n), dtype=torch.uint8, device="cuda")
for _ in range(4):
    _hip.loss_grad(sh, p, masks, x, cc, None, n, 1.0 / n, g[:P], g[P:], ws)
torch.cuda.synchronize()
print(g[P].item())
