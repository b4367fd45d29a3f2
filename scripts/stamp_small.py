"""dev aid: phase cycle stamps (library built with -DRNVP_STAMP) of one small training step"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probaforms_amd import _hip
d, c, h, L, n = [int(v) for v in os.environ.get("SHAPE", "2,1,10,8,32").split(",")]
shp = _hip.RnvpShape.make(L, d, c, (h,), "tanh", 1)
P = _hip.param_count(shp)
rng = np.random.default_rng(0)
p = torch.tensor(rng.standard_normal(P) * 0.1, dtype=torch.float32).cuda()
x = torch.randn(n, d, device="cuda"); cc = torch.randn(n, c, device="cuda") if c else None
masks = torch.tensor([[(j + l) % 2 for j in range(d)] for l in range(L)], dtype=torch.uint8).cuda()
ws = torch.empty(_hip.workspace_bytes(shp, _hip.OP_TRAIN, n), dtype=torch.uint8, device="cuda")
g = torch.empty(P, device="cuda"); loss = torch.empty(1, device="cuda")
for i in range(3):
    _hip.loss_grad(shp, p, masks, x, cc, None, n, 1.0 / n, g, loss, ws); torch.cuda.synchronize()
    print("---", flush=True)
