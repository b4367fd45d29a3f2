"""phase stamps of the resident fit kernel (library built with EXTRA=-DRC_STAMP): C1 defaults, batch 32"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probaforms_amd import _hip
L, d, c, hidden, batch, nb = [eval(v) for v in os.environ.get("SHAPE", "8;2;1;(10,);32;32").split(";")]
n = nb * batch
masks = torch.tensor(((np.arange(d)[None] + np.arange(L)[:, None]) % 2).astype(np.uint8)).cuda()
shape = _hip.RnvpShape.make(L, d, c, hidden, "tanh", alt_masks=1)
P = _hip.param_count(shape)
p = (torch.rand(P, device="cuda") - 0.5) * 0.4; m = torch.zeros(P, device="cuda"); v = torch.zeros(P, device="cuda")
x = torch.randn(n, d, device="cuda"); cc = torch.randn(n, c, device="cuda"); perm = torch.randperm(n, device="cuda")
ws = torch.empty(_hip.workspace_bytes(shape, _hip.OP_TRAIN, batch), dtype=torch.uint8, device="cuda")
hist = torch.empty(nb, device="cuda"); gbuf = torch.empty(P, device="cuda")
_hip.fit_epoch(shape, p, masks, x, cc, perm, n, batch, gbuf, hist, m, v, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, ws)
torch.cuda.synchronize()
