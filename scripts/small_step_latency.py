"""dev aid: GPU time of one small training step (loss+grad kernels only), averaged over back-to-back launches"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probaforms_amd import _hip
for shape in os.environ.get("SHAPES", "2,1,10,8,32;5,3,10,8,32;16,4,128,8,32;16,4,128,8,256").split(";"):
    d, c, h, L, n = [int(v) for v in shape.split(",")]
    shp = _hip.RnvpShape.make(L, d, c, (h,), "tanh", 1)
    P = _hip.param_count(shp)
    rng = np.random.default_rng(0)
    p = torch.tensor(rng.standard_normal(P) * 0.1, dtype=torch.float32).cuda()
    x = torch.randn(n, d, device="cuda"); cc = torch.randn(n, c, device="cuda") if c else None
    masks = torch.tensor([[(j + l) % 2 for j in range(d)] for l in range(L)], dtype=torch.uint8).cuda()
    ws = torch.empty(_hip.workspace_bytes(shp, _hip.OP_TRAIN, n), dtype=torch.uint8, device="cuda")
    g = torch.empty(P + 1, device="cuda")
    m = torch.zeros(P, device="cuda"); v = torch.zeros(P, device="cuda")
    loss = torch.empty(1, device="cuda")
    for _ in range(20):
        _hip.train_step(shp, p, masks, x, cc, None, n, 1.0 / n, g, loss, m, v, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    K = 500
    e0.record()
    for i in range(K):
        _hip.train_step(shp, p, masks, x, cc, None, n, 1.0 / n, g, loss, m, v, 1e-3, 0.9, 0.999, 1e-8, 0.0, i + 2, ws)
    e1.record(); torch.cuda.synchronize()
    print("d=%d c=%d h=%d L=%d rows=%d: %.1f us per train step" % (d, c, h, L, n, e0.elapsed_time(e1) / K * 1e3))
