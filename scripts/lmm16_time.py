import sys, os, numpy as np, torch
sys.path.insert(0, "/root/repo")
from probaforms_amd import _hip
for hidden, L, d, c, n in (((128, 128), 8, 16, 4, 262144), ((64, 64), 8, 16, 4, 262144), ((10, 20, 15), 8, 2, 0, 262144), ((128, 128), 8, 16, 4, 4096)):
    rng = np.random.default_rng(0)
    masks = torch.as_tensor(((np.arange(d)[None] + np.arange(L)[:, None]) % 2).astype(np.uint8)).cuda()
    sh = _hip.RnvpShape.make(L, d, c, hidden, "tanh", alt_masks=0, family="lmm16")
    P = _hip.param_count(sh)
    p = torch.as_tensor((rng.uniform(-1, 1, P) * 0.1).astype(np.float32)).cuda()
    x = torch.randn(n, d, device="cuda"); cc = torch.randn(n, c, device="cuda") if c else None
    z = torch.empty(n, d, device="cuda"); lp = torch.empty(n, device="cuda"); xb = torch.empty(n, d, device="cuda"); g = torch.empty(P + 1, device="cuda")
    wsf = torch.empty(_hip.workspace_bytes(sh, _hip.OP_FORWARD, n), dtype=torch.uint8, device="cuda")
    wst = torch.empty(_hip.workspace_bytes(sh, _hip.OP_TRAIN, min(n, 65536)), dtype=torch.uint8, device="cuda")
    res = []
    nt = min(n, 65536)
    for op in ("fwd", "inv", "train"):
        def run():
            if op == "fwd": _hip.forward_logprob(sh, p, masks, x, cc, None, n, z, None, lp, None, wsf)
            elif op == "inv": _hip.inverse(sh, p, masks, z, cc, n, xb, wsf)
            else: _hip.loss_grad(sh, p, masks, x, cc, None, nt, 1.0 / nt, g[:P], g[P:], wst)
        for _ in range(2): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): run()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 5)
    print("hidden=%s d=%d rows=%d (16-row kernels): forward %.3f ms, inverse %.3f ms, loss+grad on %d rows %.3f ms" % (hidden, d, n, res[0], res[1], nt, res[2]))
