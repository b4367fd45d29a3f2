"""race / determinism soak: the same training call repeated must give bit-identical gradients (all wave modes)"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probaforms_amd import _hip
for (L, d, c, h), sizes in (((8, 16, 4, 128), (1000, 8192, 32768, 65536, 131072)), ((12, 32, 8, 256), (4096, 65536)), ((8, 64, 16, 128), (4096, 65536)), ((8, 2, 1, 10), (32, 1000))):
    for act in ("tanh", "relu"):
        shp = _hip.RnvpShape.make(L, d, c, (h,), act, 1)
        P = _hip.param_count(shp)
        g = torch.Generator(device="cuda").manual_seed(1)
        p = (torch.rand(P, device="cuda", generator=g) - 0.5) * 0.2
        masks = torch.tensor(((np.arange(d)[None] + np.arange(L)[:, None]) % 2).astype(np.uint8)).cuda()
        for n in sizes:
            x = torch.randn(n, d, device="cuda", generator=g); cc = torch.randn(n, c, device="cuda", generator=g)
            ws = torch.empty(_hip.workspace_bytes(shp, _hip.OP_TRAIN, n), dtype=torch.uint8, device="cuda")
            ref = torch.empty(P + 1, device="cuda"); cur = torch.empty(P + 1, device="cuda")
            _hip.loss_grad(shp, p, masks, x, cc, None, n, 1.0 / n, ref[:P], ref[P:], ws)
            bad = 0
            for it in range(int(os.environ.get("ITERS", 60))):
                cur.fill_(float("nan"))
                _hip.loss_grad(shp, p, masks, x, cc, None, n, 1.0 / n, cur[:P], cur[P:], ws)
                bad += int(not torch.equal(ref, cur))
            print((L, d, c, h), act, n, "mismatching repeats:", bad, "finite:", bool(torch.isfinite(ref).all()), flush=True)
