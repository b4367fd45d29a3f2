"""us per training step of rnvp_fit_epoch on small flows at small batches: the persistent one-workgroup epoch
(rnvp_resident.hip) vs the batch-by-batch rnvp_train_step loop (three launches per step)"""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probaforms_amd import _hip

def run(L, d, c, hidden, act, batch, nb=256):
    n = nb * batch
    masks = torch.tensor(((np.arange(d)[None] + np.arange(L)[:, None]) % 2).astype(np.uint8)).cuda()
    shape = _hip.RnvpShape.make(L, d, c, hidden, act, alt_masks=1 if len(hidden) == 1 else 0)
    P = _hip.param_count(shape)
    g = torch.Generator(device="cuda").manual_seed(0)
    p0 = (torch.rand(P, device="cuda", generator=g) - 0.5) * 0.4
    x = torch.randn(n, d, device="cuda", generator=g); cc = torch.randn(n, c, device="cuda", generator=g) if c else None
    perm = torch.randperm(n, device="cuda", generator=g)
    ws = torch.empty(_hip.workspace_bytes(shape, _hip.OP_TRAIN, batch), dtype=torch.uint8, device="cuda")
    hist = torch.empty(nb, device="cuda"); gbuf = torch.empty(P, device="cuda")
    res = {}
    for mode in ("fit_epoch", "step_loop"):
        p = p0.clone(); m = torch.zeros(P, device="cuda"); v = torch.zeros(P, device="cuda")
        def epoch(first):
            if mode == "fit_epoch":
                _hip.fit_epoch(shape, p, masks, x, cc, perm, n, batch, gbuf, hist, m, v, 1e-3, 0.9, 0.999, 1e-8, 0.0, first, ws)
            else:
                for k in range(nb):
                    _hip.train_step(shape, p, masks, x, cc, perm[k * batch:(k + 1) * batch], batch, 1.0 / batch, gbuf, hist[k:k + 1], m, v,
                                    1e-3, 0.9, 0.999, 1e-8, 0.0, first + k, ws)
        epoch(1); torch.cuda.synchronize()
        t0 = time.perf_counter(); epoch(1 + nb); epoch(1 + 2 * nb); torch.cuda.synchronize()
        res[mode] = (time.perf_counter() - t0) / (2 * nb) * 1e6
        res[mode + "_loss"] = float(hist[-1])
    print("resident=%d L=%d d=%d c=%d hidden=%s %s batch=%d P=%d: fit_epoch %.1f us/step, train_step loop %.1f us/step (last loss %.4f / %.4f)" % (
        int(_hip.fit_epoch_resident(shape, batch)), L, d, c, hidden, act, batch, P, res["fit_epoch"], res["step_loop"], res["fit_epoch_loss"], res["step_loop_loss"]), flush=True)

for cfg in [(8, 2, 1, (10,), "tanh", 32), (8, 2, 1, (10,), "relu", 32), (8, 2, 1, (10,), "tanh", 8), (8, 2, 1, (10,), "tanh", 64), (8, 2, 1, (16,), "tanh", 32),
            (8, 5, 3, (10,), "tanh", 32), (8, 8, 7, (16,), "tanh", 32), (8, 8, 8, (16,), "tanh", 32), (8, 2, 1, (32,), "tanh", 32), (8, 2, 1, (32,), "tanh", 64),
            (8, 2, 1, (64,), "tanh", 32), (4, 2, 1, (64,), "tanh", 64), (8, 16, 4, (10,), "tanh", 32), (4, 16, 4, (32,), "tanh", 32), (16, 2, 1, (10,), "tanh", 32),
            (8, 2, 1, (10, 10), "tanh", 32), (8, 2, 1, (10, 10, 10), "tanh", 32), (8, 5, 3, (16, 16), "relu", 32), (8, 2, 1, (10, 20, 15), "tanh", 32)]:
    run(*cfg)
