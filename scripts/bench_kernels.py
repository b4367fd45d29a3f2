"""Kernel-level timing of the C-ABI entry points (HIP events), per config.  GPU box only."""
import sys, os, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probaforms_amd import _hip

CONFIGS = {"c2": (8, 16, 4, 128), "c3": (12, 32, 8, 256), "c4": (8, 64, 16, 128), "c2_nocond": (8, 16, 0, 128)}


def timeit(fn, iters=10, warm=3):
    # warm-up: at least `warm` calls AND at least ~50 ms of work -- a handful of launches from an idle chip run under its clock ramp
    # (round 6: the C3 training kernel read 1.25 ms cold and 1.13-1.15 ms warm)
    import time
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < float(os.environ.get("WARM_S", 0.05)):
        for _ in range(max(1, iters // 2)): fn()
        torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    which = sys.argv[1:] or list(CONFIGS)
    ops = os.environ.get("OPS", "fwd,inv,train").split(",")
    for name in which:
        L, d, c, h = CONFIGS[name] if name in CONFIGS else tuple(int(v) for v in name.split(","))      # or "L,d,c,h"
        n = int(os.environ.get("N", 1 << 20))
        shape = _hip.RnvpShape.make(L, d, c, (h,), os.environ.get("ACT", "tanh"), alt_masks=1, precision=os.environ.get("PREC") or None,
                                    small_calls=_hip.SMALL_CALLS[os.environ.get("SMALL", "invariant")])
        P = _hip.param_count(shape)
        g = torch.Generator(device="cuda").manual_seed(0)
        params = (torch.rand(P, device="cuda", generator=g) - 0.5) * 0.2
        masks = torch.tensor(((np.arange(d)[None] + np.arange(L)[:, None]) % 2).astype(np.uint8)).cuda()
        x = torch.randn(n, d, device="cuda", generator=g)
        cc = torch.randn(n, c, device="cuda", generator=g) if c else None
        z = torch.empty_like(x); lp = torch.empty(n, device="cuda"); tot = torch.empty(1, device="cuda")
        useful = 4 * h * (d + c) * L
        res = {"config": name, "n": n}
        if "fwd" in ops:
            ws = torch.empty(_hip.workspace_bytes(shape, 0, n), dtype=torch.uint8, device="cuda")
            ms = timeit(lambda: _hip.forward_logprob(shape, params, masks, x, cc, None, n, z, None, lp, tot, ws))
            res["fwd_ms"] = ms; res["fwd_TF_useful"] = useful * n / ms / 1e9; res["fwd_Mrows_s"] = n / ms / 1e3
        if "inv" in ops:
            ws = torch.empty(_hip.workspace_bytes(shape, 1, n), dtype=torch.uint8, device="cuda")
            ms = timeit(lambda: _hip.inverse(shape, params, masks, x, cc, n, z, ws))
            res["inv_ms"] = ms; res["inv_TF_useful"] = useful * n / ms / 1e9; res["inv_Mrows_s"] = n / ms / 1e3
        if "train" in ops:
            nt = int(os.environ.get("NT", 65536))
            ws = torch.empty(_hip.workspace_bytes(shape, 2, nt), dtype=torch.uint8, device="cuda")
            gb = torch.empty(P + 4, device="cuda")
            idx = torch.randperm(n, device="cuda")[:nt].contiguous()
            ms = timeit(lambda: _hip.loss_grad(shape, params, masks, x, cc, idx, nt, 1.0 / nt, gb[:P], gb[P:P + 1], ws), iters=int(os.environ.get("ITERS", 5)), warm=2)
            res["train_ms"] = ms; res["train_TF_useful"] = 3 * useful * nt / ms / 1e9; res["train_Mrows_s"] = nt / ms / 1e3
            res["train_path"] = _hip.kernel_path(shape, None, 2)
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
