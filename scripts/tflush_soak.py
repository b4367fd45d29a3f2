"""Soak of the barrier-free gradient flush (RNVP_NS_TFLUSH / RNVP_WIDE_TFLUSH): the same rnvp_loss_grad call repeated, every result
compared bit for bit with the first -- a lost or double-counted slot window (the arrival-count race of profiles/r05_wide_tflush_ab.txt)
or a hang would show here.  GPU box only:  python scripts/tflush_soak.py [repeats]"""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probaforms_amd import _hip

CASES = [("c2 net-split", (8, 16, 4, 128), 65536), ("c2 ragged net-split", (8, 16, 4, 128), 16960), ("c3 net-split", (12, 32, 8, 256), 16960),
         ("c3 net-split 32768", (12, 32, 8, 256), 32768), ("c3 wide", (12, 32, 8, 256), 65536), ("c3 wide, 2 row groups", (12, 32, 8, 256), 131072),
         ("narrow net, one hidden tile", (6, 10, 3, 16), 30000), ("odd tile count", (5, 16, 4, 112), 50000), ("c2 no cond", (8, 16, 0, 128), 40000)]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
bad = 0
for name, (L, d, c, h), n in CASES:
    shape = _hip.RnvpShape.make(L, d, c, (h,), "tanh", alt_masks=1)
    P = _hip.param_count(shape)
    g = torch.Generator(device="cuda").manual_seed(1)
    params = (torch.rand(P, device="cuda", generator=g) - 0.5) * 0.2
    x = torch.randn(n + 1000, d, device="cuda", generator=g); cc = torch.randn(n + 1000, c, device="cuda", generator=g) if c else None
    idx = torch.randperm(n + 1000, device="cuda", generator=g)[:n].contiguous()
    ws = torch.empty(_hip.workspace_bytes(shape, 2, n), dtype=torch.uint8, device="cuda")
    first = None
    t0 = time.time()
    for r in range(reps):
        gb = torch.full((P + 1,), float("nan"), device="cuda")
        _hip.loss_grad(shape, params, None, x, cc, idx, n, 1.0 / n, gb[:P], gb[P:P + 1], ws)
        if first is None:
            first = gb.clone(); disp = _hip.last_dispatch(_hip.PROFILE_TRAIN)
        elif not torch.equal(first, gb):
            bad += 1
            print("  MISMATCH %s launch %d: %d of %d values differ" % (name, r, int((first != gb).sum()), P + 1))
    torch.cuda.synchronize()
    print("%-28s %-18s %-9s grid %3d rows %6d: %d launches, %.2f s, finite %s" % (name, disp["kernel"], disp["variant"], disp["grid"], n, reps,
          time.time() - t0, bool(torch.isfinite(first).all())))
print("mismatches:", bad)
sys.exit(1 if bad else 0)
