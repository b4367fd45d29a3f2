"""dev aid: rnvp_fit_epoch (the call RealNVP.fit makes per epoch) on random data, for per-kernel timing under rocprofv3.
   CASES="d,c,h,L,n,batch;..." python scripts/step_profile.py      (default: C2 at batch 32, C2 and C3 at batch 65536)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probaforms_amd import _hip
for case in os.environ.get("CASES", "16,4,128,8,16384,32;16,4,128,8,1000000,65536;32,8,256,12,262144,65536").split(";"):
    d, c, h, L, n, batch = [int(v) for v in case.split(",")]
    shp = _hip.RnvpShape.make(L, d, c, (h,), "tanh", 1)
    P = _hip.param_count(shp)
    gen = torch.Generator(device="cuda").manual_seed(0)
    p = (torch.rand(P, device="cuda", generator=gen) - 0.5) * 0.2
    x = torch.randn(n, d, device="cuda", generator=gen); cc = torch.randn(n, c, device="cuda", generator=gen)
    ws = torch.empty(_hip.workspace_bytes(shp, _hip.OP_TRAIN, batch), dtype=torch.uint8, device="cuda")
    g = torch.empty(P + 1, device="cuda"); m = torch.zeros(P, device="cuda"); v = torch.zeros(P, device="cuda")
    nb = (n + batch - 1) // batch
    hist = torch.zeros(nb, device="cuda")
    reps = int(os.environ.get("REPS", "4"))
    step = 1
    for r in range(reps + 1):
        perm = torch.randperm(n, device="cuda", generator=gen)
        if r == 1:
            torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
        _hip.fit_epoch(shp, p, None, x, cc, perm, n, batch, g[:P], hist, m, v, 1e-3, 0.9, 0.999, 1e-8, 0.0, step, ws)
        step += nb
    e1.record(); torch.cuda.synchronize()
    print("d=%d c=%d h=%d L=%d n=%d batch=%d: %.2f us per step (%d steps per epoch), last dispatch %s, loss %.4f"
          % (d, c, h, L, n, batch, e0.elapsed_time(e1) / (reps * nb) * 1e3, nb, _hip.last_dispatch(_hip.PROFILE_TRAIN), float(hist[-1])))
