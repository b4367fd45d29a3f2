import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probaforms_amd.models import RealNVP
n = int(os.environ.get("N", 16_000_000)); d, c = 64, 16
rng = np.random.default_rng(0)
Xs = rng.standard_normal((4096, d)).astype(np.float32); Cs = rng.standard_normal((4096, c)).astype(np.float32)
m2 = RealNVP(n_layers=8, hidden=(128,), batch_size=4096, n_epochs=1, lr=1e-3, prior_rng="device"); m2.fit(Xs, Cs)
C = rng.standard_normal((n, c), dtype=np.float32); Ch = torch.from_numpy(C)
eng = m2.nf.engine(); dev = eng.device
rows = m2.nf.pipelined_rows(n)
acc = {}
def tick(k, t0):
    t = time.perf_counter(); acc[k] = acc.get(k, 0) + t - t0; return t
for rep in range(2):
    acc.clear(); evs = []; T0 = time.perf_counter()
    t = time.perf_counter()
    out = torch.empty((n, d), dtype=torch.float32, pin_memory=True); t = tick("alloc out", t)
    zdev = [torch.empty((rows + 15, d), device=dev) for _ in range(2)]
    cdev = [torch.empty((rows + 15, c), device=dev) for _ in range(2)]
    cpin = [torch.empty((rows + 15, c), pin_memory=True) for _ in range(2)]; t = tick("alloc bufs", t)
    cur = torch.cuda.current_stream(dev); h2d, d2h = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    ev_in = [torch.cuda.Event() for _ in range(2)]; ev_k = [torch.cuda.Event() for _ in range(2)]; ev_out = [torch.cuda.Event() for _ in range(2)]
    for k, lo in enumerate(range(0, n, rows)):
        i, m = k % 2, min(rows, n - lo)
        t = time.perf_counter()
        if k >= 2: ev_out[i].synchronize()
        t = tick("wait", t)
        np.copyto(cpin[i][:m].numpy(), C[lo:lo + m]); t = tick("host copy c", t)
        with torch.cuda.stream(h2d):
            cdev[i][:m].copy_(cpin[i][:m], non_blocking=True); ev_in[i].record(h2d)
        t = tick("h2d issue", t)
        cur.wait_event(ev_in[i]); zdev[i][:m].normal_(); t = tick("normal", t)
        eng.inverse(zdev[i][:m], cdev[i][:m], out=zdev[i][:m]); ev_k[i].record(cur); t = tick("inverse issue", t)
        d2h.wait_event(ev_k[i])
        with torch.cuda.stream(d2h):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record(d2h)
            out[lo:lo + m].copy_(zdev[i][:m], non_blocking=True); ev_out[i].record(d2h); e1.record(d2h); evs.append((e0, e1))
        t = tick("d2h issue", t)
    d2h.synchronize(); t = tick("final sync", t)
    print("total %.1f ms" % ((time.perf_counter() - T0) * 1e3), {k: round(v * 1e3, 1) for k, v in acc.items()}, flush=True)
    print("d2h device time total %.1f ms" % sum(a.elapsed_time(b) for a, b in evs))
    del out
