"""time cvae_loss_grad on a shape outside the register-chained MFMA path: any-shape MFMA kernels vs one thread per row"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from probaforms_amd import _hip

def run(d, c, lat, hidden, act, n, fam, reps=10):
    shape = _hip.CvaeShape.make(d, c, lat, hidden, act, family=fam)
    P = _hip.cvae_param_count(shape)
    g = torch.Generator(device="cuda").manual_seed(1)
    p = torch.randn(P, device="cuda", generator=g) * 0.1
    X = torch.randn(n, d, device="cuda", generator=g); C = torch.randn(n, max(c, 1), device="cuda", generator=g)[:, :c].contiguous() if c else None
    eps = torch.randn(n, lat, device="cuda", generator=g)
    ws = torch.empty(_hip.cvae_workspace_bytes(shape, n), dtype=torch.uint8, device="cuda")
    grad = torch.empty(P, device="cuda"); loss = torch.empty(1, device="cuda")
    for _ in range(2):
        _hip.cvae_loss_grad(shape, p, X, C, None, eps, n, 1.0 / n, 0.3, grad, loss, ws)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        _hip.cvae_loss_grad(shape, p, X, C, None, eps, n, 1.0 / n, 0.3, grad, loss, ws)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, _hip.cvae_kernel_path(shape), float(loss)

for (d, c, lat, hidden, act) in [(10, 5, 10, (128, 128), "relu"), (4, 2, 3, (7, 9), "relu"), (16, 4, 2, (128,), "tanh"), (10, 5, 10, (256, 256), "tanh")]:
    for n in (32, 1024, 65536):
        row = []
        for fam in ("generic", "lmm", "auto"):
            try:
                ms, path, l = run(d, c, lat, hidden, act, n, fam, reps=3 if fam == "generic" and n > 10000 else 10)
                row.append("%s: %.3f ms (path %d, loss %.5f)" % (fam, ms, path, l))
            except Exception as e:
                row.append("%s: %s" % (fam, str(e)[:60]))
        print(d, c, lat, hidden, act, "n=%d" % n, " | ".join(row), flush=True)
