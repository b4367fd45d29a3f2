"""kernel-level time of the CVAE C5 step (cvae_train_step, 65536 rows) via the library's own event bracket + wall clock"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probaforms_amd import _hip
n = int(os.environ.get("N", 65536))
shape = _hip.CvaeShape.make(16, 4, 2, (128,), "tanh")
P = _hip.cvae_param_count(shape)
g = torch.Generator(device="cuda").manual_seed(0)
p = (torch.rand(P, device="cuda", generator=g) - 0.5) * 0.2
X = torch.randn(n, 16, device="cuda", generator=g); C = torch.randn(n, 4, device="cuda", generator=g); eps = torch.randn(n, 2, device="cuda", generator=g)
ws = torch.empty(_hip.cvae_workspace_bytes(shape, n), dtype=torch.uint8, device="cuda")
grad = torch.empty(P, device="cuda"); loss = torch.empty(1, device="cuda"); m = torch.zeros(P, device="cuda"); v = torch.zeros(P, device="cuda")
def step(i):
    _hip.cvae_train_step(shape, p, X, C, None, eps, n, 1.0 / n, 0.001, grad, loss, m, v, 1e-3, 0.9, 0.999, 1e-8, 0.0, i + 1, ws)
for i in range(200): step(i)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for i in range(200): step(i + 200)
b.record(); torch.cuda.synchronize()
_hip.profile_enable(True) if hasattr(_hip, "profile_enable") else None
print("lib %s: step %.2f us (n=%d) loss %.5f" % (os.path.basename(_hip.LIB_PATH), a.elapsed_time(b) / 200 * 1e3, n, float(loss)))
