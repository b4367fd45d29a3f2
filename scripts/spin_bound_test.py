import sys, os, time, torch
sys.path.insert(0, "/root/repo")
from probaforms_amd import _hip
shape = _hip.RnvpShape.make(8, 16, 4, (128,), "tanh", alt_masks=1)
P = _hip.param_count(shape); n = 65536
g = torch.Generator(device="cuda").manual_seed(1)
params = (torch.rand(P, device="cuda", generator=g) - 0.5) * 0.2
x = torch.randn(n, 16, device="cuda", generator=g); c = torch.randn(n, 4, device="cuda", generator=g)
ws = torch.empty(_hip.workspace_bytes(shape, 2, n), dtype=torch.uint8, device="cuda")
gb = torch.zeros(P + 1, device="cuda")
torch.cuda.synchronize(); t0 = time.time()
_hip.loss_grad(shape, params, None, x, c, None, n, 1.0 / n, gb[:P], gb[P:P + 1], ws)
torch.cuda.synchronize()
print("lib %s: %.3f s, loss %s" % (os.path.basename(_hip.LIB_PATH), time.time() - t0, float(gb[P])))
