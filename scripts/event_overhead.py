"""dev aid: what the library's HIP-event brackets around the hot kernels cost the fused single-GPU step"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from probaforms_amd import _engine, _hip
from probaforms_amd.models import NormalizingFlow, RealNVPLayer, StandardNormalPrior
dev = torch.device("cuda", 0)
torch.manual_seed(0)
D, CD, H, L, N, B = 16, 4, (128,), 8, 1_000_000, 65536
layers = [RealNVPLayer(D, CD, (torch.arange(D) + i) % 2, H, "tanh") for i in range(L)]
nf = NormalizingFlow(layers, StandardNormalPrior(D, dev, host_rng=False))
for p in nf.parameters(): p.data = p.data.to(dev)
eng = nf.engine(); opt = _engine.FlatAdam(eng.flat.numel(), dev, lr=1e-3, weight_decay=0.0)
Xh, Ch = bench.make_data(N, D, CD, 0); X = torch.from_numpy(Xh).to(dev); C = torch.from_numpy(Ch).to(dev)
nb = len(_engine.batch_bounds(N, B)); xs = torch.empty(N, D, device=dev)
perm = torch.randperm(N, device=dev); losses = torch.zeros(nb, device=dev)
def step(i):
    eng.fit_epoch(opt, X, C, perm, B, losses); eng.sample(N, C, 1000 + i, row_offset=0, out=xs)
for mode in ("off", "on", "off", "on"):
    _hip.profile_enable(20 * nb + 8 if mode == "on" else 0)
    for i in range(2): step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(20): step(i)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print("events %s: %.3f ms per step" % (mode, dt * 1e3))
    if mode == "on":
        n, ms = _hip.profile_read(_hip.PROFILE_TRAIN); print("   train kernel avg %.1f us over %d" % (ms / n * 1e3, n))
_hip.profile_enable(0)
