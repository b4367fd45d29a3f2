"""does the per-launch HIP event bracket (rnvp_profile_enable) cost queue time?  C2 epoch (16 batches of 65536) with and without it"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from probaforms_amd import _engine, _hip
from probaforms_amd.models import NormalizingFlow, RealNVPLayer, StandardNormalPrior
dev = torch.device("cuda:0")
Xh, Ch = bench.make_data(1_000_000, 16, 4, 0)
X = torch.from_numpy(Xh).to(dev); C = torch.from_numpy(Ch).to(dev)
torch.manual_seed(0)
layers = [RealNVPLayer(16, 4, (torch.arange(16) + i) % 2, (128,), "tanh") for i in range(8)]
nf = NormalizingFlow(layers, StandardNormalPrior(16, dev, host_rng=False))
for p in nf.parameters(): p.data = p.data.to(dev)
eng = nf.engine(); opt = _engine.FlatAdam(eng.flat.numel(), dev, lr=1e-3, weight_decay=0.0)
perm = torch.randperm(1_000_000, device=dev); losses = torch.zeros(16, device=dev)
def epochs(k):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): eng.fit_epoch(opt, X, C, perm, 65536, losses)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3
epochs(3)
for rep in range(3):
    _hip.profile_enable(0); a = epochs(10)
    _hip.profile_enable(16 * 10 + 8); b = epochs(10); _hip.profile_read(_hip.PROFILE_TRAIN)
    print("epoch of 16 batches: %.3f ms without the event bracket, %.3f ms with it" % (a, b), flush=True)
_hip.profile_enable(0)
