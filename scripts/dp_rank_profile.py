"""One rank's step of the 8-GPU strong-batch regime (8 192 rows) through rnvp_fit_epoch_dp on a one-rank RCCL communicator, for
rocprofv3 --kernel-trace --stats: which launches make up the step.   usage: python3 scripts/dp_rank_profile.py c3|c2 [epochs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from probaforms_amd import _engine, _hip
from probaforms_amd.models import NormalizingFlow, RealNVPLayer, StandardNormalPrior

key = sys.argv[1] if len(sys.argv) > 1 else "c3"
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda:0")
w = bench.WORKLOADS[key]
d, c, L, hidden = w["d"], w["c"], w["L"], w["hidden"]
comm = _hip.dp_init(_hip.dp_unique_id(), 0, 1)
torch.manual_seed(0)
layers = [RealNVPLayer(d, c, (torch.arange(d) + i) % 2, hidden, "tanh") for i in range(L)]
nf = NormalizingFlow(layers, StandardNormalPrior(d, dev, host_rng=False))
for p in nf.parameters():
    p.data = p.data.to(dev)
eng = nf.engine(); opt = _engine.FlatAdam(eng.flat.numel(), dev, lr=1e-3, weight_decay=0.0)
gen = torch.Generator(device=dev).manual_seed(5)
nsteps, rb = 64, 8192
n = nsteps * rb
X = torch.randn(n, d, device=dev, generator=gen); C = torch.randn(n, c, device=dev, generator=gen)
perm = torch.randperm(n, device=dev, generator=gen); losses = torch.zeros(nsteps, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for ep in range(epochs):
    if ep == epochs - 1: e0.record()
    eng.fit_epoch_dp(opt, comm, X, C, perm, rb, losses)
e1.record(); torch.cuda.synchronize(dev)
print("%s: %.1f us per rank step (last of %d epochs of %d steps), dispatch %r" % (key, e0.elapsed_time(e1) / nsteps * 1e3, epochs, nsteps, _hip.last_dispatch(_hip.PROFILE_TRAIN)))
_hip.dp_destroy(comm)
