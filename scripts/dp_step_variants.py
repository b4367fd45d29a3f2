"""dev aid: where the data-parallel step loses time on ONE rank (RCCL initialised, world size 1)"""
import os, sys, time
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
import bench
from probaforms_amd import _engine
from probaforms_amd.models import NormalizingFlow, RealNVPLayer, StandardNormalPrior
D, CDIM, L, H, B, N = bench.D, bench.CDIM, bench.LAYERS, bench.HIDDEN, bench.BATCH, 262144
dev = torch.device("cuda", 0)
torch.manual_seed(0)
nf = NormalizingFlow([RealNVPLayer(D, CDIM, (torch.arange(D) + i) % 2, H, "tanh") for i in range(L)], StandardNormalPrior(D, dev))
for p in nf.parameters(): p.data = p.data.to(dev)
eng = nf.engine(); opt = _engine.FlatAdam(eng.flat.numel(), dev, lr=1e-3, weight_decay=0.0)
X = torch.randn(N, D, device=dev); C = torch.randn(N, CDIM, device=dev); perm = torch.randperm(N, device=dev)
z = torch.randn(B, D, device=dev); xs = torch.empty_like(z); losses = torch.zeros(64, device=dev); P = eng.P
comm = torch.cuda.Stream()
def fused(i):
    eng.train_step(opt, X, C, perm[:B], B, 1.0 / B, losses[i:i + 1]); eng.inverse(z, C[:B], out=xs)
def unfused(i):
    g = eng.loss_grad(X, C, perm[:B], B, 1.0 / B); eng.inverse(z, C[:B], out=xs); losses[i:i + 1].copy_(g[P:P + 1]); eng.adam(opt)
def ar_async(i):
    g = eng.loss_grad(X, C, perm[:B], B, 1.0 / B); w = dist.all_reduce(g[:P + 1], async_op=True); eng.inverse(z, C[:B], out=xs); w.wait()
    losses[i:i + 1].copy_(g[P:P + 1]); eng.adam(opt)
def ar_sync_after(i):
    g = eng.loss_grad(X, C, perm[:B], B, 1.0 / B); eng.inverse(z, C[:B], out=xs); dist.all_reduce(g[:P + 1])
    losses[i:i + 1].copy_(g[P:P + 1]); eng.adam(opt)
def ar_sync_before(i):
    g = eng.loss_grad(X, C, perm[:B], B, 1.0 / B); dist.all_reduce(g[:P + 1]); eng.inverse(z, C[:B], out=xs)
    losses[i:i + 1].copy_(g[P:P + 1]); eng.adam(opt)
for name, fn in (("fused", fused), ("unfused, no collective", unfused), ("all_reduce async under inverse", ar_async),
                 ("all_reduce sync after inverse", ar_sync_after), ("all_reduce sync before inverse", ar_sync_before)):
    for i in range(8): fn(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(48): fn(i)
    torch.cuda.synchronize(); print("%-34s %.1f us/step" % (name, (time.perf_counter() - t0) / 48 * 1e6), flush=True)
dist.destroy_process_group()
