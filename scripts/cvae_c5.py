"""BASELINE.json configs[4]: CVAE on the C2 data (n=1M, d=16, cond=4), hidden=(128,), latent 2,
batch 65536 -- device-resident step time of the fused CVAE kernel + Adam, and API-level fit."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from probaforms_amd import _hip
from probaforms_amd.models import CVAE
n = 1_000_000
X, C = bench.make_data(n, 16, 4, 0)
torch.manual_seed(0)
m = CVAE(latent_dim=2, hidden=(128,), batch_size=65536, n_epochs=1, lr=1e-3)
t0 = time.perf_counter(); m.fit(X, C); torch.cuda.synchronize(); t1 = time.perf_counter()
print("first fit (1 epoch + full-data loss) %.1f ms" % ((t1 - t0) * 1e3))
m.n_epochs = 3
t0 = time.perf_counter(); m.fit(X, C); torch.cuda.synchronize(); t1 = time.perf_counter()
print("API-level fit 3 epochs %.1f ms -> %.2f M rows/s; loss_history %s" % ((t1 - t0) * 1e3, 3 * n / (t1 - t0) / 1e6, [round(float(v), 4) for v in m.loss_history]))
core = m._core
Xd, Cd = torch.from_numpy(X).cuda(), torch.from_numpy(C).cuda()
B = 65536
eps = torch.randn(B, 2, device="cuda"); idx = torch.randperm(n, device="cuda")[:B].contiguous()
g = core.grads(); ws = core.workspace(B)
def step():
    _hip.cvae_loss_grad(core.shape, core.sync(), Xd, Cd, idx, eps, B, 1.0 / B, 0.001, g[:core.P], g[core.P:core.P + 1], ws)
    _hip.adam_step(core.sync(), g[:core.P], m.opt.exp_avg[:core.P], m.opt.exp_avg_sq[:core.P], core.P, 1e-3, 0.9, 0.999, 1e-8, 0.0, 5)
for _ in range(3): step()
torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): step()
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 10
print("device-resident train step, 65536 rows: %.3f ms -> %.1f M rows/s" % (ms, B / ms / 1e3))
for noise in ("host", "device"):
    mm = CVAE(latent_dim=2, hidden=(128,), batch_size=65536, n_epochs=8, lr=1e-3, noise_rng=noise)
    mm.fit(X, C); torch.cuda.synchronize()
    t0 = time.perf_counter(); mm.fit(X, C); torch.cuda.synchronize(); t1 = time.perf_counter()
    print("API-level fit 8 epochs, noise_rng=%s: %.1f ms = %.2f ms/epoch -> %.1f M rows/s" % (noise, (t1 - t0) * 1e3, (t1 - t0) * 1e3 / 8, 8 * n / (t1 - t0) / 1e6))
t0 = time.perf_counter(); xs = m.sample(C); t1 = time.perf_counter()
print("sample(1M): %.1f ms" % ((t1 - t0) * 1e3))
