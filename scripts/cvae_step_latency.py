"""dev aid: device time of one CVAE training step (loss+grad kernels + Adam) against the batch size"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from probaforms_amd import _hip
from probaforms_amd.models import CVAE
X, C = bench.make_data(70000, 16, 4, 0)
for hidden in ((128,), (10,)):
    torch.manual_seed(0)
    m = CVAE(latent_dim=2, hidden=hidden, batch_size=4096, n_epochs=1, lr=1e-3, noise_rng="device")
    m.fit(X[:8192], C[:8192]); core = m._core
    Xd, Cd = torch.from_numpy(X).cuda(), torch.from_numpy(C).cuda()
    for B in (32, 1024, 8192, 65536):
        eps = torch.randn(B, 2, device="cuda"); idx = torch.randperm(70000, device="cuda")[:B].contiguous()
        g = core.grads(); ws = core.workspace(B)
        def step(t):
            _hip.cvae_loss_grad(core.shape, core.sync(), Xd, Cd, idx, eps, B, 1.0 / B, 0.001, g[:core.P], g[core.P:core.P + 1], ws)
            _hip.adam_step(core.sync(), g[:core.P], m.opt.exp_avg[:core.P], m.opt.exp_avg_sq[:core.P], core.P, 1e-3, 0.9, 0.999, 1e-8, 0.0, t)
        for t in range(5): step(t + 1)
        torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for t in range(100): step(t + 6)
        b.record(); torch.cuda.synchronize()
        print("hidden %s batch %5d: %.1f us per step" % (hidden, B, a.elapsed_time(b) / 100 * 1e3))
