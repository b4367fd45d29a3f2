"""us per CVAE training step through cvae_fit_epoch: the persistent one-workgroup epoch (rnvp_resident.hip) vs the
batch-by-batch loop (family='generic' pins it inside the same entry point) and vs per-batch cvae_train_step calls"""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probaforms_amd import _hip

def run(d, c, lat, hidden, act, batch, nb=256):
    n = nb * batch
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(n, d, device="cuda", generator=g); cc = torch.randn(n, c, device="cuda", generator=g) if c else None
    eps = torch.randn(n, lat, device="cuda", generator=g); perm = torch.randperm(n, device="cuda", generator=g)
    out = []
    P = _hip.cvae_param_count(_hip.CvaeShape.make(d, c, lat, hidden, act))
    p0 = (torch.rand(P, device="cuda", generator=g) - 0.5) * 0.4
    for fam in ("auto", "generic", "steps"):
        shape = _hip.CvaeShape.make(d, c, lat, hidden, act, family="auto" if fam == "steps" else fam)
        p = p0.clone(); m = torch.zeros(P, device="cuda"); v = torch.zeros(P, device="cuda")
        ws = torch.empty(_hip.cvae_workspace_bytes(shape, batch), dtype=torch.uint8, device="cuda")
        gb = torch.empty(P, device="cuda"); hist = torch.empty(nb, device="cuda")
        def epoch(first):
            if fam == "steps":          # what CVAE.fit issued per batch before cvae_fit_epoch: the library's pick of kernels, one call per batch
                for k in range(nb):
                    _hip.cvae_train_step(shape, p, x, cc, perm[k * batch:(k + 1) * batch], eps[k * batch:(k + 1) * batch], batch, 1.0 / batch, 0.001,
                                         gb, hist[k:k + 1], m, v, 1e-3, 0.9, 0.999, 1e-8, 0.0, first + k, ws)
                return
            _hip.cvae_fit_epoch(shape, p, x, cc, perm, eps, n, batch, 0.001, gb, hist, m, v, 1e-3, 0.9, 0.999, 1e-8, 0.0, first, ws)
        epoch(1); torch.cuda.synchronize()
        t0 = time.perf_counter(); epoch(1 + nb); epoch(1 + 2 * nb); torch.cuda.synchronize()
        out.append(((time.perf_counter() - t0) / (2 * nb) * 1e6, float(hist[-1]), _hip.cvae_fit_epoch_resident(shape, batch), _hip.cvae_kernel_path(shape)))
    print("d=%d c=%d latent=%d hidden=%s %s batch=%d: cvae_fit_epoch %.1f us/step (one launch per epoch: %s) | loop pinned to the VALU kernels %.1f | "
          "cvae_train_step per batch (kernel path %d) %.1f | last loss %.5f / %.5f / %.5f" % (d, c, lat, hidden, act, batch, out[0][0], out[0][2], out[1][0],
          out[2][3], out[2][0], out[0][1], out[1][1], out[2][1]), flush=True)

for cfg in [(2, 1, 2, (10,), "tanh", 32), (5, 3, 2, (10,), "tanh", 32), (2, 1, 2, (10,), "relu", 32), (2, 1, 2, (10,), "tanh", 128),
            (16, 4, 2, (16,), "tanh", 32), (16, 4, 8, (32,), "tanh", 32), (16, 4, 2, (128,), "tanh", 32)]:
    run(*cfg)
