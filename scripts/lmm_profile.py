"""kernel-level view of the lmm path: python scripts/lmm_profile.py h1,h2[,h3]   (run under rocprofv3 --kernel-trace --stats)"""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import bench
from probaforms_amd.models import RealNVP
hidden = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "128,128").split(","))
n = 262144
Xh, Ch = bench.make_data(n, 16, 4, 0)
torch.manual_seed(0)
m = RealNVP(n_layers=8, hidden=hidden, batch_size=65536, n_epochs=2, lr=1e-3, prior_rng="device")
m.fit(Xh, Ch); m.sample(Ch); m.nf.log_prob_samples(Xh, Ch); torch.cuda.synchronize()
