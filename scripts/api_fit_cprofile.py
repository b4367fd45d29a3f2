"""cProfile of one 32-epoch RealNVP.fit on the C2 arrays (host-side overheads)."""
import os, sys, time, cProfile, pstats
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from probaforms_amd.models import RealNVP
n = 1_000_000
X, C = bench.make_data(n, 16, 4, 0)
torch.manual_seed(0)
m = RealNVP(n_layers=8, hidden=(128,), batch_size=65536, n_epochs=1, lr=1e-3)
m.fit(X, C); torch.cuda.synchronize()
m.n_epochs = 32
m.fit(X, C); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable(); m.fit(X, C); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(30)
