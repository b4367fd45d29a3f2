import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
import bench
from probaforms_amd.models import RealNVP
n = 262144
Xh, Ch = bench.make_data(n, 16, 4, 0)
for hidden in [(128,), (64, 64), (128, 128), (10, 20, 15)]:
    torch.manual_seed(0)
    m = RealNVP(n_layers=8, hidden=hidden, batch_size=65536, n_epochs=1, lr=1e-3, prior_rng="device")
    m.fit(Xh, Ch)
    torch.cuda.synchronize(); t0 = time.perf_counter(); m.fit(Xh, Ch); torch.cuda.synchronize(); tf = time.perf_counter() - t0
    m.sample(Ch); t0 = time.perf_counter(); m.sample(Ch); ts = time.perf_counter() - t0
    lp0 = time.perf_counter(); m.nf.log_prob_samples(Xh, Ch); torch.cuda.synchronize(); tl = time.perf_counter() - lp0
    print(hidden, "fit epoch %.1f ms (%.2f M rows/s)  sample %.1f ms  log_prob %.1f ms" % (tf * 1e3, n / tf / 1e6, ts * 1e3, tl * 1e3))
