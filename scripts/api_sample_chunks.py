import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
import bench
from probaforms_amd.models import RealNVP
Xh, Ch = bench.make_data(1_000_000, 16, 4, 0)
for prior in ("device", "host"):
    torch.manual_seed(0)
    m = RealNVP(n_layers=8, hidden=(128,), batch_size=65536, n_epochs=1, lr=1e-3, prior_rng=prior)
    m.fit(Xh, Ch)
    for chunk in (32 << 20, 16 << 20, 8 << 20, 4 << 20):
        m.nf.PIPELINE_CHUNK_BYTES = chunk
        m.sample(Ch)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); xs = m.sample(Ch); ts.append(time.perf_counter() - t0)
        print(prior, "chunk %2d MB pipelined=%s  sample(1M): best %.2f ms  median %.2f ms" % (chunk >> 20, bool(m.nf.pipelined_rows(1_000_000)), min(ts) * 1e3, sorted(ts)[2] * 1e3))
