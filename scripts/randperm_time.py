"""torch.randperm on the host against the same permutation drawn on the device (rnvp_randperm_torch_cpu): python scripts/randperm_time.py"""
import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
from probaforms_amd import _hip
from probaforms_amd.models.nflow import HostStreamOnDevice
for n in (65536, 262144, 1000000, 4000000, 16000000):
    g = torch.Generator(); g.manual_seed(5)
    t0 = time.perf_counter(); ref = torch.randperm(n, generator=g); th = time.perf_counter() - t0
    g.manual_seed(5)
    st, mt = HostStreamOnDevice._unpack(g)
    out = torch.empty(n, dtype=torch.int64, device="cuda")
    ws = torch.empty(_hip.randperm_workspace_bytes(n), dtype=torch.uint8, device="cuda")
    ts = []
    for _ in range(5):
        mtd = torch.from_numpy(mt.copy()).cuda(); torch.cuda.synchronize()
        t0 = time.perf_counter(); _hip.randperm_torch_cpu(mtd, n, out, ws); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print("n=%d: torch.randperm on the host %.2f ms; on the device %.3f ms (best of 5, incl. launch); identical %s"
          % (n, th * 1e3, min(ts) * 1e3, bool(torch.equal(out.cpu(), ref))))
