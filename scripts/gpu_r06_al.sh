# round 6: flow kernels with the wave number in an SGPR, the epilogue's rows re-derived, the prior draw out of line (fewer spills) against the previous build
cd /root/repo
python -m pytest tests/test_hip_kernels.py tests/test_bench_sizes_gpu.py tests/test_prior.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
OPS=fwd,inv CFGS="c2 c3 c4" bash scripts/gpu_ab.sh _old ""
python - <<'PY'
# fused sampling (prior drawn in-kernel) -- the call the benchmark's timed region makes
import os, sys, time, torch
sys.path.insert(0, "/root/repo")
for lib in ("_old", ""):
    import subprocess
    code = r'''
import os, sys, torch
sys.path.insert(0, "/root/repo")
from probaforms_amd import _hip
import numpy as np
for name, (L, d, c, h) in (("c2", (8, 16, 4, 128)), ("c3", (12, 32, 8, 256)), ("c4", (8, 64, 16, 128))):
    n = 1 << 20
    shape = _hip.RnvpShape.make(L, d, c, (h,), "tanh", alt_masks=1)
    P = _hip.param_count(shape)
    g = torch.Generator(device="cuda").manual_seed(0)
    params = (torch.rand(P, device="cuda", generator=g) - 0.5) * 0.2
    cc = torch.randn(n, c, device="cuda", generator=g)
    xs = torch.empty(n, d, device="cuda")
    ws = torch.empty(_hip.workspace_bytes(shape, 1, n), dtype=torch.uint8, device="cuda")
    f = lambda: _hip.sample(shape, params, None, cc, n, 5, 0, xs, ws)
    for _ in range(60): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(30): f()
    b.record(); torch.cuda.synchronize()
    print("  ", name, "sample_ms=%.4f" % (a.elapsed_time(b) / 30), "checksum %.6f" % float(xs.double().abs().sum()))
'''
    env = dict(os.environ, RNVP_HIP_LIB="/root/repo/probaforms_amd/csrc/librnvp_hip%s.so" % lib)
    print("variant [%s]" % lib, flush=True)
    subprocess.run([sys.executable, "-c", code], env=env)
PY
