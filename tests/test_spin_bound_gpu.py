"""The time-out path of the training kernel's barrier-free flush, where the driver runs the GPU tests.

csrc/librnvp_hip_spt.so is the product library with ONE object rebuilt under -DRNVP_SPIN_TEST=1 (csrc/Makefile): wave 5 of
workgroup 0 withholds one arrival, so the flushing waves' bounded wait (rnvp_mfma_layer.h spin_nap) must give up, raise the
step's error word and end the launch -- within about a second, with the protocol NaN as the loss and NO Adam step applied --
instead of hanging the device.  The library is loaded once per process, so the variant runs in a child process
(RNVP_HIP_LIB selects it)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SPT = os.path.join(ROOT, "probaforms_amd", "csrc", "librnvp_hip_spt.so")

CHILD = r"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, %(root)r)
from probaforms_amd import _hip, _engine
from probaforms_amd.models import RealNVP
assert os.path.basename(_hip.LIB_PATH) == "librnvp_hip_spt.so", _hip.LIB_PATH
shape = _hip.RnvpShape.make(8, 16, 4, (128,), "tanh", alt_masks=1)
P = _hip.param_count(shape); n = 65536
g = torch.Generator(device="cuda").manual_seed(1)
params = (torch.rand(P, device="cuda", generator=g) - 0.5) * 0.2
x = torch.randn(n, 16, device="cuda", generator=g); c = torch.randn(n, 4, device="cuda", generator=g)
ws = torch.empty(_hip.workspace_bytes(shape, _hip.OP_TRAIN, n), dtype=torch.uint8, device="cuda")
gb = torch.zeros(P + 4, device="cuda"); loss = torch.zeros(1, device="cuda")
m = torch.zeros(P, device="cuda"); v = torch.zeros(P, device="cuda")
before = params.clone()
torch.cuda.synchronize(); t0 = time.time()
_hip.train_step(shape, params, None, x, c, None, n, 1.0 / n, gb[:P], loss, m, v, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, ws)
torch.cuda.synchronize(); dt = time.time() - t0
bits = int(loss.cpu().view(torch.int32)[0]) & 0xffffffff
print("SPT step %%.3f s loss bits %%08x params_changed %%d moments_nonzero %%d" %% (
    dt, bits, int((params != before).any()), int((m != 0).any() or (v != 0).any())))
assert dt < 20.0, dt
assert bits == _engine.PROTOCOL_NAN_BITS, hex(bits)
assert not bool((params != before).any()) and not bool((m != 0).any()) and not bool((v != 0).any())
# a batch too small to reach the withheld arrival still trains (the variant is otherwise the product)
_hip.train_step(shape, params, None, x, c, None, 512, 1.0 / 512, gb[:P], loss, m, v, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, ws)
torch.cuda.synchronize()
assert np.isfinite(float(loss)) and bool((params != before).any())
# and the class API turns the protocol NaN into an exception
rng = np.random.default_rng(0)
X = rng.normal(size=(n, 16)).astype(np.float32); C = rng.normal(size=(n, 4)).astype(np.float32)
torch.manual_seed(0)
model = RealNVP(n_layers=8, hidden=(128,), batch_size=65536, n_epochs=1, lr=1e-3)
try:
    model.fit(X, C)
except RuntimeError as e:
    assert "protocol error" in str(e), e
    print("SPT fit raised:", str(e)[:60])
else:
    raise AssertionError("RealNVP.fit did not raise on the protocol NaN")
print("SPT OK")
"""


@pytest.mark.gpu
def test_flush_timeout_ends_the_launch_and_skips_adam():
    assert os.path.exists(SPT), "librnvp_hip_spt.so missing: make -C probaforms_amd/csrc"
    env = dict(os.environ, RNVP_HIP_LIB=SPT)
    out = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "SPT OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
