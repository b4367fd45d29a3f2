"""The reference's prior stream (`torch.randn` on the global CPU generator, nflow.py:141) drawn on the device.

CPU (`-m "not gpu"`): the C restatement of torch's CPU draw (oracle/prior_torch_oracle.c: mt19937, 24-bit uniforms, 16-element
Box-Muller blocks, avx_mathfun.h's log256_ps / sincos256_ps one lane at a time with GCC's multiply-add contractions) against
torch.randn ITSELF -- the reference's arithmetic lives in PyTorch -- on streams of up to 4M numbers and the block / tail edges.
GPU (`-m gpu`): rnvp_prior_normal_torch_cpu against torch.randn, bit for bit, values and generator state; RealNVP.sample with
the default (host) prior gives the bytes the host draw gives.
"""
import ctypes as C
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _olib():
    from oracle import oracle as o
    o.build()
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_build", "libprior_torch_oracle.so"))
    lib.prior_torch_randn.restype, lib.prior_torch_randn.argtypes = C.c_int, [C.c_void_p, C.c_int64, C.c_void_p]
    return lib


SIZES = [16, 17, 31, 32, 100, 623, 624, 625, 640, 1000, 4101, 100003, 1 << 22]


@pytest.mark.parametrize("n", SIZES)
def test_oracle_restatement_equals_torch_randn(n):
    from probaforms_amd.models.nflow import HostStreamOnDevice as H
    lib = _olib()
    g = torch.Generator(); g.manual_seed(1234 + n)
    torch.rand(n % 700, generator=g)                       # start at an arbitrary position of a block
    st, mt = H._unpack(g)
    mt = mt.copy()
    ref = torch.randn(n, generator=g).numpy()
    out = np.empty(n, np.float32)
    assert lib.prior_torch_randn(mt.ctypes.data, n, out.ctypes.data) == 0
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32))
    g2 = torch.Generator(); g2.manual_seed(0)
    H._pack(g2, st, mt)                                    # the advanced twister state, written back
    assert torch.equal(g2.get_state(), g.get_state())
    assert torch.equal(torch.randn(40, generator=g2), torch.randn(40, generator=g))


@pytest.mark.gpu
def test_device_draw_is_validated_on_this_host():
    from probaforms_amd.models.nflow import HostStreamOnDevice as H
    assert H.usable("cuda"), "the device draw differs from this host's torch.randn (capability %s): the host draw stays in use" % \
        torch.backends.cpu.get_cpu_capability()


@pytest.mark.gpu
@pytest.mark.parametrize("n", SIZES + [1 << 20, 638976 + 100, 2 * 638976, 16 * 1000 * 1000 + 5, 25 * 1000 * 1000 + 3])   # ... several jump segments, two rounds
def test_device_draw_equals_torch_randn(n):
    from probaforms_amd.models.nflow import HostStreamOnDevice as H
    if not H.usable("cuda"):
        pytest.skip("device draw not validated on this host")
    g = torch.Generator(); g.manual_seed(99 + n % 1000)
    torch.rand(n % 700, generator=g)
    g2 = torch.Generator(); g2.set_state(g.get_state())
    ref = torch.randn(n, generator=g)
    hs = H("cuda", g2)
    got = hs.randn((n,)).cpu()
    assert torch.equal(ref.view(torch.int32), got.view(torch.int32))
    assert torch.equal(g.get_state(), g2.get_state())
    # chained draws with one state round trip continue the stream
    hs.begin()
    a = hs.draw(torch.empty(48, device="cuda")); b = hs.draw(torch.empty(33, device="cuda"))
    hs.end()
    assert torch.equal(torch.randn(48, generator=g), a.cpu()) and torch.equal(torch.randn(33, generator=g), b.cpu())
    assert torch.equal(g.get_state(), g2.get_state())


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1000, 300001, 6000011])      # one draw / the chunked pipeline / two draw windows
def test_realnvp_sample_default_prior_is_the_host_stream(n):
    """RealNVP.sample with the default prior: the same bytes whether z is drawn on the host or on the device, and the global
    generator ends in the same state (n = 300001 takes the chunked sample_to_host pipeline, 6000011 rows need two draw windows)"""
    from probaforms_amd.models import RealNVP
    from probaforms_amd.models.nflow import HostStreamOnDevice as H
    rng = np.random.default_rng(0)
    X = rng.normal(size=(256, 6)).astype(np.float32); Cc = rng.normal(size=(256, 2)).astype(np.float32)
    torch.manual_seed(0)
    m = RealNVP(n_layers=4, hidden=(16,), batch_size=64, n_epochs=1, lr=1e-3)
    m.fit(X, Cc)
    Cs = rng.normal(size=(n, 2)).astype(np.float32)
    outs, states = [], []
    saved = dict(H._ok)
    try:
        for force_host in (True, False):
            H._ok.clear()
            if force_host:
                H._ok[torch.device("cuda").index] = False; H._ok[0] = False; H._ok[None] = False
            torch.manual_seed(7)
            outs.append(m.sample(Cs).copy())
            states.append(torch.get_rng_state().clone())
    finally:
        H._ok.clear(); H._ok.update(saved)
    assert outs[0].shape == (n, 6)
    assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))
    assert torch.equal(states[0], states[1])
