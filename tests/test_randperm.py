"""The reference's epoch shuffle (`DataLoader(shuffle=True)` -> RandomSampler -> `torch.randperm(n, generator=g)`, realnvp.py:235).

CPU (`-m "not gpu"`): the C restatement (oracle/randperm_torch_oracle.c: mt19937 draws, the sequential Fisher-Yates pass of ATen's
randperm_cpu) against torch.randperm ITSELF -- the reference's arithmetic lives in PyTorch -- values and generator state; and the
parallel formulation the device kernels use (rounds of deterministic reservations, csrc/rnvp_randperm.hip) simulated with numpy
against the sequential pass.  The device path itself: tests/test_randperm_gpu.py.
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
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_build", "librandperm_torch_oracle.so"))
    lib.randperm_torch.restype, lib.randperm_torch.argtypes = C.c_int, [C.c_void_p, C.c_int64, C.c_void_p]
    return lib


@pytest.mark.parametrize("n", [0, 1, 2, 3, 7, 100, 623, 624, 625, 1000, 65536, 100003, 1000000])
def test_oracle_restatement_equals_torch_randperm(n):
    from probaforms_amd.models.nflow import HostStreamOnDevice as H
    lib = _olib()
    for seed in (0, 12345, 2 ** 40 + 17, 2 ** 63 - 1):
        g = torch.Generator(); g.manual_seed(seed)
        torch.rand(seed % 700, generator=g)                    # start at an arbitrary position of a block
        st, mt = H._unpack(g)
        mt = mt.copy()
        ref = torch.randperm(n, generator=g).numpy()
        out = np.empty(n, np.int64)
        assert lib.randperm_torch(mt.ctypes.data, n, out.ctypes.data) == 0
        assert np.array_equal(out, ref)
        g2 = torch.Generator(); g2.manual_seed(0)
        H._pack(g2, st, mt)                                    # the advanced twister state, written back
        assert torch.equal(torch.randperm(40, generator=g2), torch.randperm(40, generator=g))
    assert lib.randperm_torch(mt.ctypes.data, 0xffffffff // 20, None) == -1      # torch's other branch: refused


@pytest.mark.parametrize("n", [2, 5, 1000, 50000])
def test_rounds_of_deterministic_reservations_give_the_sequential_shuffle(n):
    """the formulation of csrc/rnvp_randperm.hip: every pending iteration i bids for cells i and H[i], the smaller i wins, winners of
    both cells swap, losers bid again -- the sequential permutation whatever the order inside a round"""
    rng = np.random.default_rng(n)
    H = np.arange(n - 1) + (rng.integers(0, 2 ** 32, n - 1, dtype=np.uint64) % (n - np.arange(n - 1)).astype(np.uint64)).astype(np.int64)
    seq = np.arange(n)
    for i in range(n - 1):
        j = H[i]; seq[i], seq[j] = seq[j], seq[i]
    r = np.arange(n)
    pending = rng.permutation(n - 1)                           # any order: priorities are the indices
    rounds = 0
    while pending.size:
        bids = np.full(n, np.iinfo(np.int64).max)
        np.minimum.at(bids, pending, pending)
        np.minimum.at(bids, H[pending], pending)
        win = (bids[pending] == pending) & (bids[H[pending]] == pending)
        w = pending[win]
        a, b = r[w].copy(), r[H[w]].copy()
        r[w] = b; r[H[w]] = a
        pending = rng.permutation(pending[~win])
        rounds += 1
    assert np.array_equal(r, seq) and rounds <= 64
