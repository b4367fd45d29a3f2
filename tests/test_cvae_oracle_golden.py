"""CVAE oracle (oracle/cvae_oracle.c) against fixtures produced by the reference's CVAE
(tests/golden/make_golden_cvae.py): this pins the oracle for SURVEY.md 8(f) rank 1."""
import os

import numpy as np
import pytest
from conftest import GOLDEN
from oracle import CvaeOracle, CvaeShape, Oracle

CASES = {"default": (5, 3, 2, (10,), "tanh", 0.001), "nocond": (5, 0, 2, (10,), "tanh", 0.001),
         "c5": (16, 4, 2, (128,), "tanh", 0.001), "relu_mh": (4, 2, 3, (7, 9), "relu", 0.5)}


def load(name):
    d, c, lat, hidden, act, klw = CASES[name]
    g = np.load(os.path.join(GOLDEN, "cvae_%s.npz" % name))
    return g, CvaeShape.make(d, c, lat, hidden, act), klw, (g["C"] if c else None)


@pytest.mark.parametrize("name", list(CASES))
def test_param_count_and_encode_decode(name):
    g, s, klw, C = load(name)
    o = CvaeOracle(32)
    assert o.param_count(s) == g["init_params"].size
    mu, ls = o.encode(s, g["init_params"], g["X"], C)
    np.testing.assert_allclose(mu, g["mu"], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(ls, g["log_sigma"], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(o.decode(s, g["init_params"], g["Z"], C), g["decoded"], rtol=2e-6, atol=2e-6)


@pytest.mark.parametrize("name", list(CASES))
def test_loss_and_gradient(name):
    g, s, klw, C = load(name)
    o = CvaeOracle(32)
    loss, grad = o.loss_grad(s, g["init_params"], g["X"], C, g["eps"], klw)
    assert abs(loss - g["loss"]) < 2e-6 * max(1.0, abs(loss))
    assert np.abs(grad - g["grad"]).max() < 2e-6 * np.abs(g["grad"]).max() + 1e-9
    loss_only, none = o.loss_grad(s, g["init_params"], g["X"], C, g["eps"], klw, want_grad=False)
    assert none is None and loss_only == loss


@pytest.mark.parametrize("name", list(CASES))
def test_adam_trajectory(name):
    g, s, klw, C = load(name)
    o, adam = CvaeOracle(32), Oracle(32)
    p = g["init_params"].astype(np.float32).copy(); m = np.zeros_like(p); v = np.zeros_like(p)
    for step in range(3):
        loss, grad = o.loss_grad(s, p, g["X"], C, g["adam_eps"][step], klw)
        assert abs(loss - g["adam_loss"][step]) < 5e-5 * max(1.0, abs(loss))
        adam.adam(p, grad, m, v, step + 1, lr=0.01)
        assert np.abs(p - g["adam_p"][step]).mean() < 2e-6
