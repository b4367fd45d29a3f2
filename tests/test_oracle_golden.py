"""The CPU oracle (oracle/rnvp_oracle.c) against fixtures produced by the reference.

This is what pins the oracle: every function of the restatement is compared with
outputs of hse-cs/probaforms itself (tests/golden/make_golden.py, run in the build
container).  Tolerances are stated per quantity; integer data (masks, shuffle
indices) must match bit for bit.
"""
import numpy as np
import pytest
from cases import CASES, GRAD_STRIDE
from conftest import load_case, logp_mae_tol
from oracle import Shape, default_masks

ALL = list(CASES)


def _shape(cs):
    return Shape.make(cs["L"], cs["d"], cs["c"], cs["hidden"], cs["act"])


@pytest.mark.parametrize("name", ALL)
def test_masks_bit_exact(name):
    cs = load_case(name)                                   # realnvp.py:199
    assert np.array_equal(default_masks(cs["L"], cs["d"]), cs["masks"])


@pytest.mark.parametrize("name", ALL)
def test_param_count(name, oracle32):
    cs = load_case(name)
    assert oracle32.param_count(_shape(cs)) == cs["params"].size


@pytest.mark.parametrize("name", ALL)
def test_layer_f_per_layer(name, oracle32):
    """RealNVPLayer.f one layer at a time, each fed the reference's previous output (G2)."""
    cs = load_case(name); s = _shape(cs); g = cs["gold"]
    npl = cs["params"].size // cs["L"]
    cur = cs["X"]
    for l in range(cs["L"]):
        y, ld = oracle32.layer_f(s, cs["params"][l * npl:(l + 1) * npl], cs["masks"][l], cur, cs["C"])
        np.testing.assert_allclose(y, g["G2_layer_out"][l], rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(ld, g["G2_layer_ld"][l], rtol=2e-6, atol=2e-6)
        cur = g["G2_layer_out"][l]


@pytest.mark.parametrize("name", ALL)
def test_layer_g_per_layer(name, oracle32):
    cs = load_case(name); s = _shape(cs); g = cs["gold"]
    npl = cs["params"].size // cs["L"]
    cur = cs["Z"]
    for k, l in enumerate(range(cs["L"] - 1, -1, -1)):     # nflow.py:142
        y = oracle32.layer_g(s, cs["params"][l * npl:(l + 1) * npl], cs["masks"][l], cur, cs["C"])
        np.testing.assert_allclose(y, g["G3_layer_out"][k], rtol=2e-6, atol=2e-6)
        cur = g["G3_layer_out"][k]


@pytest.mark.parametrize("name", ALL)
@pytest.mark.parametrize("prec", [32, 64])
def test_log_prob(name, prec, oracle32, oracle64):
    o = oracle32 if prec == 32 else oracle64
    cs = load_case(name); g = cs["gold"]
    z, lp, mean = o.log_prob(_shape(cs), cs["params"], cs["X"], cs["C"], cs["masks"])
    assert np.abs(z - g["G2_z"]).mean() < 2e-6
    assert np.abs(lp - g["G2_logp"]).mean() < logp_mae_tol(name)
    assert abs(mean - g["G2_mean"]) < logp_mae_tol(name)


@pytest.mark.parametrize("name", ALL)
def test_sample_and_roundtrip(name, oracle32):
    cs = load_case(name); s = _shape(cs); g = cs["gold"]
    x = oracle32.sample(s, cs["params"], cs["Z"], cs["C"], cs["masks"])
    np.testing.assert_allclose(x, g["G3_x"], rtol=1e-5, atol=2e-5)
    z, _, _ = oracle32.log_prob(s, cs["params"], cs["X"], cs["C"], cs["masks"])
    back = oracle32.sample(s, cs["params"], z, cs["C"], cs["masks"])
    assert np.abs(back - cs["X"]).max() < max(2e-5, 10 * float(g["G3_roundtrip_maxerr"]))


@pytest.mark.parametrize("name", ALL)
@pytest.mark.parametrize("tag,rows", [("G4", None), ("G8", 8)])
def test_loss_and_gradient(name, tag, rows, oracle32):
    """hand-derived backward vs autograd of the reference (realnvp.py:246-250); G8 = ragged batch."""
    cs = load_case(name); g = cs["gold"]
    X = cs["X"][:rows]; C = None if cs["C"] is None else cs["C"][:rows]
    loss, grad = oracle32.loss_grad(_shape(cs), cs["params"], X, C, cs["masks"])
    assert abs(loss - g[tag + "_loss"]) < logp_mae_tol(name)
    if cs["wsrc"] == "torch":
        ref = g[tag + "_grad"]; got = grad
    else:
        ref = g[tag + "_grad_sub"]; got = grad[::GRAD_STRIDE]
        l2 = np.sqrt((grad.astype(np.float64) ** 2).sum())
        assert abs(l2 - g[tag + "_grad_l2"]) < 1e-5 * g[tag + "_grad_l2"]
    scale = np.abs(ref).max()
    assert np.abs(got - ref).max() < 2e-6 * scale + 1e-9
    # entries the masks make dead receive exactly zero gradient in both
    assert np.array_equal(ref == 0, got == 0) or np.abs(got[ref == 0]).max() < 1e-9


@pytest.mark.parametrize("name", [n for n in ALL if CASES[n][5] == "torch"])
@pytest.mark.parametrize("wd", [0.0, 0.2])
def test_adam_trajectory(name, wd, oracle32):
    """3 optimizer steps (realnvp.py:205-207,249-251) with weight_decay 0 and 0.2."""
    cs = load_case(name); s = _shape(cs); g = cs["gold"]
    k = "G4_adam_wd%g" % wd
    p = cs["params"].astype(np.float32).copy(); m = np.zeros_like(p); v = np.zeros_like(p)
    for step in range(3):
        loss, grad = oracle32.loss_grad(s, p, cs["X"], cs["C"], cs["masks"])
        assert abs(loss - g[k + "_loss"][step]) < 5e-5 * max(1.0, abs(loss))
        oracle32.adam(p, grad, m, v, step + 1, lr=0.01, weight_decay=wd)
        mr, vr = g[k + "_m"][step], g[k + "_v"][step]
        # gradients agree to ~1e-6 of their scale (cancellation in small entries), so do m and v
        np.testing.assert_allclose(m, mr, rtol=2e-5, atol=2e-6 * np.abs(mr).max())
        np.testing.assert_allclose(v, vr, rtol=4e-5, atol=4e-6 * np.abs(vr).max())
        # |dp| <= lr per step; sign-like sensitivity where the gradient is ~0
        assert np.abs(p - g[k + "_p"][step]).max() < 2e-4 * 0.01 * (step + 1) + 1e-7 or \
            np.mean(np.abs(p - g[k + "_p"][step])) < 1e-6


def test_prior_closed_form():
    """MultivariateNormal(0, I).log_prob == -0.5 (d ln 2pi + |z|^2)  (G6; nflow.py:115)."""
    import os
    from conftest import GOLDEN
    f = np.load(os.path.join(GOLDEN, "prior.npz"))
    for d in (1, 2, 5, 16):
        z = f["logprob_in_d%d" % d]
        lp = -0.5 * (d * np.log(2 * np.pi) + (z.astype(np.float64) ** 2).sum(1))
        np.testing.assert_allclose(lp, f["logprob_out_d%d" % d], rtol=1e-6, atol=1e-6)


def test_philox_known_answers_and_normal_moments(oracle32, oracle64):
    """the oracle's restatement of the build's counter-based prior (include/rnvp_hip.h rnvp_prior_normal):
    Philox4x32-10 against the published Random123 known-answer vectors; Box-Muller output moments; the
    draw is a function of the GLOBAL row only (row offsets tile)."""
    kat = [([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
           ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
           ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
            [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1])]
    for ctr, key, want in kat:
        assert oracle32.philox4x32_10(ctr, key) == want
    z = oracle64.prior_normal(1234, 0, 100000, 6)
    assert abs(z.mean()) < 5e-3 and abs(z.std() - 1) < 5e-3 and abs((z ** 3).mean()) < 2e-2
    assert abs((z ** 4).mean() - 3) < 5e-2 and np.abs(np.corrcoef(z.T) - np.eye(6)).max() < 1e-2
    a = oracle32.prior_normal(77, 0, 50, 5); b = oracle32.prior_normal(77, 20, 30, 5)
    np.testing.assert_array_equal(a[20:], b)
    assert not np.array_equal(oracle32.prior_normal(78, 0, 50, 5), a)
    assert np.abs(oracle32.prior_normal(77, 0, 50, 5) - oracle64.prior_normal(77, 0, 50, 5)).max() < 1e-6


@pytest.mark.parametrize("name", ["tm", "c2", "relu_sh", "tm_nocond"])
def test_eager_torch_cpu_baseline_matches_the_reference_outputs(name):
    """oracle/torch_cpu.py (the loop bench.py times as cpu_baseline) against the golden fixtures produced by the
    reference: per-row log-prob (G2) and the inverse of the fixture's z (G3)"""
    import torch
    from conftest import load_case
    from oracle.torch_cpu import EagerFlow
    cs = load_case(name); g = cs["gold"]
    flow = EagerFlow(cs["L"], cs["d"], cs["c"], cs["hidden"], cs["act"])
    flow.load_flat(cs["params"])
    X = torch.from_numpy(cs["X"]); C = None if cs["C"] is None else torch.from_numpy(cs["C"])
    with torch.no_grad():
        lp, z = flow.log_prob_rows(X, C)
        assert np.abs(lp.numpy() - g["G2_logp"]).max() < 1e-4 and np.abs(z.numpy() - g["G2_z"]).max() < 2e-5
        flow.prior.sample = lambda shape: torch.from_numpy(cs["Z"])           # the fixture's prior draw
        x = flow.sample(C, len(cs["Z"]))
        np.testing.assert_allclose(x.numpy(), g["G3_x"], rtol=2e-5, atol=2e-5)
