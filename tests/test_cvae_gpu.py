"""CVAE on the GPU: HIP kernels through the C ABI (include/cvae_hip.h)
against the reference's own fixtures and the oracle; seeded end-to-end fit/sample against the
reference; the reference's API contract (tests/test_models.py) for the CVAE class."""
import os

import numpy as np
import pytest
import torch
from conftest import GOLDEN

pytestmark = pytest.mark.gpu

CASES = {"default": (5, 3, 2, (10,), "tanh", 0.001), "nocond": (5, 0, 2, (10,), "tanh", 0.001),
         "c5": (16, 4, 2, (128,), "tanh", 0.001), "relu_mh": (4, 2, 3, (7, 9), "relu", 0.5)}


def _dev(a):
    return None if a is None else torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).cuda().contiguous()


def _load(name, family="auto"):
    from probaforms_amd import _hip
    d, c, lat, hidden, act, klw = CASES[name]
    g = np.load(os.path.join(GOLDEN, "cvae_%s.npz" % name))
    shape = _hip.CvaeShape.make(d, c, lat, hidden, act, family=family)
    assert _hip.cvae_param_count(shape) == g["init_params"].size
    return _hip, g, shape, klw, (g["C"] if c else None)


@pytest.mark.parametrize("with_ws", [False, True])           # NULL workspace: generic kernels; with one: MFMA where supported
@pytest.mark.parametrize("name", list(CASES))
def test_encode_decode(name, with_ws):
    _hip, g, shape, klw, C = _load(name)
    n = g["X"].shape[0]
    p = _dev(g["init_params"])
    ws = torch.empty(_hip.cvae_workspace_bytes(shape, n), dtype=torch.uint8, device="cuda") if with_ws else None
    mu = torch.empty(n, shape.lat, device="cuda"); ls = torch.empty_like(mu)
    _hip.cvae_encode(shape, p, _dev(g["X"]), _dev(C), n, mu, ls, ws)
    np.testing.assert_allclose(mu.cpu().numpy(), g["mu"], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(ls.cpu().numpy(), g["log_sigma"], rtol=2e-6, atol=2e-6)
    x = torch.empty(n, shape.d, device="cuda")
    _hip.cvae_decode(shape, p, _dev(g["Z"]), _dev(C), n, x, ws)
    np.testing.assert_allclose(x.cpu().numpy(), g["decoded"], rtol=2e-6, atol=2e-6)


@pytest.fixture(params=["auto", "generic", "lmm"])
def path(request):
    """run the step on the kernels the library picks (register-chained MFMA where supported, else any-shape MFMA) and
    pinned to one thread per row / to the any-shape MFMA kernels (cvae_shape.family, per call)"""
    return request.param


def test_kernel_path_selection():
    from probaforms_amd import _hip
    for name, (d, c, lat, hidden, act, _) in CASES.items():
        want = _hip.PATH_LMM if name == "relu_mh" else _hip.PATH_MFMA
        assert _hip.cvae_kernel_path(_hip.CvaeShape.make(d, c, lat, hidden, act)) == want
        assert _hip.cvae_kernel_path(_hip.CvaeShape.make(d, c, lat, hidden, act, family="lmm")) == _hip.PATH_LMM
    assert _hip.cvae_kernel_path(_hip.CvaeShape.make(17, 4, 2, (128,), "tanh")) == _hip.PATH_LMM
    assert _hip.cvae_kernel_path(_hip.CvaeShape.make(16, 5, 2, (128,), "tanh")) == _hip.PATH_LMM
    assert _hip.cvae_kernel_path(_hip.CvaeShape.make(16, 4, 5, (128,), "tanh")) == _hip.PATH_LMM
    assert _hip.cvae_kernel_path(_hip.CvaeShape.make(16, 4, 2, (128,), "tanh", family="generic")) == _hip.PATH_GENERIC
    # a tile whose LDS image does not fit one CU stays on one thread per row (or is refused there)
    assert _hip.cvae_kernel_path(_hip.CvaeShape.make(10, 5, 10, (700, 700), "relu")) == _hip.PATH_GENERIC


LMM_SHAPES = [(5, 3, 2, (10,), "tanh"), (4, 2, 3, (7, 9), "relu"), (10, 5, 10, (128, 128), "relu"), (33, 0, 17, (64, 20, 40), "tanh"),
              (1, 0, 1, (1,), "tanh"), (70, 20, 6, (100,), "tanh"), (10, 5, 10, (256, 256), "relu")]


@pytest.mark.parametrize("d,c,lat,hidden,act", LMM_SHAPES)
@pytest.mark.parametrize("n", [1, 63, 1000])
def test_lmm_step_shapes_vs_oracle(d, c, lat, hidden, act, n):
    """the any-shape MFMA kernels (rnvp_lmm.hip) on the shapes the register-chained ones do not take -- several hidden
    layers, relu, wide layers, d > 16, latent > 4 -- and on ragged row tiles: loss, gradient, encoder, decoder ==
    float64 oracle; loss-only call == the loss of the gradient call; run-to-run bit-identical"""
    from oracle import CvaeOracle, CvaeShape
    from probaforms_amd import _hip
    rng = np.random.default_rng(d * 1000 + sum(hidden) + n)
    shape = _hip.CvaeShape.make(d, c, lat, hidden, act, family="lmm")
    assert _hip.cvae_kernel_path(shape) == _hip.PATH_LMM
    P = _hip.cvae_param_count(shape)
    p = (rng.standard_normal(P) * 0.2).astype(np.float32)
    X = rng.standard_normal((n, d)).astype(np.float32)
    C = rng.standard_normal((n, c)).astype(np.float32) if c else None
    eps = rng.standard_normal((n, lat)).astype(np.float32)
    o = CvaeOracle(64); so = CvaeShape.make(d, c, lat, hidden, act)
    lo, go = o.loss_grad(so, p, X, C, eps, 0.3)
    ws = torch.empty(_hip.cvae_workspace_bytes(shape, n), dtype=torch.uint8, device="cuda")
    grad = torch.full((P,), float("nan"), device="cuda"); loss = torch.empty(1, device="cuda")
    _hip.cvae_loss_grad(shape, _dev(p), _dev(X), _dev(C), None, _dev(eps), n, 1.0 / n, 0.3, grad, loss, ws)
    assert abs(float(loss) - lo) < 5e-6 * max(1.0, abs(lo))
    assert np.abs(grad.cpu().numpy() - go).max() < 5e-6 * np.abs(go).max()
    g2 = torch.empty_like(grad); l2 = torch.empty(1, device="cuda")
    _hip.cvae_loss_grad(shape, _dev(p), _dev(X), _dev(C), None, _dev(eps), n, 1.0 / n, 0.3, g2, l2, ws)
    assert torch.equal(grad, g2) and float(l2) == float(loss)
    l3 = torch.empty(1, device="cuda")
    _hip.cvae_loss_grad(shape, _dev(p), _dev(X), _dev(C), None, _dev(eps), n, 1.0 / n, 0.3, None, l3, ws)
    assert float(l3) == float(loss)
    mu = torch.empty(n, lat, device="cuda"); ls = torch.empty_like(mu); xr = torch.empty(n, d, device="cuda")
    _hip.cvae_encode(shape, _dev(p), _dev(X), _dev(C), n, mu, ls, ws)
    _hip.cvae_decode(shape, _dev(p), _dev(eps), _dev(C), n, xr, ws)
    mu_o, ls_o = o.encode(so, p, X, C); x_o = o.decode(so, p, eps, C)
    for got, want in ((mu, mu_o), (ls, ls_o), (xr, x_o)):
        assert np.abs(got.cpu().numpy() - want).max() < 5e-6 * max(1.0, np.abs(want).max())


def test_lmm_step_row_chunks_and_gather():
    """a batch larger than one row chunk of the weight-gradient operands (65536 rows), gathered through row_index, ==
    the sum of two half-batch calls (which each fit one chunk) and == the one-thread-per-row kernels"""
    from probaforms_amd import _hip
    d, c, lat, hidden, act, n = 4, 2, 3, (7, 9), "relu", 70001
    rng = np.random.default_rng(5)
    sl = _hip.CvaeShape.make(d, c, lat, hidden, act, family="lmm"); sv = _hip.CvaeShape.make(d, c, lat, hidden, act, family="generic")
    P = _hip.cvae_param_count(sl)
    p = _dev((rng.standard_normal(P) * 0.3).astype(np.float32))
    X = _dev(rng.standard_normal((n, d)).astype(np.float32)); C = _dev(rng.standard_normal((n, c)).astype(np.float32))
    eps = _dev(rng.standard_normal((n, lat)).astype(np.float32))
    idx = torch.from_numpy(rng.permutation(n).astype(np.int64)).cuda()
    ws = torch.empty(_hip.cvae_workspace_bytes(sl, n), dtype=torch.uint8, device="cuda")
    out = {}
    for name, shape in (("lmm", sl), ("valu", sv)):
        g = torch.empty(P + 1, device="cuda")
        _hip.cvae_loss_grad(shape, p, X, C, idx, eps, n, 1.0 / n, 0.5, g[:P], g[P:], ws)
        out[name] = g.cpu().numpy().astype(np.float64)
    h = 40000
    a = torch.empty(P + 1, device="cuda"); b = torch.empty(P + 1, device="cuda")
    _hip.cvae_loss_grad(sl, p, X, C, idx[:h].contiguous(), eps[:h].contiguous(), h, 1.0 / n, 0.5, a[:P], a[P:], ws)
    _hip.cvae_loss_grad(sl, p, X, C, idx[h:].contiguous(), eps[h:].contiguous(), n - h, 1.0 / n, 0.5, b[:P], b[P:], ws)
    halves = (a + b).cpu().numpy().astype(np.float64)
    scale = np.abs(out["valu"][:P]).max()
    assert np.abs(out["lmm"][:P] - halves[:P]).max() < 2e-5 * scale and abs(out["lmm"][P] - halves[P]) < 2e-5 * abs(halves[P])
    assert np.abs(out["lmm"][:P] - out["valu"][:P]).max() < 2e-5 * scale and abs(out["lmm"][P] - out["valu"][P]) < 2e-5 * abs(halves[P])


@pytest.mark.parametrize("d,c,lat,h", [(1, 0, 1, 1), (13, 1, 3, 37), (16, 2, 4, 200), (7, 4, 1, 16), (2, 2, 2, 300),
                                       (16, 4, 4, 128)])
@pytest.mark.parametrize("n", [1, 63, 257, 5000])
def test_mfma_step_shapes_vs_oracle(d, c, lat, h, n):
    """padded sizes, ragged row tiles, hidden widths past one flush block: MFMA step == float64 oracle"""
    from oracle import CvaeOracle, CvaeShape
    from probaforms_amd import _hip
    rng = np.random.default_rng(d * 1000 + h + n)
    shape = _hip.CvaeShape.make(d, c, lat, (h,), "tanh")
    assert _hip.cvae_kernel_path(shape) == _hip.PATH_MFMA
    P = _hip.cvae_param_count(shape)
    p = (rng.standard_normal(P) * 0.3).astype(np.float32)
    X = rng.standard_normal((n, d)).astype(np.float32)
    C = rng.standard_normal((n, c)).astype(np.float32) if c else None
    eps = rng.standard_normal((n, lat)).astype(np.float32)
    lo, go = CvaeOracle(64).loss_grad(CvaeShape.make(d, c, lat, (h,), "tanh"), p, X, C, eps, 0.3)
    ws = torch.empty(_hip.cvae_workspace_bytes(shape, n), dtype=torch.uint8, device="cuda")
    grad = torch.full((P,), float("nan"), device="cuda"); loss = torch.empty(1, device="cuda")
    _hip.cvae_loss_grad(shape, _dev(p), _dev(X), _dev(C), None, _dev(eps), n, 1.0 / n, 0.3, grad, loss, ws)
    assert abs(float(loss) - lo) < 3e-6 * max(1.0, abs(lo))
    assert np.abs(grad.cpu().numpy() - go).max() < 3e-6 * np.abs(go).max()
    g2 = torch.empty_like(grad)                                     # deterministic: run-to-run bit-identical
    _hip.cvae_loss_grad(shape, _dev(p), _dev(X), _dev(C), None, _dev(eps), n, 1.0 / n, 0.3, g2, loss, ws)
    assert torch.equal(grad, g2)
    # encoder / decoder alone on the MFMA blocks (workspace given) against the float64 oracle
    o = CvaeOracle(64); so = CvaeShape.make(d, c, lat, (h,), "tanh")
    mu = torch.empty(n, lat, device="cuda"); ls = torch.empty_like(mu); xr = torch.empty(n, d, device="cuda")
    _hip.cvae_encode(shape, _dev(p), _dev(X), _dev(C), n, mu, ls, ws)
    _hip.cvae_decode(shape, _dev(p), _dev(eps), _dev(C), n, xr, ws)
    mu_o, ls_o = o.encode(so, p, X, C); x_o = o.decode(so, p, eps, C)
    for got, want in ((mu, mu_o), (ls, ls_o), (xr, x_o)):
        assert np.abs(got.cpu().numpy() - want).max() < 3e-6 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("name", list(CASES))
def test_loss_grad_and_adam(name, path):
    from oracle import CvaeOracle, CvaeShape
    _hip, g, shape, klw, C = _load(name, family=path)
    d, c, lat, hidden, act, _ = CASES[name]
    n = g["X"].shape[0]; P = g["init_params"].size
    x, cc = _dev(g["X"]), _dev(C)
    ws = torch.empty(_hip.cvae_workspace_bytes(shape, n), dtype=torch.uint8, device="cuda")
    grad = torch.full((P,), float("nan"), device="cuda"); loss = torch.empty(1, device="cuda")
    _hip.cvae_loss_grad(shape, _dev(g["init_params"]), x, cc, None, _dev(g["eps"]), n, 1.0 / n, klw, grad, loss, ws)
    assert abs(float(loss) - g["loss"]) < 2e-6 * max(1.0, abs(float(g["loss"])))
    assert np.abs(grad.cpu().numpy() - g["grad"]).max() < 3e-6 * np.abs(g["grad"]).max() + 1e-9
    # loss only (per-epoch evaluation) + gathered / sharded batch vs the oracle
    l2 = torch.empty(1, device="cuda")
    _hip.cvae_loss_grad(shape, _dev(g["init_params"]), x, cc, None, _dev(g["eps"]), n, 1.0 / n, klw, None, l2, ws)
    assert float(l2) == float(loss)
    perm = np.random.default_rng(1).permutation(n).astype(np.int64)
    idx = torch.from_numpy(perm).cuda()
    a = torch.empty(P + 1, device="cuda"); b = torch.empty(P + 1, device="cuda"); k = 17
    eps = _dev(g["eps"])
    _hip.cvae_loss_grad(shape, _dev(g["init_params"]), x, cc, idx[:k], eps[:k], k, 1.0 / n, klw, a[:P], a[P:], ws)
    _hip.cvae_loss_grad(shape, _dev(g["init_params"]), x, cc, idx[k:].contiguous(), eps[k:].contiguous(), n - k, 1.0 / n, klw,
                        b[:P], b[P:], ws)
    o = CvaeOracle(32); s = CvaeShape.make(d, c, lat, hidden, act)
    lo, go = o.loss_grad(s, g["init_params"], g["X"][perm], None if C is None else C[perm], g["eps"], klw)
    tot = (a + b).cpu().numpy()
    assert abs(tot[P] - lo) < 5e-6 * max(1.0, abs(lo))
    assert np.abs(tot[:P] - go).max() < 5e-6 * np.abs(go).max() + 1e-9
    # 3 Adam steps against the reference's trajectory
    p = _dev(g["init_params"]).clone(); m = torch.zeros(P, device="cuda"); v = torch.zeros(P, device="cuda")
    for step in range(3):
        _hip.cvae_loss_grad(shape, p, x, cc, None, _dev(g["adam_eps"][step]), n, 1.0 / n, klw, grad, loss, ws)
        assert abs(float(loss) - g["adam_loss"][step]) < 5e-5 * max(1.0, abs(float(loss)))
        _hip.adam_step(p, grad, m, v, P, 0.01, 0.9, 0.999, 1e-8, 0.0, step + 1)
        assert np.abs(p.cpu().numpy() - g["adam_p"][step]).mean() < 2e-6
    # the fused step (cvae_train_step: optimizer inside the scatter kernel on the MFMA path) walks the same trajectory, bit for bit
    p2 = _dev(g["init_params"]).clone(); m2 = torch.zeros(P, device="cuda"); v2 = torch.zeros(P, device="cuda")
    g2 = torch.empty(P, device="cuda"); l2 = torch.empty(1, device="cuda")
    for step in range(3):
        _hip.cvae_train_step(shape, p2, x, cc, None, _dev(g["adam_eps"][step]), n, 1.0 / n, klw, g2, l2, m2, v2,
                             0.01, 0.9, 0.999, 1e-8, 0.0, step + 1, ws)
    assert torch.equal(p2, p) and torch.equal(m2, m) and torch.equal(v2, v)


@pytest.mark.parametrize("name", ["default", "nocond", "relu_mh"])
def test_seeded_fit_and_sample_match_reference(name):
    """same seed -> same init, same shuffle, same eps stream, same latent draws as the reference CVAE"""
    from probaforms_amd.models import CVAE
    d, c, lat, hidden, act, klw = CASES[name]
    g = np.load(os.path.join(GOLDEN, "cvae_%s.npz" % name))
    X, C = g["X"], (g["C"] if c else None)
    m = CVAE(latent_dim=lat, hidden=hidden, activation=act, KL_weight=klw, lr=0.01, n_epochs=3, batch_size=16)
    torch.manual_seed(5)
    assert m.fit(X, C) is m
    hist = np.array([float(v) for v in m.loss_history], np.float32)
    np.testing.assert_allclose(hist, g["fit_loss_history"], rtol=2e-4, atol=2e-5)
    xs = m.sample(C if C is not None else X.shape[0])
    assert xs.shape == g["fit_sample"].shape and xs.dtype == np.float32
    assert np.abs(xs - g["fit_sample"]).max() < 2e-3
    sd = m.encoder.state_dict()
    assert list(sd)[-4:] == ["mu.weight", "mu.bias", "log_sigma.weight", "log_sigma.bias"]
    m.fit(X, C)                                                      # re-initialises: history restarts
    assert len(m.loss_history) == 3


def test_api_contract_like_reference_tests():
    from probaforms_amd.models import CVAE
    n = 100
    X = np.random.normal(size=(n, 5)); C = np.random.normal(size=(n, 3))
    gen = CVAE(); gen.fit(X, C)
    assert gen.sample(C).shape == X.shape
    gen = CVAE(); gen.fit(X, C=None)
    assert gen.sample(C=n).shape == X.shape


def _c5_like(n, seed=0):
    rng = np.random.default_rng(seed)
    X = rng.normal(size=(n, 16)).astype(np.float32)
    C = (rng.random(size=(n, 4)) > 0.5).astype(np.float32)
    X[:, :4] += C
    return X, C


@pytest.mark.parametrize("n", [70_000, 70_001])
def test_lookahead_draws_replay_the_serial_generator_stream(n):
    """the worker-thread replay of fit's CPU-generator draws (seeds, per-batch eps, full-data eps; permutations on the
    shared pool) consumes the stream exactly like the serial loop: same values, same final generator state.  Since round 4
    the normals are drawn ON THE DEVICE with the host's bits where that is validated (nflow.HostStreamOnDevice; n = 70 001
    gives draws whose size is not a multiple of 16: torch's redrawn tail)"""
    from probaforms_amd.models.cvae import _FitDraws
    from probaforms_amd.models.nflow import HostStreamOnDevice
    from probaforms_amd._engine import batch_bounds, loader_permutation
    lat, epochs = 3, 3
    bounds = batch_bounds(n, 16_384)
    torch.manual_seed(11)
    serial = []
    for _ in range(epochs):
        perm = loader_permutation(n)
        eps = torch.cat([torch.randn(e - s, lat) for (s, e) in bounds])
        serial.append((perm, eps, torch.randn(n, lat)))
    end_state = torch.get_rng_state()
    torch.manual_seed(11)
    draws = _FitDraws(n, bounds, lat, epochs, torch.device("cuda"), slots=2)
    assert draws.on_device == HostStreamOnDevice.usable("cuda")
    for ref in serial:
        slot, *got = draws.next_epoch()
        # (the first epochs' permutations are drawn on the device where that is validated -- _engine.DeviceShuffle -- the rest on the pool)
        assert (got[0].is_cuda or got[0].is_pinned()) and torch.equal(got[0].cpu(), ref[0])
        for a, b in zip(got[1:], ref[1:]):
            assert (a.is_cuda if draws.on_device else a.is_pinned()) and torch.equal(a.cpu(), b)
        draws.release(slot)          # (everything was read back by .cpu() above: no event needed)
    draws.finish()
    assert torch.equal(torch.get_rng_state(), end_state)


@pytest.mark.parametrize("noise", ["host", "device"])
def test_large_fit_is_deterministic_and_learns(noise):
    from probaforms_amd.models import CVAE
    X, C = _c5_like(70_000)
    hists = []
    for _ in range(2):
        torch.manual_seed(3)
        m = CVAE(latent_dim=2, hidden=(64,), batch_size=8192, n_epochs=4, lr=3e-3, noise_rng=noise).fit(X, C)
        hists.append(np.array([float(v) for v in m.loss_history]))
        assert len(m.loss_history) == 4 and all(v.device.type == "cpu" and v.dim() == 0 for v in m.loss_history)
    assert np.array_equal(hists[0], hists[1])
    assert np.isfinite(hists[0]).all() and hists[0][-1] < hists[0][0]
    assert m.sample(C[:100]).shape == (100, 16)


def test_device_noise_slots_are_not_recycled_under_their_readers(monkeypatch):
    """with the reference's noise stream drawn on the device the eps buffers of an epoch are read in
    place by that epoch's kernels; a slot handed back behind its UPLOADS could be overwritten by the draw worker while those
    kernels were still queued.  A fit long enough to recycle all eight slots (16 epochs, each heavy enough for the GPU to fall
    behind the worker) must equal, bit for bit, the same fit with the noise drawn on the host."""
    from probaforms_amd.models import CVAE
    from probaforms_amd.models.nflow import HostStreamOnDevice
    if not HostStreamOnDevice.usable("cuda"):
        pytest.skip("this torch build's CPU randn is not the one the device restates")
    X, C = _c5_like(200_000)
    runs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("RNVP_HOST_PRIOR_ON_DEVICE", mode)
        torch.manual_seed(5)
        m = CVAE(latent_dim=2, hidden=(128,), batch_size=4096, n_epochs=16, lr=1e-3).fit(X, C)
        runs[mode] = (np.array([float(v) for v in m.loss_history]), m._core.flat.detach().cpu().numpy().copy(), torch.get_rng_state())
    assert np.array_equal(runs["1"][0], runs["0"][0])
    assert np.array_equal(runs["1"][1], runs["0"][1])
    assert torch.equal(runs["1"][2], runs["0"][2])


def test_device_noise_matches_host_noise_statistically():
    """the two noise sources train to the same loss level (different numbers, same distribution)"""
    from probaforms_amd.models import CVAE
    X, C = _c5_like(40_000, seed=1)
    out = {}
    for noise in ("host", "device"):
        torch.manual_seed(7)
        m = CVAE(latent_dim=2, hidden=(32,), batch_size=4096, n_epochs=6, lr=3e-3, noise_rng=noise).fit(X, C)
        out[noise] = float(m.loss_history[-1])
    assert abs(out["host"] - out["device"]) < 0.05 * abs(out["host"]), out
    with pytest.raises(ValueError):
        CVAE(noise_rng="gpu")


RESIDENT_CVAE = [
    # d, c, lat, hidden, act, n, batch, weight_decay
    (2, 1, 2, (10,), "tanh", 160, 32, 0.0),             # CVAE() defaults (cvae.py:145) on 2-d data with one condition
    (5, 3, 2, (10,), "tanh", 100, 32, 0.0),             # the reference's own test sizes; ragged last batch
    (5, 0, 2, (10,), "tanh", 75, 25, 0.2),              # no condition, ragged row tiles, weight decay
    (16, 4, 8, (16,), "relu", 300, 128, 0.0),           # 8 waves, the widest latent
    (16, 15, 3, (13,), "tanh", 70, 32, 0.0),            # 31 encoder inputs (two input tiles)
    (9, 6, 5, (32,), "tanh", 100, 48, 0.0),             # two hidden tiles
    (1, 0, 1, (3,), "tanh", 20, 7, 0.0),
    (2, 1, 2, (10,), "tanh", 5, 1, 0.0),                # one row per batch
]


@pytest.mark.parametrize("d,c,lat,hidden,act,n,batch,wd", RESIDENT_CVAE)
def test_resident_cvae_fit_epoch_vs_step_loop(d, c, lat, hidden, act, n, batch, wd):
    """cvae_fit_epoch on a model that fits one CU's LDS runs the epoch in one persistent workgroup (rnvp_resident.hip): the
    same trajectory as the batch-by-batch cvae_train_step loop (pinned by the oracle and the reference fixtures above) up to
    the rounding of another summation order, per-batch losses included; bit-identical run to run; family=generic pins the
    loop inside the same entry point"""
    from probaforms_amd import _hip
    rng = np.random.default_rng(d * 100 + lat * 10 + n)
    shape = _hip.CvaeShape.make(d, c, lat, hidden, act)
    loop_shape = _hip.CvaeShape.make(d, c, lat, hidden, act, family="generic")
    assert _hip.cvae_fit_epoch_resident(shape, batch) and not _hip.cvae_fit_epoch_resident(loop_shape, batch)
    assert not _hip.cvae_fit_epoch_resident(shape, 129)
    assert not _hip.cvae_fit_epoch_resident(_hip.CvaeShape.make(d, c, lat, (10, 10), act), batch)
    assert not _hip.cvae_fit_epoch_resident(_hip.CvaeShape.make(d, c, 9, hidden, act), batch)
    P = _hip.cvae_param_count(shape)
    p0 = (rng.uniform(-1, 1, P) * 0.4).astype(np.float32)
    x = _dev(rng.standard_normal((n, d)).astype(np.float32))
    cc = _dev(rng.standard_normal((n, c)).astype(np.float32)) if c else None
    perm = torch.from_numpy(rng.permutation(n).astype(np.int64)).cuda()
    eps = _dev(rng.standard_normal((2, n, lat)).astype(np.float32))
    nb = (n + batch - 1) // batch
    ws = torch.empty(_hip.cvae_workspace_bytes(shape, batch), dtype=torch.uint8, device="cuda")
    adam = (2e-3, 0.9, 0.999, 1e-8, wd)
    klw = 0.05

    def epochs(sh, manual):
        p = _dev(p0).clone(); m = torch.zeros(P, device="cuda"); v = torch.zeros(P, device="cuda")
        g = torch.empty(P, device="cuda"); hist = torch.full((2 * nb,), float("nan"), device="cuda")
        for ep in range(2):                           # the second epoch continues the optimizer's step count
            if not manual:
                _hip.cvae_fit_epoch(sh, p, x, cc, perm, eps[ep], n, batch, klw, g, hist[ep * nb:(ep + 1) * nb], m, v, *adam, 1 + ep * nb, ws)
                continue
            for k in range(nb):
                rows = min(batch, n - k * batch)
                _hip.cvae_train_step(sh, p, x, cc, perm[k * batch:k * batch + rows].contiguous(), eps[ep, k * batch:k * batch + rows].contiguous(),
                                     rows, 1.0 / rows, klw, g, hist[ep * nb + k:ep * nb + k + 1], m, v, *adam, 1 + ep * nb + k, ws)
        return [t.cpu().numpy().astype(np.float64) for t in (hist, p, m, v)]

    res, again, loop, manual = epochs(shape, False), epochs(shape, False), epochs(loop_shape, False), epochs(shape, True)
    for a, b in zip(res, again):
        assert np.array_equal(a, b)
    for a, b in zip(loop, manual):                    # the entry point's loop IS the per-batch calls (other kernels: tolerance)
        np.testing.assert_allclose(a, b, rtol=2e-4, atol=2e-6)
    assert np.isfinite(res[0]).all() and np.isfinite(res[1]).all()
    np.testing.assert_allclose(res[0], manual[0], rtol=2e-5, atol=2e-6)
    assert np.abs(res[1] - manual[1]).mean() < 2e-6 and np.abs(res[1] - manual[1]).max() < 2e-4
    assert np.abs(res[2] - manual[2]).max() < 1e-5 * max(1.0, np.abs(manual[2]).max())
    assert np.abs(res[3] - manual[3]).max() < 1e-5 * max(1.0, np.abs(manual[3]).max())
    assert np.abs(res[1] - p0).max() > 1e-3


@pytest.mark.parametrize("name", ["default", "nocond"])
def test_resident_cvae_adam_trajectory_vs_reference(name):
    """the reference's own 3-step trajectory (parameters and losses from the reference run with the recorded eps draws)
    through the resident CVAE kernel: one cvae_fit_epoch call per step on the fixture's batch"""
    _hip, g, shape, klw, C = _load(name)
    n = g["X"].shape[0]; P = g["init_params"].size
    assert _hip.cvae_fit_epoch_resident(shape, n)
    x, cc = _dev(g["X"]), _dev(C)
    p = _dev(g["init_params"]).clone(); m = torch.zeros(P, device="cuda"); v = torch.zeros(P, device="cuda")
    gb = torch.empty(P, device="cuda"); loss = torch.empty(1, device="cuda")
    ws = torch.empty(_hip.cvae_workspace_bytes(shape, n), dtype=torch.uint8, device="cuda")
    perm = torch.arange(n, device="cuda")
    for step in range(3):
        _hip.cvae_fit_epoch(shape, p, x, cc, perm, _dev(g["adam_eps"][step]), n, n, klw, gb, loss, m, v, 0.01, 0.9, 0.999, 1e-8, 0.0,
                            step + 1, ws)
        assert abs(float(loss) - g["adam_loss"][step]) < 5e-5 * max(1.0, abs(float(loss)))
        assert np.abs(p.cpu().numpy() - g["adam_p"][step]).mean() < 2e-6
