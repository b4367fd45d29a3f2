"""HIP kernels (through the C ABI, include/rnvp_hip.h) against the CPU oracle and the golden
fixtures produced by the reference.  Runs on the GPU box: `pytest -m gpu`."""
import numpy as np
import pytest
import torch
from cases import CASES, GRAD_STRIDE
from conftest import load_case, logp_mae_tol

pytestmark = pytest.mark.gpu

ALL = list(CASES)


def _dev(a, dtype=torch.float32):
    return None if a is None else torch.as_tensor(np.ascontiguousarray(a)).to(dtype).cuda().contiguous()


# masks declared alternating (register-chained MFMA path where it applies) / read from the table: the VALU kernels
# ("table") and the any-shape MFMA kernels with LDS-resident activations ("table/lmm", forced for every fixture)
PATHS = ["declared", "table", "table/lmm"]


# "declared/bx3": same masks declaration, GEMM1 of the forward / inverse kernels on the split-bf16 path with
# LDS-staged weights (rnvp_shape.precision = RNVP_PREC_BX3); "declared" pins RNVP_PREC_F32
FLOW_PATHS = ["declared", "declared/bx3", "table", "table/lmm"]


def _setup(name, path="declared"):
    from probaforms_amd import _hip
    cs = load_case(name)
    alt = _hip.RnvpShape.classify_masks(cs["masks"])
    assert alt == 1
    prec = "bx3" if path.endswith("/bx3") else "f32"
    lmm = path.endswith("/lmm")
    path = path.split("/")[0]
    # rnvp_shape.family (per call): "table" pins the VALU kernels, "table/lmm" asks for the any-shape MFMA kernels
    shape = _hip.RnvpShape.make(cs["L"], cs["d"], cs["c"], cs["hidden"], cs["act"], alt_masks=alt if path == "declared" else 0,
                                precision=prec, family="lmm" if lmm else ("valu" if path == "table" else "auto"))
    assert _hip.param_count(shape) == cs["params"].size
    if path == "table":
        assert _hip.kernel_path(shape, cs["masks"], _hip.OP_TRAIN) == (_hip.PATH_LMM if lmm else _hip.PATH_GENERIC)
    return _hip, cs, shape, _dev(cs["params"]), _dev(cs["masks"], torch.uint8)


def _ws(_hip, shape, op, n):
    nb = _hip.workspace_bytes(shape, op, n)
    return torch.empty(max(nb, 16), dtype=torch.uint8, device="cuda")


@pytest.mark.parametrize("path", FLOW_PATHS)
@pytest.mark.parametrize("name", ALL)
def test_forward_vs_golden_and_oracle(name, path, oracle32, oracle64):
    from oracle import Shape
    _hip, cs, shape, params, masks = _setup(name, path)
    n, d = cs["X"].shape
    x, c = _dev(cs["X"]), _dev(cs["C"])
    z = torch.empty(n, d, device="cuda"); ld = torch.empty(n, device="cuda")
    lp = torch.empty(n, device="cuda"); tot = torch.empty(1, device="cuda")
    _hip.forward_logprob(shape, params, masks, x, c, None, n, z, ld, lp, tot, _ws(_hip, shape, _hip.OP_FORWARD, n))
    torch.cuda.synchronize()
    g = cs["gold"]
    z, ld, lp, tot = z.cpu().numpy(), ld.cpu().numpy(), lp.cpu().numpy(), float(tot.item())
    # vs the reference's own outputs (golden): same bar as the oracle itself is held to
    assert np.abs(z - g["G2_z"]).mean() < 2e-6
    assert np.abs(lp - g["G2_logp"]).mean() < logp_mae_tol(name)
    assert abs(tot / n - g["G2_mean"]) < logp_mae_tol(name)
    np.testing.assert_allclose(ld, g["G2_layer_ld"].sum(0), rtol=1e-5, atol=1e-5 * max(1.0, np.abs(ld).max()))
    # vs the float64 referee: the HIP path must be no further from the truth than the reference is
    s = Shape.make(cs["L"], cs["d"], cs["c"], cs["hidden"], cs["act"])
    _, lp64, _ = oracle64.log_prob(s, cs["params"], cs["X"], cs["C"], cs["masks"])
    err_hip = np.abs(lp - lp64).mean(); err_ref = np.abs(g["G2_logp"] - lp64).mean()
    assert err_hip < max(3 * err_ref, 2e-6), (err_hip, err_ref)


@pytest.mark.parametrize("name", ALL)
def test_single_layer_f_and_g(name, oracle32):
    """L=1 calls are exactly RealNVPLayer.f / .g (realnvp.py:73-129)."""
    _hip, cs, shape, params, masks = _setup(name)
    g = cs["gold"]
    n, d = cs["X"].shape
    npl = cs["params"].size // cs["L"]
    c = _dev(cs["C"])
    for l in range(cs["L"]):
        one = _hip.RnvpShape.make(1, cs["d"], cs["c"], cs["hidden"], cs["act"], alt_masks=1 + (l & 1))
        xin = _dev(cs["X"] if l == 0 else g["G2_layer_out"][l - 1])
        y = torch.empty(n, d, device="cuda"); ld = torch.empty(n, device="cuda")
        _hip.forward_logprob(one, params[l * npl:(l + 1) * npl], masks[l], xin, c, None, n, y, ld, None, None,
                             _ws(_hip, one, _hip.OP_FORWARD, n))
        np.testing.assert_allclose(y.cpu().numpy(), g["G2_layer_out"][l], rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(ld.cpu().numpy(), g["G2_layer_ld"][l], rtol=2e-6, atol=2e-6)
    for k, l in enumerate(range(cs["L"] - 1, -1, -1)):
        one = _hip.RnvpShape.make(1, cs["d"], cs["c"], cs["hidden"], cs["act"], alt_masks=1 + (l & 1))
        zin = _dev(cs["Z"] if k == 0 else g["G3_layer_out"][k - 1])
        y = torch.empty(n, d, device="cuda")
        _hip.inverse(one, params[l * npl:(l + 1) * npl], masks[l], zin, c, n, y, _ws(_hip, one, _hip.OP_INVERSE, n))
        np.testing.assert_allclose(y.cpu().numpy(), g["G3_layer_out"][k], rtol=2e-6, atol=2e-6)


@pytest.mark.parametrize("path", FLOW_PATHS)
@pytest.mark.parametrize("name", ALL)
def test_inverse_and_roundtrip(name, path):
    _hip, cs, shape, params, masks = _setup(name, path)
    n, d = cs["Z"].shape
    zt, c = _dev(cs["Z"]), _dev(cs["C"])
    x = torch.empty(n, d, device="cuda")
    wsi = _ws(_hip, shape, _hip.OP_INVERSE, n)
    _hip.inverse(shape, params, masks, zt, c, n, x, wsi)
    np.testing.assert_allclose(x.cpu().numpy(), cs["gold"]["G3_x"], rtol=1e-5, atol=2e-5)
    # encode -> decode round trip, in place (x_out may alias z)
    xx = _dev(cs["X"]); z = torch.empty(n, d, device="cuda")
    _hip.forward_logprob(shape, params, masks, xx, c, None, n, z, None, None, None, _ws(_hip, shape, _hip.OP_FORWARD, n))
    _hip.inverse(shape, params, masks, z, c, n, z, wsi)
    assert (z - xx).abs().max().item() < max(2e-5, 10 * float(cs["gold"]["G3_roundtrip_maxerr"]))


@pytest.mark.parametrize("path", PATHS)
@pytest.mark.parametrize("name", ALL)
@pytest.mark.parametrize("tag,rows", [("G4", None), ("G8", 8)])
def test_loss_grad(name, tag, rows, path, oracle32):
    from oracle import Shape
    _hip, cs, shape, params, masks = _setup(name, path)
    g = cs["gold"]
    X = cs["X"][:rows]; C = None if cs["C"] is None else cs["C"][:rows]
    n = X.shape[0]
    grad = torch.full((cs["params"].size,), float("nan"), device="cuda"); loss = torch.empty(1, device="cuda")
    ws = _ws(_hip, shape, _hip.OP_TRAIN, n)
    _hip.loss_grad(shape, params, masks, _dev(X), _dev(C), None, n, 1.0 / n, grad, loss, ws)
    grad = grad.cpu().numpy(); loss = float(loss.item())
    # the loss is ONE float32 number of magnitude |loss|: allow 4 ulp of it on top of the log-prob MAE bar
    assert abs(loss - g[tag + "_loss"]) < max(logp_mae_tol(name), 5e-7 * abs(loss))
    if cs["wsrc"] == "torch":
        ref, got = g[tag + "_grad"], grad
    else:
        ref, got = g[tag + "_grad_sub"], grad[::GRAD_STRIDE]
        l2 = np.sqrt((grad.astype(np.float64) ** 2).sum())
        assert abs(l2 - g[tag + "_grad_l2"]) < 1e-5 * g[tag + "_grad_l2"]
    assert np.abs(got - ref).max() < 3e-6 * np.abs(ref).max() + 1e-9
    # full-vector check against the oracle (fixtures of the large shapes are subsampled)
    s = Shape.make(cs["L"], cs["d"], cs["c"], cs["hidden"], cs["act"])
    _, gor = oracle32.loss_grad(s, cs["params"], X, C, cs["masks"])
    assert np.abs(grad - gor).max() < 3e-6 * np.abs(gor).max() + 1e-9
    assert np.abs(grad[gor == 0]).max(initial=0.0) < 1e-9      # dead (masked) entries stay exactly ~0


@pytest.mark.parametrize("path", PATHS)
def test_loss_grad_gather_and_shards(path, oracle32):
    """row_index gather + two shards scaled by 1/B_global add up to the full-batch result."""
    from oracle import Shape
    _hip, cs, shape, params, masks = _setup("tm", path)
    n = cs["X"].shape[0]
    perm = np.random.default_rng(0).permutation(n).astype(np.int64)
    x, c, idx = _dev(cs["X"]), _dev(cs["C"]), _dev(perm, torch.int64)
    P = cs["params"].size
    ws = _ws(_hip, shape, _hip.OP_TRAIN, n)
    full = torch.empty(P + 1, device="cuda")
    _hip.loss_grad(shape, params, masks, x, c, idx, n, 1.0 / n, full[:P], full[P:], ws)
    a = torch.empty(P + 1, device="cuda"); b = torch.empty(P + 1, device="cuda")
    k = 40
    _hip.loss_grad(shape, params, masks, x, c, idx[:k], k, 1.0 / n, a[:P], a[P:], ws)
    _hip.loss_grad(shape, params, masks, x, c, idx[k:].contiguous(), n - k, 1.0 / n, b[:P], b[P:], ws)
    np.testing.assert_allclose((a + b).cpu().numpy(), full.cpu().numpy(), rtol=2e-5, atol=2e-6)
    s = Shape.make(cs["L"], cs["d"], cs["c"], cs["hidden"], cs["act"])
    lo, go = oracle32.loss_grad(s, cs["params"], cs["X"][perm], cs["C"][perm], cs["masks"])
    np.testing.assert_allclose(full[:P].cpu().numpy(), go, rtol=1e-4, atol=3e-6 * np.abs(go).max())
    # empty shard: zeros, no launch
    z = torch.full((P + 1,), 7.0, device="cuda")
    _hip.loss_grad(shape, params, masks, x, c, idx[:0], 0, 1.0 / n, z[:P], z[P:], ws)
    assert float(z.abs().max().item()) == 0.0


@pytest.mark.parametrize("path", PATHS)
@pytest.mark.parametrize("name", ["c1_L8", "tm", "d8"])
@pytest.mark.parametrize("wd", [0.0, 0.2])
def test_adam_trajectory_vs_reference(name, wd, path):
    """3 fused train steps against the reference's own Adam trajectory (G4)."""
    _hip, cs, shape, params, masks = _setup(name, path)
    g = cs["gold"]; k = "G4_adam_wd%g" % wd
    n = cs["X"].shape[0]; P = cs["params"].size
    x, c = _dev(cs["X"]), _dev(cs["C"])
    p = params.clone(); m = torch.zeros(P, device="cuda"); v = torch.zeros(P, device="cuda")
    gb = torch.empty(P, device="cuda"); loss = torch.empty(3, device="cuda")
    ws = _ws(_hip, shape, _hip.OP_TRAIN, n)
    for step in range(3):
        _hip.train_step(shape, p, masks, x, c, None, n, 1.0 / n, gb, loss[step:step + 1], m, v,
                        0.01, 0.9, 0.999, 1e-8, wd, step + 1, ws)
        mr, vr, pr = g[k + "_m"][step], g[k + "_v"][step], g[k + "_p"][step]
        np.testing.assert_allclose(m.cpu().numpy(), mr, rtol=2e-5, atol=3e-6 * np.abs(mr).max())
        np.testing.assert_allclose(v.cpu().numpy(), vr, rtol=4e-5, atol=6e-6 * np.abs(vr).max())
        assert np.abs(p.cpu().numpy() - pr).mean() < 2e-6
    np.testing.assert_allclose(loss.cpu().numpy(), g[k + "_loss"], rtol=5e-5, atol=5e-5)


def test_adam_kernel_matches_oracle_bit_exact(oracle32):
    rng = np.random.default_rng(5)
    for n in (1, 3, 4, 1001, 76032):
        p = rng.normal(size=n).astype(np.float32); g = rng.normal(size=n).astype(np.float32) * 1e-2
        m = rng.normal(size=n).astype(np.float32) * 1e-2; v = (rng.normal(size=n).astype(np.float32) * 1e-2) ** 2
        pd, gd, md, vd = (torch.from_numpy(a.copy()).cuda() for a in (p, g, m, v))
        from probaforms_amd import _hip
        _hip.adam_step(pd, gd, md, vd, n, 1e-3, 0.9, 0.999, 1e-8, 0.2, 7)
        oracle32.adam(p, g, m, v, 7, lr=1e-3, weight_decay=0.2)
        # same op order, no FMA contraction, IEEE sqrt/div on both sides: bit-exact
        assert np.array_equal(md.cpu().numpy(), m)
        assert np.array_equal(vd.cpu().numpy(), v)
        assert np.array_equal(pd.cpu().numpy(), p)


@pytest.mark.parametrize("name,prec", [("c2", "f32"), ("c2", "bx3"), ("c3", "bx3"), ("c4", "bx3"), ("c4", "f32")])
def test_large_batch_properties(name, prec):
    """Size-independent checks at a benchmark-sized batch: round trip and the log-prob identity
    logp == logdet - 0.5 (d ln 2pi + |z|^2), tiles spanning many blocks (both kernel families of the MFMA path)."""
    from probaforms_amd import _hip
    from cases import numpy_params
    L, d, c, hidden, act, _ = CASES[name]
    shape = _hip.RnvpShape.make(L, d, c, hidden, act, alt_masks=1, precision=prec)
    params = _dev(numpy_params(name)); masks = _dev(load_case(name)["masks"], torch.uint8)
    n = 200_003                                                    # ragged on purpose
    gen = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(n, d, device="cuda", generator=gen); cc = torch.randn(n, c, device="cuda", generator=gen)
    z = torch.empty_like(x); ld = torch.empty(n, device="cuda"); lp = torch.empty(n, device="cuda")
    tot = torch.empty(1, device="cuda")
    _hip.forward_logprob(shape, params, masks, x, cc, None, n, z, ld, lp, tot, _ws(_hip, shape, _hip.OP_FORWARD, n))
    ident = ld - 0.5 * (d * np.log(2 * np.pi) + (z.double() ** 2).sum(1)).float()
    assert (lp - ident).abs().max().item() < 2e-4 * (d / 16)
    assert abs(tot.item() - lp.double().sum().item()) < 1e-5 * abs(lp.double().sum().item())
    back = torch.empty_like(x)
    _hip.inverse(shape, params, masks, z, cc, n, back, _ws(_hip, shape, _hip.OP_INVERSE, n))
    assert (back - x).abs().max().item() < 5e-4 * (d / 16) and (back - x).abs().mean().item() < 2e-6 * (d / 16)


def test_fused_train_step_equals_loss_grad_plus_adam():
    """rnvp_train_step (optimizer fused into the gradient scatter on the MFMA path) is bit-identical to
    rnvp_loss_grad followed by rnvp_adam_step"""
    for name in ("c2", "tm"):
        _hip, cs, shape, params, masks = _setup(name)
        n = cs["X"].shape[0]; P = cs["params"].size
        x, c = _dev(cs["X"]), _dev(cs["C"])
        ws = _ws(_hip, shape, _hip.OP_TRAIN, n)
        pa, pb = params.clone(), params.clone()
        ma, va, mb, vb = (torch.zeros(P, device="cuda") for _ in range(4))
        ga, gb = torch.empty(P, device="cuda"), torch.empty(P, device="cuda")
        la, lb = torch.empty(1, device="cuda"), torch.empty(1, device="cuda")
        for step in (1, 2):
            _hip.train_step(shape, pa, masks, x, c, None, n, 1.0 / n, ga, la, ma, va, 0.01, 0.9, 0.999, 1e-8, 0.1, step, ws)
            _hip.loss_grad(shape, pb, masks, x, c, None, n, 1.0 / n, gb, lb, ws)
            _hip.adam_step(pb, gb, mb, vb, P, 0.01, 0.9, 0.999, 1e-8, 0.1, step)
            assert torch.equal(ga, gb) and torch.equal(la, lb)
            assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb)


@pytest.mark.parametrize("prec", ["f32", "bx3"])
@pytest.mark.parametrize("act", ["tanh", "relu"])
@pytest.mark.parametrize("L,d,c,h,n", [(3, 16, 4, 48, 37), (1, 32, 8, 16, 5), (2, 64, 16, 32, 100), (5, 16, 0, 16, 1),
                                       (3, 32, 8, 80, 300), (2, 7, 3, 200, 129), (2, 40, 9, 144, 513), (3, 24, 5, 272, 70)])
def test_mfma_path_edge_shapes_vs_oracle(L, d, c, h, n, act, prec, oracle32):
    """odd layer counts, hidden sizes that are not a multiple of the flush interval, ragged and tiny
    batches, gathered rows: MFMA kernels (forward, inverse, loss+grad) against the oracle"""
    from oracle import Shape
    from probaforms_amd import _hip
    rng = np.random.default_rng(L * 1000 + d + h)
    shape = _hip.RnvpShape.make(L, d, c, (h,), act, alt_masks=1, precision=prec)     # bx3: chunked stages (h = 144..272)
    assert _hip.kernel_path(shape, None, _hip.OP_TRAIN) == _hip.PATH_MFMA
    P = _hip.param_count(shape)
    params = (rng.uniform(-1, 1, size=P) * 0.12).astype(np.float32)      # ~ the default init range for these widths
    N = 3 * n + 7
    X = rng.normal(size=(N, d)).astype(np.float32); C = rng.normal(size=(N, c)).astype(np.float32) if c else None
    idx = rng.permutation(N)[:n].astype(np.int64)
    s = Shape.make(L, d, c, (h,), act)
    pd, xd, cd, id_ = _dev(params), _dev(X), _dev(C), _dev(idx, torch.int64)
    masks = _dev(((np.arange(d)[None] + np.arange(L)[:, None]) % 2).astype(np.uint8), torch.uint8)
    z = torch.empty(n, d, device="cuda"); lp = torch.empty(n, device="cuda"); tot = torch.empty(1, device="cuda")
    _hip.forward_logprob(shape, pd, masks, xd, cd, id_, n, z, None, lp, tot, _ws(_hip, shape, _hip.OP_FORWARD, n))
    Xg = X[idx]; Cg = None if C is None else C[idx]
    zo, lpo, _ = oracle32.log_prob(s, params, Xg, Cg)
    # random weights of this size amplify rounding through exp(s): bound the mean tightly, the max loosely
    assert np.abs(z.cpu().numpy() - zo).mean() < 2e-6 and np.abs(z.cpu().numpy() - zo).max() < 2e-4
    # |log p| reaches several hundred with these weights: the bar is 2 float32 ulp of that magnitude
    assert np.abs(lp.cpu().numpy() - lpo).mean() < max(1e-5, 2.4e-7 * np.abs(lpo).max())
    assert abs(float(tot) - lpo.astype(np.float64).sum()) < 2e-5 * max(1.0, np.abs(lpo).sum())
    back = torch.empty_like(z)
    _hip.inverse(shape, pd, masks, z, _dev(Cg), n, back, _ws(_hip, shape, _hip.OP_INVERSE, n))
    assert np.abs(back.cpu().numpy() - Xg).mean() < 2e-6 and np.abs(back.cpu().numpy() - Xg).max() < 5e-4
    grad = torch.empty(P, device="cuda"); loss = torch.empty(1, device="cuda")
    _hip.loss_grad(shape, pd, masks, xd, cd, id_, n, 1.0 / n, grad, loss, _ws(_hip, shape, _hip.OP_TRAIN, n))
    lo, go = oracle32.loss_grad(s, params, Xg, Cg)
    assert abs(float(loss) - lo) < max(1e-5, 5e-7 * abs(lo))
    assert np.abs(grad.cpu().numpy() - go).max() < 3e-6 * np.abs(go).max() + 1e-9


@pytest.mark.parametrize("n", [3000, 20000, 40000, 70000])
def test_loss_grad_linearity_across_row_tilings(n, oracle32):
    """The training kernel picks its row tiles per wave from the batch size (1, 2 or 4 on C2's geometry).
    Size-independent property: the gradient of a batch is the sum of the gradients of its parts (same
    1/B scale), whatever tiling each call used; the smallest case is also checked against the oracle."""
    from oracle import Shape
    from cases import numpy_params
    _hip, cs, shape, params, masks = _setup("c2")
    L, d, c, hidden, act, _ = CASES["c2"]
    P = cs["params"].size
    gen = torch.Generator(device="cuda").manual_seed(n)
    x = torch.randn(n, d, device="cuda", generator=gen); cc = torch.randn(n, c, device="cuda", generator=gen)
    ws = _ws(_hip, shape, _hip.OP_TRAIN, n)
    full = torch.empty(P + 1, device="cuda")
    _hip.loss_grad(shape, params, masks, x, cc, None, n, 1.0 / n, full[:P], full[P:], ws)
    acc = torch.zeros(P + 1, device="cuda", dtype=torch.float64)
    cuts = [0, 1000, n // 3, n]
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        part = torch.empty(P + 1, device="cuda")
        idx = torch.arange(lo, hi, device="cuda", dtype=torch.int64)
        _hip.loss_grad(shape, params, masks, x, cc, idx, hi - lo, 1.0 / n, part[:P], part[P:], ws)
        acc += part.double()
    scale = full[:P].abs().max().item()
    assert (full[:P].double() - acc[:P]).abs().max().item() < 2e-6 * scale
    assert abs(full[P].item() - acc[P].item()) < 2e-6 * abs(full[P].item())
    if n == 3000:
        lo_, go = oracle32.loss_grad(Shape.make(L, d, c, hidden, act), numpy_params("c2"), x.cpu().numpy(), cc.cpu().numpy())
        assert abs(full[P].item() - lo_) < max(1e-5, 5e-7 * abs(lo_))
        assert np.abs(full[:P].cpu().numpy() - go).max() < 3e-6 * np.abs(go).max() + 1e-9


def test_large_inverse_16m_rows():
    """BASELINE.json configs[3]-sized sampling call (16M draws, d=64, cond=16): index arithmetic beyond
    2^31 elements; checked through the round trip on a slice"""
    from probaforms_amd import _hip
    L, d, c, h = 8, 64, 16, 128
    n = 16 * 1024 * 1024 + 3
    shape = _hip.RnvpShape.make(L, d, c, (h,), "tanh", alt_masks=1)
    g = torch.Generator(device="cuda").manual_seed(0)
    params = (torch.rand(_hip.param_count(shape), device="cuda", generator=g) - 0.5) * 0.2
    masks = _dev(((np.arange(d)[None] + np.arange(L)[:, None]) % 2).astype(np.uint8), torch.uint8)
    z = torch.randn(n, d, device="cuda", generator=g); cc = torch.randn(n, c, device="cuda", generator=g)
    keep = z[-1000:].clone()
    _hip.inverse(shape, params, masks, z, cc, n, z, _ws(_hip, shape, _hip.OP_INVERSE, n))       # in place
    tail = z[-1000:].contiguous(); zz = torch.empty_like(tail)
    _hip.forward_logprob(shape, params, masks, tail, cc[-1000:].contiguous(), None, 1000, zz, None, None, None,
                         _ws(_hip, shape, _hip.OP_FORWARD, 1000))
    assert (zz - keep).abs().max().item() < 1e-3 and (zz - keep).abs().mean().item() < 5e-6


def test_calls_can_be_captured_in_a_hip_graph():
    """the entry points only enqueue work on the caller's stream (no allocation, no synchronisation):
    a training step + sampling captured once and replayed gives the same results as eager calls"""
    _hip, cs, shape, params, masks = _setup("c2")
    n = cs["X"].shape[0]; P = cs["params"].size
    x, c = _dev(cs["X"]), _dev(cs["C"])
    ws = _ws(_hip, shape, _hip.OP_TRAIN, n); wsi = _ws(_hip, shape, _hip.OP_INVERSE, n)
    zin = _dev(cs["Z"])

    def run(p, m, v, g, loss, xs, step):
        _hip.train_step(shape, p, masks, x, c, None, n, 1.0 / n, g, loss, m, v, 0.01, 0.9, 0.999, 1e-8, 0.0, step, ws)
        _hip.inverse(shape, p, masks, zin, c, n, xs, wsi)

    def fresh():
        return (params.clone(), torch.zeros(P, device="cuda"), torch.zeros(P, device="cuda"), torch.empty(P, device="cuda"),
                torch.empty(1, device="cuda"), torch.empty_like(zin))

    ref = fresh(); run(*ref, 1)                                       # eager (also warms up one-time setup)
    cap = fresh()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        run(*fresh(), 1)
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        run(*cap, 1)
    graph.replay(); torch.cuda.synchronize()
    for a, b in zip(ref, cap):
        assert torch.equal(a, b)


# ---- counter-based device prior (rnvp_prior_normal / rnvp_sample; nflow.py:141) ------------------------------
@pytest.mark.parametrize("d", [1, 5, 16, 64, 80])
def test_prior_normal_vs_oracle_and_row_offsets(d, oracle64):
    from probaforms_amd import _hip
    n, seed = 3001, 0x1234567890ABCDEF
    z = torch.empty(n, d, device="cuda")
    _hip.prior_normal(seed, 0, n, d, z)
    want = oracle64.prior_normal(seed, 0, n, d)
    assert np.abs(z.cpu().numpy() - want).max() < 2e-6            # same Philox words, Box-Muller to float rounding
    part = torch.empty(1000, d, device="cuda")
    _hip.prior_normal(seed, 2001, 1000, d, part)                  # rows 2001.. of the same stream: bit-identical
    assert torch.equal(part, z[2001:])
    big = torch.empty(7, d, device="cuda")
    _hip.prior_normal(seed, (1 << 33) + 5, 7, d, big)             # 64-bit row counters
    assert np.abs(big.cpu().numpy() - oracle64.prior_normal(seed, (1 << 33) + 5, 7, d)).max() < 2e-6


@pytest.mark.parametrize("path", FLOW_PATHS)
@pytest.mark.parametrize("name", ["c2", "c3", "c4", "tm", "tm_nocond", "reg1d", "relu_mh"])
def test_fused_sample_equals_prior_then_inverse(name, path, oracle32):
    """rnvp_sample (prior drawn inside the inverse kernel) == rnvp_inverse(rnvp_prior_normal) bit for bit, for any
    split of the rows; and matches the oracle's sample() on the oracle's own draw"""
    from oracle import Shape
    _hip, cs, shape, params, masks = _setup(name, path)
    if path.startswith("declared") and len(cs["hidden"]) > 1:
        pytest.skip("several hidden layers: same (generic) kernels as the table path")
    n, d, cdim, seed = 777, cs["d"], cs["c"], 99
    rng = np.random.default_rng(5)
    Cn = rng.standard_normal((n, cdim)).astype(np.float32) if cdim else None
    c = _dev(Cn)
    ws = _ws(_hip, shape, _hip.OP_INVERSE, n)
    z = torch.empty(n, d, device="cuda")
    _hip.prior_normal(seed, 0, n, d, z)
    ref = torch.empty_like(z)
    _hip.inverse(shape, params, masks, z, c, n, ref, ws)
    x = torch.empty_like(z)
    _hip.sample(shape, params, masks, c, n, seed, 0, x, ws)
    assert torch.equal(x, ref)
    lo = 300                                                       # two chunks with global row offsets
    xa = torch.empty(lo, d, device="cuda"); xb = torch.empty(n - lo, d, device="cuda")
    _hip.sample(shape, params, masks, None if c is None else c[:lo].contiguous(), lo, seed, 0, xa, ws)
    _hip.sample(shape, params, masks, None if c is None else c[lo:].contiguous(), n - lo, seed, lo, xb, ws)
    assert torch.equal(torch.cat([xa, xb]), ref)
    so = Shape.make(cs["L"], cs["d"], cs["c"], cs["hidden"], cs["act"])
    want = oracle32.sample(so, cs["params"], oracle32.prior_normal(seed, 0, n, d), Cn, cs["masks"])
    err = np.abs(x.cpu().numpy() - want)
    assert err.mean() < 5e-6 * max(1.0, np.abs(want).mean()), err.mean()


# ---- user masks, wide shapes, run-to-run determinism -----------------------------------------
def _rand_flow(L, d, c, hidden, act, seed, scale=0.3):
    from probaforms_amd import _hip
    sh = _hip.RnvpShape.make(L, d, c, hidden, act, alt_masks=0)
    rng = np.random.default_rng(seed)
    P = _hip.param_count(sh)
    return sh, (rng.uniform(-1, 1, P) * scale).astype(np.float32), rng


@pytest.mark.parametrize("family", ["valu", "lmm"])
@pytest.mark.parametrize("kind", ["blocks", "all_ones_layer", "random", "wide_d80_c20"])
def test_user_masks_and_wide_shapes_vs_oracle(kind, family, oracle32, oracle64):
    """RealNVPLayer(mask=...) accepts any {0,1} mask (realnvp.py:65-68): non-alternating tables, a layer whose mask is
    all ones (identity, log-det 0), and d > 64 / cdim > 16 run on the generic kernels -- forward, inverse, gradient"""
    from oracle import Shape
    from probaforms_amd import _hip
    if kind == "wide_d80_c20":
        L, d, c, hidden, act, n = 3, 80, 20, (24,), "tanh", 301
    else:
        L, d, c, hidden, act, n = 4, 6, 2, (9,), "relu" if kind == "random" else "tanh", 203
    sh, p, rng = _rand_flow(L, d, c, hidden, act, 11)
    if kind == "blocks":
        masks = np.array([[1, 1, 0, 0, 1, 0], [0, 0, 1, 1, 0, 1], [1, 0, 0, 0, 0, 1], [0, 1, 1, 1, 1, 0]], np.uint8)
    elif kind == "all_ones_layer":
        masks = np.array([[1, 0, 1, 0, 1, 0], [1, 1, 1, 1, 1, 1], [0, 1, 0, 1, 0, 1], [0, 0, 0, 0, 0, 0]], np.uint8)
    elif kind == "random":
        masks = rng.integers(0, 2, (L, d)).astype(np.uint8)
    else:
        masks = ((np.arange(d)[None] // 3 + np.arange(L)[:, None]) % 2).astype(np.uint8)
    assert _hip.RnvpShape.classify_masks(masks) == 0
    sh.family = _hip.FAMILIES[family]
    assert _hip.kernel_path(sh, masks, _hip.OP_TRAIN) == (_hip.PATH_LMM if family == "lmm" else _hip.PATH_GENERIC)
    X = rng.standard_normal((n, d)).astype(np.float32); C = rng.standard_normal((n, c)).astype(np.float32)
    so = Shape.make(L, d, c, hidden, act)
    params, mk, x, cc = _dev(p), _dev(masks, torch.uint8), _dev(X), _dev(C)
    z = torch.empty(n, d, device="cuda"); lp = torch.empty(n, device="cuda"); ld = torch.empty(n, device="cuda")
    _hip.forward_logprob(sh, params, mk, x, cc, None, n, z, ld, lp, None, _ws(_hip, sh, _hip.OP_FORWARD, n))
    z32, lp32, _ = oracle32.log_prob(so, p, X, C, masks)
    _, lp64, _ = oracle64.log_prob(so, p, X, C, masks)
    assert np.abs(z.cpu().numpy() - z32).max() < 2e-5 * max(1.0, np.abs(z32).max()) * (4 if d > 64 else 1)
    assert np.abs(lp.cpu().numpy() - lp64).mean() < max(3 * np.abs(lp32 - lp64).mean(), 1e-5)
    xb = torch.empty_like(z)
    _hip.inverse(sh, params, mk, z, cc, n, xb, _ws(_hip, sh, _hip.OP_INVERSE, n))
    assert (xb - x).abs().max().item() < 2e-4
    assert np.abs(xb.cpu().numpy() - oracle32.sample(so, p, z.cpu().numpy(), C, masks)).max() < 2e-5
    P = p.size
    g = torch.empty(P, device="cuda"); loss = torch.empty(1, device="cuda")
    _hip.loss_grad(sh, params, mk, x, cc, None, n, 1.0 / n, g, loss, _ws(_hip, sh, _hip.OP_TRAIN, n))
    lo, go = Oracle64Grad(oracle64, so, p, X, C, masks)
    assert abs(loss.item() - lo) < 1e-5 * max(1.0, abs(lo))
    assert np.abs(g.cpu().numpy() - go).max() < 3e-6 * np.abs(go).max() + 1e-9
    if kind == "all_ones_layer":     # nets of an identity layer get no gradient (exactly zero, as autograd gives)
        npl = P // L
        assert not g[npl:2 * npl].any().item()


def Oracle64Grad(oracle64, so, p, X, C, masks):
    lo, go = oracle64.loss_grad(so, p.astype(np.float64), X.astype(np.float64), C.astype(np.float64), masks)
    return float(lo), np.asarray(go, dtype=np.float64)


@pytest.mark.parametrize("n", [65536, 262144])
def test_train_step_is_bitwise_reproducible_at_bench_sizes(n):
    """no float atomics anywhere: the same rnvp_train_step twice gives identical bits (C2 shape; 65 536 rows runs the
    net-split kernel, 262 144 rows the four-wave one with several row groups per workgroup)"""
    from probaforms_amd import _hip
    L, d, c, hidden = 8, 16, 4, (128,)
    shape = _hip.RnvpShape.make(L, d, c, hidden, "tanh", alt_masks=1)
    P = _hip.param_count(shape)
    g0 = torch.Generator(device="cuda").manual_seed(3)
    p0 = (torch.rand(P, device="cuda", generator=g0) - 0.5) * 0.3
    x = torch.randn(n, d, device="cuda", generator=g0); cc = torch.randn(n, c, device="cuda", generator=g0)
    perm = torch.randperm(n, device="cuda", generator=g0)
    ws = _ws(_hip, shape, _hip.OP_TRAIN, n)
    outs = []
    for _ in range(2):
        p = p0.clone(); m = torch.zeros(P, device="cuda"); v = torch.zeros(P, device="cuda")
        g = torch.empty(P, device="cuda"); loss = torch.empty(2, device="cuda")
        for step in (1, 2):
            _hip.train_step(shape, p, None, x, cc, perm, n, 1.0 / n, g, loss[step - 1:step], m, v, 1e-3, 0.9, 0.999, 1e-8, 0.0, step, ws)
        outs.append((p, m, v, g, loss))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert torch.isfinite(outs[0][4]).all()


def test_unaligned_rows_take_the_guarded_path():
    """x / out at an odd float offset (a contiguous view that is not 16-byte aligned) must not fault: the kernels
    fall back to scalar row I/O and give the same values"""
    _hip, cs, shape, params, masks = _setup("c2")
    n, d = cs["X"].shape
    x, c = _dev(cs["X"]), _dev(cs["C"])
    buf = torch.zeros(n * d + 1, device="cuda"); xu = buf[1:].view(n, d); xu.copy_(x)
    assert xu.data_ptr() % 16 == 4 and xu.is_contiguous()
    z = torch.empty(n, d, device="cuda"); zu = torch.empty(n * d + 1, device="cuda")[1:].view(n, d)
    ws = _ws(_hip, shape, _hip.OP_FORWARD, n)
    _hip.forward_logprob(shape, params, masks, x, c, None, n, z, None, None, None, ws)
    _hip.forward_logprob(shape, params, masks, xu, c, None, n, zu, None, None, None, ws)
    assert torch.equal(z, zu)
    back = torch.empty(n * d + 1, device="cuda")[1:].view(n, d)
    _hip.inverse(shape, params, masks, zu, c, n, back, _ws(_hip, shape, _hip.OP_INVERSE, n))
    assert (back - x).abs().max().item() < 1e-4
    P = cs["params"].size
    g1 = torch.empty(P, device="cuda"); g2 = torch.empty(P, device="cuda"); l1 = torch.empty(1, device="cuda"); l2 = torch.empty(1, device="cuda")
    wt = _ws(_hip, shape, _hip.OP_TRAIN, n)
    _hip.loss_grad(shape, params, masks, x, c, None, n, 1.0 / n, g1, l1, wt)
    _hip.loss_grad(shape, params, masks, xu, c, None, n, 1.0 / n, g2, l2, wt)
    assert torch.equal(g1, g2) and torch.equal(l1, l2)


def _sweep_shapes():
    """seeded sweep over the padding boundaries of the MFMA geometries: d around 16 / 32 / 64, cdim around 0 / 4 / 8 / 16,
    hidden widths around the 16-wide tiles and the bx3 stage chunks, ragged row counts"""
    rng = np.random.default_rng(2024)
    ds = [1, 2, 3, 7, 15, 16, 17, 24, 31, 32, 33, 48, 63, 64]
    cs = [0, 1, 3, 4, 5, 8, 9, 15, 16]
    hs = [1, 5, 15, 16, 17, 33, 64, 100, 129, 200]
    ns = [1, 15, 16, 17, 63, 255, 256, 257, 1000, 2049]
    out = []
    for i in range(120):
        out.append((int(rng.integers(1, 5)), int(rng.choice(ds)), int(rng.choice(cs)), int(rng.choice(hs)), int(rng.choice(ns)),
                    ["tanh", "relu"][i % 2], ["f32", "bx3"][(i // 2) % 2]))
    return out


@pytest.mark.parametrize("L,d,c,h,n,act,prec", _sweep_shapes())
def test_shape_sweep_vs_oracle(L, d, c, h, n, act, prec, oracle32, oracle64):
    """forward (z, log-prob, sum), inverse, fused sampling and loss + gradient on randomly combined boundary shapes"""
    from oracle import Shape
    from probaforms_amd import _hip
    rng = np.random.default_rng(L * 7919 + d * 131 + c * 17 + h + n)
    shape = _hip.RnvpShape.make(L, d, c, (h,), act, alt_masks=1, precision=prec)
    assert _hip.kernel_path(shape, None, _hip.OP_FORWARD) == _hip.PATH_MFMA
    P = _hip.param_count(shape)
    params = (rng.uniform(-1, 1, size=P) * min(0.5, 1.5 / np.sqrt(h + d + c))).astype(np.float32)
    X = rng.standard_normal((n, d)).astype(np.float32); C = rng.standard_normal((n, c)).astype(np.float32) if c else None
    s = Shape.make(L, d, c, (h,), act)
    pd, xd, cd = _dev(params), _dev(X), _dev(C)
    z = torch.empty(n, d, device="cuda"); lp = torch.empty(n, device="cuda"); tot = torch.empty(1, device="cuda")
    _hip.forward_logprob(shape, pd, None, xd, cd, None, n, z, None, lp, tot, _ws(_hip, shape, _hip.OP_FORWARD, n))
    z64, lp64, _ = oracle64.log_prob(s, params, X, C)
    z32, lp32, _ = oracle32.log_prob(s, params, X, C)
    scale = max(1.0, float(np.abs(z64).max()))
    assert np.abs(z.cpu().numpy() - z64).max() < max(4 * np.abs(z32 - z64).max(), 4e-6 * scale)
    assert np.abs(lp.cpu().numpy() - lp64).mean() < max(3 * np.abs(lp32 - lp64).mean(), 3e-6 * max(1.0, np.abs(lp64).max()))
    assert abs(float(tot) - lp64.sum()) < 1e-5 * max(1.0, np.abs(lp64).sum())
    back = torch.empty_like(z)
    _hip.inverse(shape, pd, None, z, cd, n, back, _ws(_hip, shape, _hip.OP_INVERSE, n))
    assert np.abs(back.cpu().numpy() - X).max() < 5e-4 * max(1.0, np.abs(X).max()) * scale
    xs = torch.empty_like(z); zp = torch.empty_like(z)
    _hip.sample(shape, pd, None, cd, n, 5, 11, xs, _ws(_hip, shape, _hip.OP_INVERSE, n))
    _hip.prior_normal(5, 11, n, d, zp)
    ref = torch.empty_like(z)
    _hip.inverse(shape, pd, None, zp, cd, n, ref, _ws(_hip, shape, _hip.OP_INVERSE, n))
    assert torch.equal(xs, ref)
    grad = torch.empty(P, device="cuda"); loss = torch.empty(1, device="cuda")
    _hip.loss_grad(shape, pd, None, xd, cd, None, n, 1.0 / n, grad, loss, _ws(_hip, shape, _hip.OP_TRAIN, n))
    lo, go = oracle64.loss_grad(s, params.astype(np.float64), X.astype(np.float64), None if C is None else C.astype(np.float64))
    go = np.asarray(go, np.float64)
    assert abs(float(loss) - float(lo)) < 1e-5 * max(1.0, abs(float(lo)))
    assert np.abs(grad.cpu().numpy() - go).max() < 5e-6 * np.abs(go).max() + 1e-9


def test_auto_mode_takes_lmm_for_generic_shapes_and_is_bitwise_reproducible():
    """several hidden layers / user masks / wide rows go to the any-shape MFMA kernels by default; their training step
    (input-gradient chain in-kernel, weight gradients over fixed row splits) gives identical bits run to run"""
    from probaforms_amd import _hip
    for L, d, c, hidden, n in [(4, 6, 2, (12, 20), 5000), (3, 80, 20, (24,), 3000), (8, 16, 4, (128, 128), 20000)]:
        sh, p, rng = _rand_flow(L, d, c, hidden, "tanh", 5)
        masks = rng.integers(0, 2, (L, d)).astype(np.uint8)
        for op in (_hip.OP_FORWARD, _hip.OP_INVERSE, _hip.OP_TRAIN):
            assert _hip.kernel_path(sh, masks, op) == _hip.PATH_LMM
        P = p.size
        x = _dev(rng.standard_normal((n, d)).astype(np.float32)); cc = _dev(rng.standard_normal((n, c)).astype(np.float32))
        mk = _dev(masks, torch.uint8)
        ws = _ws(_hip, sh, _hip.OP_TRAIN, n)
        outs = []
        for _ in range(2):
            pp = _dev(p); m = torch.zeros(P, device="cuda"); v = torch.zeros(P, device="cuda")
            g = torch.empty(P, device="cuda"); loss = torch.empty(2, device="cuda")
            for step in (1, 2):
                _hip.train_step(sh, pp, mk, x, cc, None, n, 1.0 / n, g, loss[step - 1:step], m, v, 1e-3, 0.9, 0.999, 1e-8, 0.0, step, ws)
            outs.append((pp, m, v, g, loss))
        for a, b in zip(*outs):
            assert torch.equal(a, b)
        assert torch.isfinite(outs[0][4]).all()
    # a shape whose 16-row LDS image does not fit falls back to the one-thread-per-row kernels
    big = _hip.RnvpShape.make(2, 16, 4, (512, 512, 512), "tanh", alt_masks=0)
    assert _hip.kernel_path(big, None, _hip.OP_TRAIN) == _hip.PATH_GENERIC


def test_lmm_row_chunks_add_up():
    """the 16-row lmm training pass works through a big batch in row chunks (bounded workspace: the dumped weight-gradient
    operands take 38 KB per row for hidden=(128,128)); the chunked gradient equals the sum of separately computed parts
    (the 64-row form's chunks: tests/test_lmm64_gpu.py)"""
    from probaforms_amd import _hip
    L, d, c, hidden, n = 8, 16, 4, (128, 128), 60000            # chunk = 28160 rows -> 3 chunks
    sh, p, rng = _rand_flow(L, d, c, hidden, "tanh", 9, scale=0.15)
    sh.family = _hip.FAMILIES["lmm16"]
    masks = ((np.arange(d)[None] + np.arange(L)[:, None]) % 2).astype(np.uint8)
    assert _hip.kernel_path(sh, masks, _hip.OP_TRAIN) == _hip.PATH_LMM
    assert _hip.workspace_bytes(sh, _hip.OP_TRAIN, 10 ** 7) == _hip.workspace_bytes(sh, _hip.OP_TRAIN, 10 ** 6) < 2 ** 31
    P = p.size
    gen = torch.Generator(device="cuda").manual_seed(4)
    x = torch.randn(n, d, device="cuda", generator=gen); cc = torch.randn(n, c, device="cuda", generator=gen)
    perm = torch.randperm(n, device="cuda", generator=gen)
    pd, mk = _dev(p), _dev(masks, torch.uint8)
    ws = _ws(_hip, sh, _hip.OP_TRAIN, n)
    full = torch.empty(P + 1, device="cuda")
    _hip.loss_grad(sh, pd, mk, x, cc, perm, n, 1.0 / n, full[:P], full[P:], ws)
    again = torch.empty(P + 1, device="cuda")
    _hip.loss_grad(sh, pd, mk, x, cc, perm, n, 1.0 / n, again[:P], again[P:], ws)
    assert torch.equal(full, again)
    acc = torch.zeros(P + 1, device="cuda", dtype=torch.float64)
    part = torch.empty(P + 1, device="cuda")
    for lo in range(0, n, 20000):
        _hip.loss_grad(sh, pd, mk, x, cc, perm[lo:lo + 20000].contiguous(), 20000, 1.0 / n, part[:P], part[P:], ws)
        acc += part.double()
    scale = acc[:P].abs().max().item()
    assert (full[:P].double() - acc[:P]).abs().max().item() < 3e-6 * scale
    assert abs(full[P].item() - acc[P].item()) < 1e-5 * abs(acc[P].item())
    # without a row index the chunks walk the arrays directly
    direct = torch.empty(P + 1, device="cuda")
    xg, cg = x[perm].contiguous(), cc[perm].contiguous()
    _hip.loss_grad(sh, pd, mk, xg, cg, None, n, 1.0 / n, direct[:P], direct[P:], ws)
    assert torch.equal(direct, full)


@pytest.mark.parametrize("n", [1, 16, 17, 32, 33, 64, 1000, 4096, 4097, 8192])
@pytest.mark.parametrize("d,c,h,act", [(16, 4, 128, "tanh"), (16, 4, 200, "relu"), (5, 0, 17, "tanh"), (2, 1, 64, "relu"),
                                       (13, 3, 40, "tanh"), (32, 8, 100, "tanh"), (24, 5, 256, "tanh"), (64, 16, 64, "tanh"),
                                       (50, 0, 130, "relu")])
def test_tile_split_small_batches_vs_oracle(d, c, h, act, n, oracle64):
    """batches of up to 8192 rows (d <= 16; 4096 for wider rows) of a flow with three or more hidden tiles run the
    tile-split training kernel (16 or 32 rows per workgroup, hidden tiles spread over its waves): loss + gradient against the float64 oracle, bitwise
    repeatable, and the same numbers through a row gather.  (ReLU cases stay small: among ~10M pre-activations one lands
    within float32 rounding of zero and flips against the float64 oracle -- the kink, not the kernel.)"""
    from oracle import Shape
    from probaforms_amd import _hip
    L = 5
    rng = np.random.default_rng(d * 1000 + h * 10 + n)
    shape = _hip.RnvpShape.make(L, d, c, (h,), act, alt_masks=1)
    assert _hip.kernel_path(shape, None, _hip.OP_TRAIN) == _hip.PATH_MFMA
    P = _hip.param_count(shape)
    params = (rng.uniform(-1, 1, size=P) * min(0.5, 1.5 / np.sqrt(h + d + c))).astype(np.float32)
    X = rng.standard_normal((n, d)).astype(np.float32); C = rng.standard_normal((n, c)).astype(np.float32) if c else None
    s = Shape.make(L, d, c, (h,), act)
    pd, xd, cd = _dev(params), _dev(X), _dev(C)
    ws = _ws(_hip, shape, _hip.OP_TRAIN, n)
    grad = torch.empty(P, device="cuda"); loss = torch.empty(1, device="cuda")
    _hip.loss_grad(shape, pd, None, xd, cd, None, n, 1.0 / n, grad, loss, ws)
    lo, go = oracle64.loss_grad(s, params.astype(np.float64), X.astype(np.float64), None if C is None else C.astype(np.float64))
    go = np.asarray(go, np.float64)
    assert abs(float(loss) - float(lo)) < 1e-5 * max(1.0, abs(float(lo)))
    assert np.abs(grad.cpu().numpy() - go).max() < 5e-6 * np.abs(go).max() + 1e-9
    g2 = torch.full((P,), float("nan"), device="cuda"); l2 = torch.empty(1, device="cuda")
    _hip.loss_grad(shape, pd, None, xd, cd, None, n, 1.0 / n, g2, l2, ws)
    assert torch.equal(grad, g2) and torch.equal(loss, l2)
    # gathered rows (row_index) give the same numbers as the contiguous copy
    perm = torch.randperm(n, device="cuda")
    inv = torch.empty_like(perm); inv[perm] = torch.arange(n, device="cuda")
    xs = xd[perm].contiguous(); cs_ = None if cd is None else cd[perm].contiguous()
    g3 = torch.empty(P, device="cuda"); l3 = torch.empty(1, device="cuda")
    _hip.loss_grad(shape, pd, None, xs, cs_, inv, n, 1.0 / n, g3, l3, ws)
    assert torch.equal(grad, g3) and torch.equal(loss, l3)


@pytest.mark.parametrize("n", [1, 17, 1000, 4096])
@pytest.mark.parametrize("d,c,h,act,prec", [(16, 4, 128, "tanh", "auto"), (5, 0, 40, "relu", "f32"), (32, 8, 256, "tanh", "auto"),
                                            (64, 16, 100, "relu", "auto"), (50, 3, 48, "tanh", "f32")])
def test_small_calls_latency_mode_vs_oracle(d, c, h, act, prec, n, oracle32, oracle64):
    """rnvp_shape.small_calls = RNVP_SMALL_LATENCY: forward (z, log-prob, sum), inverse and fused sampling of short calls on
    the tile-split kernels against the oracle; the default mode on the same rows agrees to float32 rounding; an explicit
    precision = bx3 keeps the bx3 kernels"""
    from oracle import Shape
    from probaforms_amd import _hip
    L = 4
    rng = np.random.default_rng(d * 977 + h * 13 + n)
    lat = _hip.RnvpShape.make(L, d, c, (h,), act, alt_masks=1, precision=prec, small_calls=_hip.SMALL_CALLS["latency"])
    inv = _hip.RnvpShape.make(L, d, c, (h,), act, alt_masks=1, precision=prec)
    P = _hip.param_count(lat)
    params = (rng.uniform(-1, 1, size=P) * min(0.5, 1.5 / np.sqrt(h + d + c))).astype(np.float32)
    X = rng.standard_normal((n, d)).astype(np.float32); C = rng.standard_normal((n, c)).astype(np.float32) if c else None
    s = Shape.make(L, d, c, (h,), act)
    pd, xd, cd = _dev(params), _dev(X), _dev(C)
    out = {}
    for name, shape in (("latency", lat), ("invariant", inv)):
        z = torch.empty(n, d, device="cuda"); lp = torch.empty(n, device="cuda"); tot = torch.empty(1, device="cuda")
        _hip.forward_logprob(shape, pd, None, xd, cd, None, n, z, None, lp, tot, _ws(_hip, shape, _hip.OP_FORWARD, n))
        back = torch.empty_like(z)
        _hip.inverse(shape, pd, None, z, cd, n, back, _ws(_hip, shape, _hip.OP_INVERSE, n))
        xs = torch.empty_like(z); zp = torch.empty_like(z); ref = torch.empty_like(z)
        _hip.sample(shape, pd, None, cd, n, 5, 11, xs, _ws(_hip, shape, _hip.OP_INVERSE, n))
        _hip.prior_normal(5, 11, n, d, zp)
        _hip.inverse(shape, pd, None, zp, cd, n, ref, _ws(_hip, shape, _hip.OP_INVERSE, n))
        assert torch.equal(xs, ref)
        out[name] = (z.cpu().numpy(), lp.cpu().numpy(), float(tot), back.cpu().numpy(), xs.cpu().numpy())
    z64, lp64, _ = oracle64.log_prob(s, params, X, C)
    z32, lp32, _ = oracle32.log_prob(s, params, X, C)
    scale = max(1.0, float(np.abs(z64).max()))
    z, lp, tot, back, xs = out["latency"]
    assert np.abs(z - z64).max() < max(4 * np.abs(z32 - z64).max(), 4e-6 * scale)
    assert np.abs(lp - lp64).mean() < max(3 * np.abs(lp32 - lp64).mean(), 3e-6 * max(1.0, np.abs(lp64).max()))
    assert abs(tot - lp64.sum()) < 1e-5 * max(1.0, np.abs(lp64).sum())
    assert np.abs(back - X).max() < 5e-4 * max(1.0, np.abs(X).max()) * scale
    zi, lpi, _, _, xsi = out["invariant"]
    assert np.abs(z - zi).max() < 1e-5 * scale and np.abs(xs - xsi).max() < 2e-4 * max(1.0, np.abs(xsi).max())
    pinned = _hip.RnvpShape.make(L, d, c, (h,), act, alt_masks=1, precision="bx3", small_calls=_hip.SMALL_CALLS["latency"])
    zb = torch.empty(n, d, device="cuda"); zc = torch.empty(n, d, device="cuda")
    invb = _hip.RnvpShape.make(L, d, c, (h,), act, alt_masks=1, precision="bx3")
    _hip.forward_logprob(pinned, pd, None, xd, cd, None, n, zb, None, None, None, _ws(_hip, pinned, _hip.OP_FORWARD, n))
    _hip.forward_logprob(invb, pd, None, xd, cd, None, n, zc, None, None, None, _ws(_hip, invb, _hip.OP_FORWARD, n))
    assert torch.equal(zb, zc)


# ---- out-of-bounds canaries: guard bands around every output buffer ----------------------------------
_CANARY = 0x7fc0dead          # a quiet-NaN bit pattern no kernel produces


class _Guarded:
    """a float32 output buffer of `numel` elements with 64 poisoned elements on either side (16-byte aligned interior)"""

    def __init__(self, numel):
        self.full = torch.full((numel + 128,), 0, dtype=torch.int32, device="cuda")
        self.full.fill_(_CANARY)
        self.view = self.full[64:64 + numel].view(torch.float32)
        self.numel = numel

    def check(self, what):
        lo, hi = self.full[:64], self.full[64 + self.numel:]
        assert bool((lo == _CANARY).all()) and bool((hi == _CANARY).all()), "write outside %s" % what


@pytest.mark.parametrize("L,d,c,hidden,n,family", [
    (8, 16, 4, (128,), 200, "auto"), (8, 16, 4, (128,), 70001, "auto"), (3, 5, 3, (10,), 33, "auto"), (2, 1, 1, (10,), 17, "auto"),
    (12, 32, 8, (256,), 1000, "auto"), (4, 64, 16, (128,), 4097, "auto"), (4, 33, 0, (40,), 129, "auto"),
    (3, 6, 2, (7, 9), 203, "lmm"), (3, 6, 2, (7, 9), 203, "valu"), (3, 80, 20, (24,), 301, "auto"),
    (3, 6, 2, (7, 9), 203, "lmm64"), (3, 80, 20, (24,), 301, "lmm64"), (4, 16, 4, (64, 48), 9001, "auto")])
def test_no_kernel_writes_outside_its_output_buffers(L, d, c, hidden, n, family):
    """z_out, logdet/logp, x_out, grad_out, gx_out and loss_hist sit between poisoned guard bands; every entry point runs
    (forward, inverse, fused sampling, loss + gradient, backward, one fitted epoch) and the bands must be untouched"""
    from probaforms_amd import _hip
    rng = np.random.default_rng(L + d + n)
    masks = ((np.arange(d)[None] + np.arange(L)[:, None]) % 2).astype(np.uint8)
    alt = 1 if len(hidden) == 1 and d <= 64 and c <= 16 and family == "auto" else 0
    shape = _hip.RnvpShape.make(L, d, c, hidden, "tanh", alt_masks=alt, family=family)
    P = _hip.param_count(shape)
    params = _dev((rng.uniform(-1, 1, P) * 0.2).astype(np.float32)); mk = _dev(masks, torch.uint8)
    x = _dev(rng.standard_normal((n, d)).astype(np.float32))
    cc = _dev(rng.standard_normal((n, c)).astype(np.float32)) if c else None
    z, ld, lp, tot = _Guarded(n * d), _Guarded(n), _Guarded(n), _Guarded(1)
    _hip.forward_logprob(shape, params, mk, x, cc, None, n, z.view.view(n, d), ld.view, lp.view, tot.view, _ws(_hip, shape, _hip.OP_FORWARD, n))
    xb = _Guarded(n * d)
    _hip.inverse(shape, params, mk, z.view.view(n, d), cc, n, xb.view.view(n, d), _ws(_hip, shape, _hip.OP_INVERSE, n))
    xs = _Guarded(n * d)
    _hip.sample(shape, params, mk, cc, n, 3, 5, xs.view.view(n, d), _ws(_hip, shape, _hip.OP_INVERSE, n))
    g, loss = _Guarded(P), _Guarded(1)
    _hip.loss_grad(shape, params, mk, x, cc, None, n, 1.0 / n, g.view, loss.view, _ws(_hip, shape, _hip.OP_TRAIN, n))
    g2, gx = _Guarded(P), _Guarded(n * d)
    gz = torch.randn(n, d, device="cuda"); gld = torch.randn(n, device="cuda")
    _hip.backward(shape, params, mk, x, cc, None, n, gz, gld, g2.view, gx.view.view(n, d), _ws(_hip, shape, _hip.OP_TRAIN, n))
    bs = max(1, n // 3)
    nb = (n + bs - 1) // bs
    hist, gbuf = _Guarded(nb), _Guarded(P)
    p2, m1, m2 = _Guarded(P), _Guarded(P), _Guarded(P)
    p2.view.copy_(params); m1.view.zero_(); m2.view.zero_()
    perm = torch.randperm(n, device="cuda")
    _hip.fit_epoch(shape, p2.view, mk, x, cc, perm, n, bs, gbuf.view, hist.view, m1.view, m2.view, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1,
                   _ws(_hip, shape, _hip.OP_TRAIN, bs))
    torch.cuda.synchronize()
    for buf, what in ((z, "z_out"), (ld, "logdet_out"), (lp, "logp_out"), (tot, "logp_sum"), (xb, "x_out (inverse)"),
                      (xs, "x_out (sample)"), (g, "grad_out"), (loss, "loss_out"), (g2, "grad_out (backward)"),
                      (gx, "gx_out"), (hist, "loss_hist"), (gbuf, "grad_buf"), (p2, "params"), (m1, "exp_avg"), (m2, "exp_avg_sq")):
        buf.check(what)
    for buf in (z, lp, xb, xs, g, g2, gx, hist):
        assert bool(torch.isfinite(buf.view).all())


def test_backward_entry_point_vs_loss_grad_and_oracle(oracle64):
    """rnvp_backward seeded with the loss's own seeds (gz = z / B, gld = -1 / B) equals rnvp_loss_grad, on every kernel
    family; gx_out against a central finite difference of the float64 oracle's loss"""
    from oracle import Shape
    from probaforms_amd import _hip
    for (L, d, c, hidden, fam, n) in [(8, 16, 4, (128,), "auto", 100), (8, 16, 4, (128,), "auto", 20000), (4, 6, 2, (12, 20), "lmm", 77),
                                      (4, 6, 2, (12, 20), "valu", 77), (6, 32, 8, (64,), "auto", 300)]:
        alt = 1 if len(hidden) == 1 else 0
        shape = _hip.RnvpShape.make(L, d, c, hidden, "tanh", alt_masks=alt, family=fam)
        rng = np.random.default_rng(d + n)
        masks = ((np.arange(d)[None] + np.arange(L)[:, None]) % 2).astype(np.uint8)
        P = _hip.param_count(shape)
        p = (rng.uniform(-1, 1, P) * min(0.2, 1.0 / np.sqrt(max(hidden) + d + c))).astype(np.float32)
        X = rng.standard_normal((n, d)).astype(np.float32); C = rng.standard_normal((n, c)).astype(np.float32)
        pd, mk, xd, cd = _dev(p), _dev(masks, torch.uint8), _dev(X), _dev(C)
        z = torch.empty(n, d, device="cuda")
        _hip.forward_logprob(shape, pd, mk, xd, cd, None, n, z, None, None, None, _ws(_hip, shape, _hip.OP_FORWARD, n))
        g1 = torch.empty(P, device="cuda"); loss = torch.empty(1, device="cuda")
        _hip.loss_grad(shape, pd, mk, xd, cd, None, n, 1.0 / n, g1, loss, _ws(_hip, shape, _hip.OP_TRAIN, n))
        g2 = torch.empty(P, device="cuda"); gx = torch.empty(n, d, device="cuda")
        _hip.backward(shape, pd, mk, xd, cd, None, n, (z / n).contiguous(), torch.full((n,), -1.0 / n, device="cuda"), g2, gx,
                      _ws(_hip, shape, _hip.OP_TRAIN, n))
        scale = float(g1.abs().max())
        assert float((g1 - g2).abs().max()) < 5e-6 * scale, (hidden, fam, n)       # the seeds differ by one rounding (z * inv_B in-kernel)
        if n <= 100:      # d loss / d x[r][j] by finite differences of the float64 oracle
            so = Shape.make(L, d, c, hidden, "tanh")
            gxh = gx.cpu().numpy()
            for (r, j) in [(0, 0), (3, d - 1), (n - 1, d // 2)]:
                e = 1e-4
                Xp, Xm = X.astype(np.float64).copy(), X.astype(np.float64).copy()
                Xp[r, j] += e; Xm[r, j] -= e
                lp_, _ = oracle64.loss_grad(so, p.astype(np.float64), Xp, C.astype(np.float64), masks=masks if not alt else None)
                lm_, _ = oracle64.loss_grad(so, p.astype(np.float64), Xm, C.astype(np.float64), masks=masks if not alt else None)
                fd = (float(lp_) - float(lm_)) / (2 * e)
                assert abs(gxh[r, j] - fd) < 2e-5 * max(1.0, np.abs(gxh).max()) + 1e-7, (hidden, fam, r, j, gxh[r, j], fd)


@pytest.mark.parametrize("L,d,c,h,n", [(8, 16, 4, 128, 20000), (4, 32, 8, 64, 9000), (3, 12, 3, 40, 16500)])
def test_training_forward_on_split_bf16_keeps_the_gradient_tolerance(L, d, c, h, n, oracle64):
    """rnvp_shape.precision = bx3 also moves GEMM1 of the training kernel's forward phase to split-bf16 MFMA (row-parallel
    launches: more than 8192 rows): loss and gradient against the float64 oracle at the tolerance of the f32 kernel"""
    from oracle import Shape
    from probaforms_amd import _hip
    rng = np.random.default_rng(d + n)
    s = Shape.make(L, d, c, (h,), "tanh")
    X = rng.standard_normal((n, d)).astype(np.float32); C = rng.standard_normal((n, c)).astype(np.float32)
    res = {}
    for prec in ("f32", "bx3"):
        shape = _hip.RnvpShape.make(L, d, c, (h,), "tanh", alt_masks=1, precision=prec)
        P = _hip.param_count(shape)
        if prec == "f32":
            params = (rng.uniform(-1, 1, size=P) * min(0.5, 1.5 / np.sqrt(h + d + c))).astype(np.float32)
        grad = torch.empty(P, device="cuda"); loss = torch.empty(1, device="cuda")
        _hip.loss_grad(shape, _dev(params), None, _dev(X), _dev(C), None, n, 1.0 / n, grad, loss, _ws(_hip, shape, _hip.OP_TRAIN, n))
        res[prec] = (float(loss), grad.cpu().numpy())
    lo, go = oracle64.loss_grad(s, params.astype(np.float64), X.astype(np.float64), C.astype(np.float64))
    go = np.asarray(go, np.float64)
    for prec, (l, g) in res.items():
        assert abs(l - float(lo)) < 1e-5 * max(1.0, abs(float(lo))), prec
        assert np.abs(g - go).max() < 5e-6 * np.abs(go).max() + 1e-9, prec
    assert not np.array_equal(res["f32"][1], res["bx3"][1])          # two different kernels did run


RESIDENT_CASES = [
    # L, d, c, hidden, act, n, batch, weight_decay, user_masks
    (8, 2, 1, (10,), "tanh", 160, 32, 0.0, False),            # the reference's defaults on a 2-d sample with one condition
    (8, 2, 0, (10,), "tanh", 100, 32, 0.0, False),            # ragged last batch (4 rows), no condition
    (4, 5, 3, (10,), "tanh", 75, 25, 0.2, False),             # ragged row tiles; weight decay
    (2, 16, 4, (16,), "relu", 300, 64, 0.0, False),           # 4 waves, two input tiles
    (2, 3, 1, (10,), "tanh", 300, 128, 0.0, False),           # 8 waves
    (2, 16, 15, (13,), "tanh", 70, 32, 0.0, False),           # the widest input: 31 columns
    (3, 9, 6, (32,), "tanh", 100, 32, 0.0, False),            # two hidden tiles, 15 inputs
    (16, 3, 2, (20,), "tanh", 64, 16, 0.0, False),            # 16 layers, two hidden tiles (one partly filled)
    (5, 7, 2, (9,), "relu", 90, 7, 0.0, True),                # user masks, a batch smaller than one tile
    (6, 13, 2, (17,), "tanh", 200, 48, 0.0, True),
    (2, 1, 1, (10,), "tanh", 17, 5, 0.0, False),
    (8, 2, 1, (10, 10), "tanh", 160, 32, 0.0, False),         # two hidden layers (k_fit_resident_deep)
    (4, 5, 3, (10, 16, 7), "tanh", 75, 25, 0.2, False),       # three, ragged tiles, weight decay
    (3, 16, 4, (16, 16), "relu", 200, 64, 0.0, False),        # full tiles, two input tiles
    (5, 3, 1, (5, 9, 4), "relu", 90, 96, 0.0, True),          # user masks, one batch larger than the data
    (16, 2, 0, (3, 2), "tanh", 40, 8, 0.0, False),
    (8, 2, 1, (10, 20, 15), "tanh", 96, 32, 0.0, False),      # the docstring network (realnvp.py:22-38): a two-tile hidden layer
    (3, 5, 3, (32, 32), "relu", 70, 32, 0.1, False),          # two full tiles per hidden layer
    (2, 16, 4, (17, 9), "tanh", 50, 25, 0.0, True),
    (2, 2, 1, (10,), "tanh", 5, 1, 0.0, False),               # one row per batch
    (2, 3, 0, (4, 4), "relu", 3, 1, 0.1, True),
    (4, 8, 3, (16,), "relu", 96, 32, 0.1, True),              # net-split form (rnvp_resident_ns.hip): a full hidden tile, 12 input columns
    (3, 4, 2, (12,), "tanh", 60, 17, 0.0, False),             # its second row tile holds one row; the widest trimmed hidden tile
    (8, 2, 1, (32,), "tanh", 64, 32, 0.0, False),             # two hidden tiles on a one-k-step input
    (4, 3, 1, (10,), "tanh", 100, 40, 0.1, False),            # three row tiles (all eight waves walk the chain), then a two-tile batch
    (8, 2, 1, (10,), "relu", 200, 64, 0.0, True),             # four row tiles, user masks, a ragged one-tile batch at the end
]


@pytest.mark.parametrize("L,d,c,hidden,act,n,batch,wd,user_masks", RESIDENT_CASES)
def test_resident_fit_epoch_vs_step_loop(L, d, c, hidden, act, n, batch, wd, user_masks):
    """rnvp_fit_epoch on a model that fits one CU's LDS runs the whole epoch in one persistent workgroup
    (rnvp_resident.hip); it must walk the same trajectory as the batch-by-batch rnvp_train_step loop (the kernels the
    oracle and the reference fixtures pin) up to the rounding of a different summation order, and reproduce itself
    bit for bit"""
    from probaforms_amd import _hip
    rng = np.random.default_rng(L * 100 + d * 10 + n)
    if user_masks:
        masks = (rng.random((L, d)) < 0.5).astype(np.uint8); alt = 0
    else:
        masks = ((np.arange(d)[None] + np.arange(L)[:, None]) % 2).astype(np.uint8); alt = 1 if len(hidden) == 1 else 0
    shape = _hip.RnvpShape.make(L, d, c, hidden, act, alt_masks=alt)
    assert _hip.fit_epoch_resident(shape, batch)
    assert not _hip.fit_epoch_resident(_hip.RnvpShape.make(L, d, c, hidden, act, alt_masks=alt, family="valu"), batch)
    assert not _hip.fit_epoch_resident(shape, 129) and not _hip.fit_epoch_resident(_hip.RnvpShape.make(L, d, c, (33,), act), batch)
    assert not _hip.fit_epoch_resident(_hip.RnvpShape.make(L, d, c, (10, 33), act), batch)            # at most two tiles per hidden layer
    assert not _hip.fit_epoch_resident(_hip.RnvpShape.make(L, d, c, (4, 4, 4, 4), act), batch)
    assert not _hip.fit_epoch_resident(_hip.RnvpShape.make(17, d, c, hidden, act), batch)
    P = _hip.param_count(shape)
    p0 = (rng.uniform(-1, 1, P) * 0.3).astype(np.float32)
    x = _dev(rng.standard_normal((n, d)).astype(np.float32))
    cc = _dev(rng.standard_normal((n, c)).astype(np.float32)) if c else None
    mk = _dev(masks, torch.uint8)
    perm = torch.from_numpy(rng.permutation(n).astype(np.int64)).cuda()
    nb = (n + batch - 1) // batch
    ws = _ws(_hip, shape, _hip.OP_TRAIN, batch)
    adam = (2e-3, 0.9, 0.999, 1e-8, wd)

    def epoch_resident(first_step, p, m, v):
        hist = torch.full((nb,), float("nan"), device="cuda"); gbuf = torch.empty(P, device="cuda")
        _hip.fit_epoch(shape, p, mk, x, cc, perm, n, batch, gbuf, hist, m, v, *adam, first_step, ws)
        return hist

    def epoch_loop(first_step, p, m, v):
        hist = torch.full((nb,), float("nan"), device="cuda"); gbuf = torch.empty(P, device="cuda")
        for k in range(nb):
            rows = min(batch, n - k * batch)
            _hip.train_step(shape, p, mk, x, cc, perm[k * batch:k * batch + rows].contiguous(), rows, 1.0 / rows, gbuf, hist[k:k + 1], m, v,
                            *adam, first_step + k, ws)
        return hist

    def epoch_nullmasks(first_step, p, m, v):        # alt_masks declared: the masks pointer may be NULL
        hist = torch.full((nb,), float("nan"), device="cuda"); gbuf = torch.empty(P, device="cuda")
        _hip.fit_epoch(shape, p, None, x, cc, perm, n, batch, gbuf, hist, m, v, *adam, first_step, ws)
        return hist

    out = {}
    runs = [("resident", epoch_resident), ("loop", epoch_loop), ("again", epoch_resident)] + ([("nullmasks", epoch_nullmasks)] if alt else [])
    for name, fn in runs:
        p = _dev(p0).clone(); m = torch.zeros(P, device="cuda"); v = torch.zeros(P, device="cuda")
        h1 = fn(1, p, m, v)
        h2 = fn(1 + nb, p, m, v)                      # a second epoch continues the optimizer's step count
        out[name] = [t.cpu().numpy().astype(np.float64) for t in (torch.cat([h1, h2]), p, m, v)]
    for a, b in zip(out["resident"], out["again"]):
        assert np.array_equal(a, b)
    if alt:
        for a, b in zip(out["resident"], out["nullmasks"]):
            assert np.array_equal(a, b)
    hr, pr, mr, vr = out["resident"]; hl, pl, ml, vl = out["loop"]
    assert np.isfinite(hr).all() and np.isfinite(pr).all()
    np.testing.assert_allclose(hr, hl, rtol=2e-5, atol=2e-5)
    assert np.abs(pr - pl).mean() < 2e-6 and np.abs(pr - pl).max() < 2e-4
    assert np.abs(mr - ml).max() < 1e-5 * max(1.0, np.abs(ml).max())
    assert np.abs(vr - vl).max() < 1e-5 * max(1.0, np.abs(vl).max())
    assert np.abs(pr - p0).max() > 1e-3            # it did train


@pytest.mark.parametrize("L,d,c,hidden,batch", [(8, 2, 1, (10,), 32), (4, 5, 3, (10, 12), 25), (2, 16, 4, (128,), 64),
                                                   (4, 2, 1, (10,), 48), (3, 8, 7, (16,), 32)])
def test_fit_epochs_equals_consecutive_fit_epoch_calls(L, d, c, hidden, batch):
    """rnvp_fit_epochs (several epochs, one library call; one persistent launch where the model is LDS-resident) walks
    exactly the trajectory of one rnvp_fit_epoch call per epoch, bit for bit, on the resident kernels and on the loop"""
    from probaforms_amd import _hip
    rng = np.random.default_rng(L + d)
    n, E = 90, 3
    masks = _dev(((np.arange(d)[None] + np.arange(L)[:, None]) % 2).astype(np.uint8), torch.uint8)
    shape = _hip.RnvpShape.make(L, d, c, hidden, "tanh", alt_masks=1 if len(hidden) == 1 else 0)
    P = _hip.param_count(shape)
    p0 = (rng.uniform(-1, 1, P) * 0.3).astype(np.float32)
    x = _dev(rng.standard_normal((n, d)).astype(np.float32)); cc = _dev(rng.standard_normal((n, c)).astype(np.float32))
    perms = torch.from_numpy(np.stack([rng.permutation(n) for _ in range(E)]).astype(np.int64)).cuda()
    nb = (n + batch - 1) // batch
    ws = _ws(_hip, shape, _hip.OP_TRAIN, batch)
    out = []
    for one_call in (True, False):
        p = _dev(p0).clone(); m = torch.zeros(P, device="cuda"); v = torch.zeros(P, device="cuda")
        g = torch.empty(P, device="cuda"); hist = torch.full((E, nb), float("nan"), device="cuda")
        if one_call:
            _hip.fit_epochs(shape, p, masks, x, cc, perms, n, batch, E, g, hist, m, v, 2e-3, 0.9, 0.999, 1e-8, 0.0, 1, ws)
        else:
            for e in range(E):
                _hip.fit_epoch(shape, p, masks, x, cc, perms[e], n, batch, g, hist[e], m, v, 2e-3, 0.9, 0.999, 1e-8, 0.0, 1 + e * nb, ws)
        out.append((hist.clone(), p.clone(), m.clone(), v.clone()))
    for a, b in zip(*out):
        assert torch.equal(a, b)
    assert bool(torch.isfinite(out[0][0]).all())


@pytest.mark.parametrize("name", ["c1_L4", "c1_L8", "tm", "tm_nocond", "reg1d", "d8", "relu_sh", "tanh_mh", "relu_mh"])
@pytest.mark.parametrize("wd", [0.0, 0.2])
def test_resident_adam_trajectory_vs_reference(name, wd):
    """the reference's own 3-step Adam trajectory on one batch (G4: parameters, both moments, losses from the reference run)
    through the resident fit: three one-batch epochs in ONE launch (rnvp_fit_epochs), every resident kernel form (one
    hidden layer of one and two tiles, two hidden layers, tanh / relu, with and without a condition, d = 1)"""
    from probaforms_amd import _hip
    cs = load_case(name)
    g = cs["gold"]; k = "G4_adam_wd%g" % wd
    if k + "_p" not in g:
        pytest.skip("fixture holds no Adam trajectory for this case")
    n = cs["X"].shape[0]
    alt = _hip.RnvpShape.classify_masks(cs["masks"])
    shape = _hip.RnvpShape.make(cs["L"], cs["d"], cs["c"], cs["hidden"], cs["act"], alt_masks=alt)
    assert _hip.fit_epoch_resident(shape, n)
    P = cs["params"].size
    x, c = _dev(cs["X"]), _dev(cs["C"])
    p = _dev(cs["params"]).clone(); m = torch.zeros(P, device="cuda"); v = torch.zeros(P, device="cuda")
    gb = torch.empty(P, device="cuda"); loss = torch.empty(3, device="cuda")
    perms = torch.arange(n, device="cuda").repeat(3).contiguous()
    _hip.fit_epochs(shape, p, _dev(cs["masks"], torch.uint8), x, c, perms, n, n, 3, gb, loss, m, v, 0.01, 0.9, 0.999, 1e-8, wd, 1,
                    _ws(_hip, shape, _hip.OP_TRAIN, n))
    mr, vr, pr = g[k + "_m"][2], g[k + "_v"][2], g[k + "_p"][2]
    np.testing.assert_allclose(m.cpu().numpy(), mr, rtol=2e-5, atol=3e-6 * np.abs(mr).max())
    np.testing.assert_allclose(v.cpu().numpy(), vr, rtol=4e-5, atol=6e-6 * np.abs(vr).max())
    assert np.abs(p.cpu().numpy() - pr).mean() < 2e-6
    np.testing.assert_allclose(loss.cpu().numpy(), g[k + "_loss"], rtol=5e-5, atol=5e-5)


def test_epoch_entry_points_argument_checks():
    """the epoch-level entry points refuse what they cannot run (status codes, nothing launched) and treat empty work as a no-op"""
    from probaforms_amd import _hip
    shape = _hip.RnvpShape.make(2, 2, 1, (10,), "tanh", alt_masks=1)
    P = _hip.param_count(shape)
    p = torch.zeros(P, device="cuda"); m = torch.zeros(P, device="cuda"); v = torch.zeros(P, device="cuda"); g = torch.empty(P, device="cuda")
    x = torch.randn(8, 2, device="cuda"); c = torch.randn(8, 1, device="cuda"); perm = torch.arange(8, device="cuda")
    hist = torch.full((4,), 7.0, device="cuda"); ws = _ws(_hip, shape, _hip.OP_TRAIN, 4)
    adam = (1e-3, 0.9, 0.999, 1e-8, 0.0)
    _hip.fit_epochs(shape, p, None, x, c, perm, 8, 4, 0, g, hist, m, v, *adam, 1, ws)          # no epochs: nothing happens
    _hip.fit_epoch(shape, p, None, x, c, perm, 0, 4, g, hist, m, v, *adam, 1, ws)               # no rows
    assert float(hist.min()) == 7.0 and float(p.abs().max()) == 0.0
    for bad in (dict(batch=0), dict(first=0), dict(hist=None), dict(perm=None), dict(x=None), dict(m=None)):
        with pytest.raises(RuntimeError):
            _hip.fit_epochs(shape, p, None, bad.get("x", x), c, bad.get("perm", perm), 8, bad.get("batch", 4), 1, g, bad.get("hist", hist),
                            bad.get("m", m), v, *adam, bad.get("first", 1), ws)
    cs = _hip.CvaeShape.make(2, 1, 2, (10,), "tanh")
    Pc = _hip.cvae_param_count(cs)
    pc = torch.zeros(Pc, device="cuda"); mc = torch.zeros(Pc, device="cuda"); vc = torch.zeros(Pc, device="cuda"); gc = torch.empty(Pc, device="cuda")
    eps = torch.randn(8, 2, device="cuda"); wsc = torch.empty(_hip.cvae_workspace_bytes(cs, 4), dtype=torch.uint8, device="cuda")
    _hip.cvae_fit_epoch(cs, pc, x, c, perm, eps, 0, 4, 0.001, gc, hist, mc, vc, *adam, 1, wsc)   # no rows
    for bad in (dict(batch=0), dict(first=0), dict(hist=None), dict(eps=None), dict(perm=None)):
        with pytest.raises(RuntimeError):
            _hip.cvae_fit_epoch(cs, pc, x, c, bad.get("perm", perm), bad.get("eps", eps), 8, bad.get("batch", 4), 0.001, gc, bad.get("hist", hist),
                                mc, vc, *adam, bad.get("first", 1), wsc)
    assert float(hist.min()) == 7.0
