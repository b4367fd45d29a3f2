"""The any-shape training kernel on 64-row blocks with in-kernel weight gradients (rnvp_lmm64.hip, `family="lmm64"`; what
`auto` runs from 8192 rows per call on) against the float64 oracle, against the 16-row form it replaces, run to run, across
row chunks, and through `rnvp_backward`.  /root/reference/probaforms/models/realnvp.py:22-38,91-101,246-250."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev(a, dtype=torch.float32):
    return None if a is None else torch.as_tensor(np.ascontiguousarray(a)).to(dtype).cuda().contiguous()


def _ws(_hip, shape, op, n):
    return torch.empty(max(_hip.workspace_bytes(shape, op, n), 16), dtype=torch.uint8, device="cuda")


def _flow(L, d, c, hidden, act, seed, family, scale=None):
    from probaforms_amd import _hip
    sh = _hip.RnvpShape.make(L, d, c, hidden, act, alt_masks=0, family=family)
    rng = np.random.default_rng(seed)
    P = _hip.param_count(sh)
    if scale is None:
        scale = min(0.3, 1.5 / np.sqrt(max(hidden) + d + c))
    return sh, (rng.uniform(-1, 1, P) * scale).astype(np.float32), rng


def _masks(kind, L, d, rng):
    if kind == "alt":
        return ((np.arange(d)[None] + np.arange(L)[:, None]) % 2).astype(np.uint8)
    if kind == "blocks":
        return ((np.arange(d)[None] // 3 + np.arange(L)[:, None]) % 2).astype(np.uint8)
    m = rng.integers(0, 2, (L, d)).astype(np.uint8)
    m[0] = 1                                            # an identity layer: its nets get exactly zero gradient
    return m


def _loss_grad(_hip, sh, p, masks, X, C, row_index=None, n=None):
    n = X.shape[0] if n is None else n
    P = p.size
    g = torch.empty(P + 1, device="cuda")
    _hip.loss_grad(sh, _dev(p), _dev(masks, torch.uint8), _dev(X), _dev(C) if C is not None and C.shape[1] else None, row_index, n,
                   1.0 / n, g[:P], g[P:], _ws(_hip, sh, _hip.OP_TRAIN, n))
    return g


SHAPES = [
    # L, d, c, hidden, act, masks, n
    (4, 6, 2, (9,), "tanh", "blocks", 203),
    (4, 6, 2, (9,), "relu", "random", 64),
    (4, 6, 2, (12, 20), "tanh", "random", 65),
    (3, 2, 0, (10, 20, 15), "tanh", "alt", 1000),         # the reference docstring's network, no condition
    (3, 80, 20, (24,), "tanh", "blocks", 301),            # wide rows: 64-outputs-by-16-inputs units do not apply (100 inputs)
    (2, 5, 3, (70, 33), "relu", "random", 130),           # ragged tiles on every side
    (8, 16, 4, (128, 128), "tanh", "alt", 2000),
    (2, 16, 4, (128,), "tanh", "random", 777),            # one hidden layer with user masks
    (2, 3, 1, (4,), "tanh", "alt", 1),
]


@pytest.mark.parametrize("L,d,c,hidden,act,mk,n", SHAPES)
def test_lmm64_loss_grad_vs_float64_oracle_and_the_16_row_form(L, d, c, hidden, act, mk, n, oracle64):
    from oracle import Shape
    from probaforms_amd import _hip
    sh, p, rng = _flow(L, d, c, hidden, act, 7 + d + n, "lmm64")
    masks = _masks(mk, L, d, rng)
    X = rng.standard_normal((n, d)).astype(np.float32); C = rng.standard_normal((n, c)).astype(np.float32)
    assert _hip.kernel_path(sh, masks, _hip.OP_TRAIN) == _hip.PATH_LMM
    g = _loss_grad(_hip, sh, p, masks, X, C)
    assert _hip.last_dispatch(_hip.PROFILE_TRAIN)["kernel"] == "k_lmm_train64"
    again = _loss_grad(_hip, sh, p, masks, X, C)
    assert torch.equal(g, again)                          # partials summed in a fixed order: identical bits run to run
    so = Shape.make(L, d, c, hidden, act)
    lo, go = oracle64.loss_grad(so, p.astype(np.float64), X.astype(np.float64), C.astype(np.float64), masks)
    go = np.asarray(go, np.float64)
    gh = g.cpu().numpy().astype(np.float64)
    P = p.size
    assert abs(gh[P] - float(lo)) < 1e-5 * max(1.0, abs(float(lo)))
    assert np.abs(gh[:P] - go).max() < 3e-6 * np.abs(go).max() + 1e-9
    if mk == "random":                                    # layer 0 is the identity: exactly zero, as autograd gives
        assert not g[:P // L].any().item()
    sh16, _, _ = _flow(L, d, c, hidden, act, 0, "lmm16")
    g16 = _loss_grad(_hip, sh16, p, masks, X, C)
    assert _hip.last_dispatch(_hip.PROFILE_TRAIN)["kernel"] == "k_lmm_train"
    assert (g - g16).abs().max().item() < 3e-6 * g16[:P].abs().max().item() + 1e-9


def test_auto_takes_the_64_row_form_from_8192_rows_on():
    from probaforms_amd import _hip
    L, d, c, hidden = 8, 16, 4, (128, 128)
    sh, p, rng = _flow(L, d, c, hidden, "tanh", 3, "auto")
    masks = _masks("alt", L, d, rng)
    for n, kernel in ((4096, "k_lmm_train"), (8192, "k_lmm_train64"), (20000, "k_lmm_train64")):
        X = rng.standard_normal((n, d)).astype(np.float32); C = rng.standard_normal((n, c)).astype(np.float32)
        g = _loss_grad(_hip, sh, p, masks, X, C)
        assert _hip.last_dispatch(_hip.PROFILE_TRAIN)["kernel"] == kernel
        assert torch.isfinite(g).all()
    # a net whose weight-gradient units exceed a wave's register slots keeps the 16-row form
    big, pb, _ = _flow(2, 16, 4, (192, 192), "tanh", 3, "lmm64")
    X = rng.standard_normal((9000, d)).astype(np.float32); C = rng.standard_normal((9000, c)).astype(np.float32)
    _loss_grad(_hip, big, pb, _masks("alt", 2, d, rng), X, C)
    assert _hip.last_dispatch(_hip.PROFILE_TRAIN)["kernel"] == "k_lmm_train"


def test_lmm64_row_chunks_and_gathered_rows_add_up():
    """a call of more than 262144 rows goes through in chunks on one grid (bounded workspace); the chunked gradient equals the sum
    of separately computed parts, gathered rows give the bits of the direct walk"""
    from probaforms_amd import _hip
    L, d, c, hidden, n = 3, 6, 2, (24, 16), 300000
    sh, p, rng = _flow(L, d, c, hidden, "tanh", 9, "lmm64")
    masks = _masks("blocks", L, d, rng)
    assert _hip.workspace_bytes(sh, _hip.OP_TRAIN, 10 ** 7) == _hip.workspace_bytes(sh, _hip.OP_TRAIN, 10 ** 6) < 2 ** 31
    P = p.size
    gen = torch.Generator(device="cuda").manual_seed(4)
    x = torch.randn(n, d, device="cuda", generator=gen); cc = torch.randn(n, c, device="cuda", generator=gen)
    perm = torch.randperm(n, device="cuda", generator=gen)
    pd, mk = _dev(p), _dev(masks, torch.uint8)
    ws = _ws(_hip, sh, _hip.OP_TRAIN, n)
    full = torch.empty(P + 1, device="cuda")
    _hip.loss_grad(sh, pd, mk, x, cc, perm, n, 1.0 / n, full[:P], full[P:], ws)
    disp = _hip.last_dispatch(_hip.PROFILE_TRAIN)
    assert disp["kernel"] == "k_lmm_train64" and disp["launches"] == 4 and disp["rows"] == n - 262144
    again = torch.empty(P + 1, device="cuda")
    _hip.loss_grad(sh, pd, mk, x, cc, perm, n, 1.0 / n, again[:P], again[P:], ws)
    assert torch.equal(full, again)
    acc = torch.zeros(P + 1, device="cuda", dtype=torch.float64)
    part = torch.empty(P + 1, device="cuda")
    for lo in range(0, n, 100000):
        _hip.loss_grad(sh, pd, mk, x, cc, perm[lo:lo + 100000].contiguous(), 100000, 1.0 / n, part[:P], part[P:], ws)
        acc += part.double()
    scale = acc[:P].abs().max().item()
    assert (full[:P].double() - acc[:P]).abs().max().item() < 3e-6 * scale
    assert abs(full[P].item() - acc[P].item()) < 1e-5 * abs(acc[P].item())
    direct = torch.empty(P + 1, device="cuda")
    _hip.loss_grad(sh, pd, mk, x[perm].contiguous(), cc[perm].contiguous(), None, n, 1.0 / n, direct[:P], direct[P:], ws)
    assert torch.equal(direct, full)


@pytest.mark.parametrize("n", [77, 9000])
def test_lmm64_backward_entry_point(n):
    """rnvp_backward (caller's d loss / d z and d loss / d logdet, d loss / d x out) on the 64-row form: seeded with the loss's own
    seeds it equals rnvp_loss_grad; gx equals the 16-row form's"""
    from probaforms_amd import _hip
    L, d, c, hidden = 4, 6, 2, (12, 20)
    sh, p, rng = _flow(L, d, c, hidden, "tanh", 21, "lmm64")
    sh16, _, _ = _flow(L, d, c, hidden, "tanh", 21, "lmm16")
    masks = _masks("alt", L, d, rng)
    X = rng.standard_normal((n, d)).astype(np.float32); C = rng.standard_normal((n, c)).astype(np.float32)
    pd, mk, xd, cd = _dev(p), _dev(masks, torch.uint8), _dev(X), _dev(C)
    P = p.size
    z = torch.empty(n, d, device="cuda")
    _hip.forward_logprob(sh, pd, mk, xd, cd, None, n, z, None, None, None, _ws(_hip, sh, _hip.OP_FORWARD, n))
    g1 = _loss_grad(_hip, sh, p, masks, X, C)
    gz = (z / n).contiguous(); gld = torch.full((n,), -1.0 / n, device="cuda")
    out = {}
    for name, s in (("64", sh), ("16", sh16)):
        g2 = torch.empty(P, device="cuda"); gx = torch.empty(n, d, device="cuda")
        _hip.backward(s, pd, mk, xd, cd, None, n, gz, gld, g2, gx, _ws(_hip, s, _hip.OP_TRAIN, n))
        out[name] = (g2, gx)
    scale = float(g1[:P].abs().max())
    assert float((g1[:P] - out["64"][0]).abs().max()) < 5e-6 * scale
    assert float((out["64"][0] - out["16"][0]).abs().max()) < 5e-6 * scale
    gxs = float(out["16"][1].abs().max())
    assert float((out["64"][1] - out["16"][1]).abs().max()) < 5e-6 * gxs


def test_lmm64_fit_through_the_class_api_matches_the_16_row_form():
    """RealNVP(hidden=(32, 32)).fit on 20000-row batches: the epoch loop (loss + gradient, Adam) on the 64-row form tracks the 16-row
    form's losses to rounding"""
    from probaforms_amd import _hip
    L, d, c, hidden, n = 4, 6, 2, (32, 32), 20000
    rng = np.random.default_rng(5)
    X = rng.standard_normal((n, d)).astype(np.float32); C = rng.standard_normal((n, c)).astype(np.float32)
    masks = _masks("alt", L, d, rng)
    losses = {}
    for fam in ("lmm64", "lmm16"):
        sh, p, _ = _flow(L, d, c, hidden, "tanh", 13, fam)
        P = p.size
        pp = _dev(p); m = torch.zeros(P, device="cuda"); v = torch.zeros(P, device="cuda")
        g = torch.empty(P, device="cuda"); loss = torch.empty(5, device="cuda")
        ws = _ws(_hip, sh, _hip.OP_TRAIN, n)
        xd, cd, mk = _dev(X), _dev(C), _dev(masks, torch.uint8)
        for step in range(1, 6):
            _hip.train_step(sh, pp, mk, xd, cd, None, n, 1.0 / n, g, loss[step - 1:step], m, v, 1e-2, 0.9, 0.999, 1e-8, 0.0, step, ws)
        losses[fam] = loss.cpu().numpy()
    assert losses["lmm64"][-1] < losses["lmm64"][0]
    assert np.abs(losses["lmm64"] - losses["lmm16"]).max() < 2e-5 * np.abs(losses["lmm16"]).max()


@pytest.mark.parametrize("L,d,c,hidden,n", [(3, 6, 2, (7, 9), 203), (2, 16, 4, (128, 128), 8193), (3, 80, 20, (24,), 64), (2, 5, 0, (33,), 1)])
def test_lmm64_stays_inside_the_workspace_it_asked_for(L, d, c, hidden, n):
    """the saved layer inputs, hidden activations, gradient seeds and per-workgroup partials all live in the caller's workspace:
    a poisoned band behind exactly rnvp_workspace_bytes stays untouched, and a workspace that is too small is refused"""
    from probaforms_amd import _hip
    sh, p, rng = _flow(L, d, c, hidden, "tanh", 31 + n, "lmm64")
    masks = _masks("alt", L, d, rng)
    X = rng.standard_normal((n, d)).astype(np.float32); C = rng.standard_normal((n, c)).astype(np.float32)
    nb = _hip.workspace_bytes(sh, _hip.OP_TRAIN, n)
    buf = torch.full((nb + 4096,), 0xA5, dtype=torch.uint8, device="cuda")
    P = p.size
    g = torch.empty(P + 1, device="cuda")
    pd, mk, xd, cd = _dev(p), _dev(masks, torch.uint8), _dev(X), (_dev(C) if c else None)
    _hip.loss_grad(sh, pd, mk, xd, cd, None, n, 1.0 / n, g[:P], g[P:], buf[:nb])
    assert _hip.last_dispatch(_hip.PROFILE_TRAIN)["kernel"] == "k_lmm_train64"
    torch.cuda.synchronize()
    assert bool((buf[nb:] == 0xA5).all()) and bool(torch.isfinite(g).all())
    with pytest.raises(Exception):
        _hip.loss_grad(sh, pd, mk, xd, cd, None, n, 1.0 / n, g[:P], g[P:], buf[:1024])


def test_lmm64_at_the_bench_size_equals_the_sum_of_its_quarters():
    """hidden=(128, 128) on 65 536 rows (bench.py's `hidden_128x128`: four blocks per workgroup, units accumulated across them) against
    four calls of 16 384 rows (one block per workgroup) and against the 16-row form on the same rows"""
    from probaforms_amd import _hip
    L, d, c, hidden, n = 8, 16, 4, (128, 128), 65536
    sh, p, rng = _flow(L, d, c, hidden, "tanh", 41, "auto", scale=0.12)
    sh16, _, _ = _flow(L, d, c, hidden, "tanh", 41, "lmm16")
    masks = _masks("alt", L, d, rng)
    gen = torch.Generator(device="cuda").manual_seed(8)
    x = torch.randn(n, d, device="cuda", generator=gen); cc = torch.randn(n, c, device="cuda", generator=gen)
    pd, mk = _dev(p), _dev(masks, torch.uint8)
    P = p.size
    ws = _ws(_hip, sh, _hip.OP_TRAIN, n)
    full = torch.empty(P + 1, device="cuda")
    _hip.loss_grad(sh, pd, mk, x, cc, None, n, 1.0 / n, full[:P], full[P:], ws)
    disp = _hip.last_dispatch(_hip.PROFILE_TRAIN)
    assert disp["kernel"] == "k_lmm_train64" and disp["grid"] == 256 and disp["rows"] == n
    acc = torch.zeros(P + 1, device="cuda", dtype=torch.float64)
    part = torch.empty(P + 1, device="cuda")
    for lo in range(0, n, 16384):
        _hip.loss_grad(sh, pd, mk, x[lo:lo + 16384].contiguous(), cc[lo:lo + 16384].contiguous(), None, 16384, 1.0 / n, part[:P], part[P:], ws)
        acc += part.double()
    scale = acc[:P].abs().max().item()
    assert (full[:P].double() - acc[:P]).abs().max().item() < 3e-6 * scale
    assert abs(full[P].item() - acc[P].item()) < 1e-5 * abs(acc[P].item())
    g16 = torch.empty(P + 1, device="cuda")
    _hip.loss_grad(sh16, pd, mk, x, cc, None, n, 1.0 / n, g16[:P], g16[P:], _ws(_hip, sh16, _hip.OP_TRAIN, n))
    assert _hip.last_dispatch(_hip.PROFILE_TRAIN)["kernel"] == "k_lmm_train"
    assert (full[:P] - g16[:P]).abs().max().item() < 3e-6 * scale
    assert abs(full[P].item() - g16[P].item()) < 1e-5 * abs(g16[P].item())
