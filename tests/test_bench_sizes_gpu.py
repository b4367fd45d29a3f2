"""Every kernel variant that serves a BASELINE.json configuration AT ITS BENCHMARK BATCH, against the float64 oracle.

The fixture-sized tests (test_hip_kernels.py) reach the row-parallel, net-split, wide and tile-split training kernels only
through whatever a 32..9000-row call dispatches to.  Here each call is sized like bench.py's (65 536-row batches, the
ragged 16 960-row tail, C3 / C4 per-GPU shards), `rnvp_last_dispatch` pins WHICH kernel served it -- so a later change of a
dispatch threshold cannot silently move a benchmark size onto an untested kernel -- and loss + full gradient are compared
with the float64 oracle (oracle/rnvp_oracle.c; math: /root/reference/probaforms/models/realnvp.py:91-101,246-250) run on
the host cores in row blocks.  Tolerances are those of test_hip_kernels.py::test_loss_grad.
"""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev(a, dtype=torch.float32):
    return None if a is None else torch.as_tensor(np.ascontiguousarray(a)).to(dtype).cuda().contiguous()


def _threads():
    try:
        from probaforms_amd._engine import effective_cpus
        return max(1, min(effective_cpus(), 32))
    except Exception:  # pragma: no cover
        return max(1, (os.cpu_count() or 2) // 2)


def oracle64_loss_grad(s, params, X, C, inv_B, block=1024):
    """float64 loss and gradient of a big batch: row blocks on a thread pool (ctypes releases the GIL), block results
    added in float64 in block order"""
    from oracle import Oracle
    o = Oracle(64)
    p64 = params.astype(np.float64)
    n = X.shape[0]
    cuts = [(a, min(a + block, n)) for a in range(0, n, block)]

    def one(ab):
        a, b = ab
        lo, g = o.loss_grad(s, p64, X[a:b].astype(np.float64), None if C is None else C[a:b].astype(np.float64), inv_B=inv_B)
        return float(lo), np.asarray(g, np.float64)
    with ThreadPoolExecutor(_threads()) as ex:
        parts = list(ex.map(one, cuts))
    return sum(p[0] for p in parts), np.sum([p[1] for p in parts], axis=0)


def oracle64_log_prob(s, params, X, C, block=4096):
    from oracle import Oracle
    o = Oracle(64)
    p64 = params.astype(np.float64)
    n = X.shape[0]
    cuts = [(a, min(a + block, n)) for a in range(0, n, block)]

    def one(ab):
        a, b = ab
        z, lp, _ = o.log_prob(s, p64, X[a:b].astype(np.float64), None if C is None else C[a:b].astype(np.float64))
        return np.asarray(z), np.asarray(lp)
    with ThreadPoolExecutor(_threads()) as ex:
        parts = list(ex.map(one, cuts))
    return np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])


C2 = (8, 16, 4, 128)
C3 = (12, 32, 8, 256)
C4 = (8, 64, 16, 128)

# (id, (L, d, c, h), rows, precision, expected kernel, variant, GEMM1 arithmetic of the forward phase)
TRAIN_CASES = [
    # C2 (configs[1]): the bench batch runs the net-split kernel with the split-bf16 forward `auto` ships, the epoch's ragged
    # tail likewise; 70 000 rows leave the one-workgroup-per-CU regime: row-parallel launch, compact dW2 records
    ("c2-65536-auto", C2, 65536, "auto", "k_mfma_train", "netsplit", "bx3"),
    ("c2-16960-auto", C2, 16960, "auto", "k_mfma_train", "netsplit", "bx3"),
    ("c2-70000-auto", C2, 70000, "auto", "k_mfma_train", "rowpar", "bx3"),
    ("c2-65536-f32", C2, 65536, "f32", "k_mfma_train", "netsplit", "f32"),
    ("c2-8192-auto", C2, 8192, "auto", "k_mfma_train_ts", "tilesplit", "f32"),
    # where `auto` switches for d <= 16 (round 6: above 64 hidden units -- five hidden tiles on)
    ("c2h64-65536-auto", (8, 16, 4, 64), 65536, "auto", "k_mfma_train", "netsplit", "f32"),
    ("c2h80-65536-auto", (8, 16, 4, 80), 65536, "auto", "k_mfma_train", "netsplit", "bx3"),
    ("c2h96-65536-auto", (8, 16, 4, 96), 65536, "auto", "k_mfma_train", "netsplit", "bx3"),
    # C3 (configs[2]): 65 536 rows per rank and the ragged tail of a 1M-row epoch take the wide kernel; 30 000 rows the net-split one
    ("c3-65536-auto", C3, 65536, "auto", "k_mfma_train_wide", "wide", "bx3"),
    ("c3-40000-auto", C3, 40000, "auto", "k_mfma_train_wide", "wide", "bx3"),
    ("c3-40000-f32", C3, 40000, "f32", "k_mfma_train_wide", "wide", "f32"),
    ("c3-16960-auto", C3, 16960, "auto", "k_mfma_train", "netsplit", "bx3"),
    # the strong-batch regime SURVEY 8(d)/(e) names: global batch 65 536 over 8 ranks = 8 192 rows per rank (and the 2 120-row
    # share of the ragged last batch): the tile-split kernel, two row tiles per workgroup from 4 097 rows on (round 5)
    ("c3-8192-auto", C3, 8192, "auto", "k_mfma_train_ts", "tilesplit", "f32"),
    ("c3-8192-f32", C3, 8192, "f32", "k_mfma_train_ts", "tilesplit", "f32"),
    ("c3-5000-auto", C3, 5000, "auto", "k_mfma_train_ts", "tilesplit", "f32"),
    ("c3-2120-auto", C3, 2120, "auto", "k_mfma_train_ts", "tilesplit", "f32"),
    ("c2-2120-auto", C2, 2120, "auto", "k_mfma_train_ts", "tilesplit", "f32"),
    # C4's geometry (d = 64: NF = 8, one row tile per wave)
    ("c4-20000-auto", C4, 20000, "auto", "k_mfma_train", "rowpar", "f32"),
    ("c4-65536-auto", C4, 65536, "auto", "k_mfma_train", "rowpar", "f32"),
]


def _model(shape_t, seed):
    from oracle import Shape
    L, d, c, h = shape_t
    rng = np.random.default_rng(seed)
    s = Shape.make(L, d, c, (h,), "tanh")
    P = 2 * L * (h * (d + c) + h + d * h + d)
    params = (rng.uniform(-1, 1, size=P) * min(0.5, 1.5 / np.sqrt(h + d + c))).astype(np.float32)
    return s, params, rng


@pytest.mark.parametrize("cid,shape_t,n,prec,kernel,variant,gemm1", TRAIN_CASES, ids=[t[0] for t in TRAIN_CASES])
def test_training_kernel_at_benchmark_batch_vs_float64_oracle(cid, shape_t, n, prec, kernel, variant, gemm1):
    from probaforms_amd import _hip
    L, d, c, h = shape_t
    s, params, rng = _model(shape_t, 1000 + n + d)
    # rows are gathered through a permutation of a larger table, as RealNVP.fit's batches are (row_index)
    N = n + 4097
    Xall = rng.standard_normal((N, d)).astype(np.float32); Call = rng.standard_normal((N, c)).astype(np.float32)
    idx = rng.permutation(N)[:n].astype(np.int64)
    shape = _hip.RnvpShape.make(L, d, c, (h,), "tanh", alt_masks=1, precision=prec)
    P = _hip.param_count(shape)
    assert P == params.size
    ws = torch.empty(_hip.workspace_bytes(shape, _hip.OP_TRAIN, n), dtype=torch.uint8, device="cuda")
    pd, xd, cd, id_ = _dev(params), _dev(Xall), _dev(Call), _dev(idx, torch.int64)
    outs = []
    for _ in range(2):
        grad = torch.full((P,), float("nan"), device="cuda"); loss = torch.full((1,), float("nan"), device="cuda")
        _hip.loss_grad(shape, pd, None, xd, cd, id_, n, 1.0 / n, grad, loss, ws)
        torch.cuda.synchronize()
        outs.append((grad.clone(), loss.clone()))
    got = _hip.last_dispatch(_hip.PROFILE_TRAIN)
    assert (got["kernel"], got["variant"], got["gemm1_fwd"], got["rows"]) == (kernel, variant, gemm1, n), got
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])      # no float atomics: same bits
    lo, go = oracle64_loss_grad(s, params, Xall[idx], Call[idx], 1.0 / n)
    g = outs[0][0].cpu().numpy().astype(np.float64); l = float(outs[0][1].item())
    gerr = np.abs(g - go).max() / np.abs(go).max()
    lerr = abs(l - lo)
    print("%s: %s grid %d R %d | loss %.6f (oracle %.6f, err %.2e) | grad err %.2e of scale" %
          (cid, got["kernel"], got["grid"], got["row_tiles"], l, lo, lerr, gerr))
    assert lerr < max(1e-5, 5e-7 * abs(lo)), (l, lo)
    assert gerr < 3e-6, gerr


FLOW_CASES = [
    # (id, shape, rows, precision, kernel, variant): the flow kernels at the sizes bench.py times them on
    ("c2-1M-auto", C2, 1_000_000, "auto"),          # round 6: auto takes the barrier-free split-bf16 form for hidden > 64 (5-9 % faster with warm clocks)
    ("c2-1M-bx3", C2, 1_000_000, "bx3"),
    ("c2-1M-f32", C2, 1_000_000, "f32"),            # the f32 register-chained kernels stay available on request (and serve hidden <= 64)
    ("c2h64-1M-auto", (8, 16, 4, 64), 1_000_000, "auto"),
    ("c2h96-1M-auto", (8, 16, 4, 96), 1_000_000, "auto"),
    ("c2h80-300k-auto", (8, 16, 4, 80), 300_000, "auto"),
    ("c3-200k-auto", C3, 200_003, "auto"),
    ("c4-200k-auto", C4, 200_003, "auto"),
]


@pytest.mark.parametrize("cid,shape_t,n,prec", FLOW_CASES, ids=[t[0] for t in FLOW_CASES])
def test_flow_kernels_at_benchmark_size_vs_float64_oracle(cid, shape_t, n, prec):
    """log-prob and inverse at bench sizes: a strided sample of the rows against the float64 oracle (every 97th row of the
    call -- the kernels cannot know which rows are checked), the round trip on all rows, and the dispatched kernel"""
    from probaforms_amd import _hip
    L, d, c, h = shape_t
    s, params, rng = _model(shape_t, 77 + d)
    gen = torch.Generator(device="cuda").manual_seed(n + d)
    x = torch.randn(n, d, device="cuda", generator=gen); cc = torch.randn(n, c, device="cuda", generator=gen)
    shape = _hip.RnvpShape.make(L, d, c, (h,), "tanh", alt_masks=1, precision=prec)
    pd = _dev(params)
    z = torch.empty(n, d, device="cuda"); lp = torch.empty(n, device="cuda"); tot = torch.empty(1, device="cuda")
    wsf = torch.empty(_hip.workspace_bytes(shape, _hip.OP_FORWARD, n), dtype=torch.uint8, device="cuda")
    _hip.forward_logprob(shape, pd, None, x, cc, None, n, z, None, lp, tot, wsf)
    torch.cuda.synchronize()
    fwd = _hip.last_dispatch(_hip.PROFILE_FORWARD)
    bx3 = prec == "bx3" or (prec == "auto" and (d > 16 or h > 64))
    assert fwd["kernel"] == ("k_flow_bx3" if bx3 else "k_mfma_flow") and fwd["rows"] == n, fwd
    assert fwd["variant"] == (("bx3_direct" if d <= 16 else "bx3_staged") if bx3 else "rowpar"), fwd
    sel = np.arange(0, n, 97)
    Xs = x[sel].cpu().numpy(); Cs = cc[sel].cpu().numpy()
    zo, lpo = oracle64_log_prob(s, params, Xs, Cs)
    zerr = np.abs(z[sel].cpu().numpy() - zo)
    lperr = np.abs(lp[sel].cpu().numpy() - lpo)
    print("%s: %s | z MAE %.2e max %.2e | logp MAE %.2e (|logp| max %.0f)" % (cid, fwd["kernel"], zerr.mean(), zerr.max(), lperr.mean(), np.abs(lpo).max()))
    assert zerr.mean() < 2e-6 and zerr.max() < 2e-4
    assert lperr.mean() < max(1e-5, 2.4e-7 * np.abs(lpo).max())
    assert abs(float(tot) - float(lp.double().sum())) < 2e-5 * float(lp.double().abs().sum())
    back = torch.empty_like(z)
    wsi = torch.empty(_hip.workspace_bytes(shape, _hip.OP_INVERSE, n), dtype=torch.uint8, device="cuda")
    _hip.inverse(shape, pd, None, z, cc, n, back, wsi)
    torch.cuda.synchronize()
    inv = _hip.last_dispatch(_hip.PROFILE_INVERSE)
    assert inv["kernel"] == fwd["kernel"] and inv["variant"] == fwd["variant"] and inv["rows"] == n, inv
    err = (back - x).abs()
    assert err.mean().item() < 2e-6 and err.max().item() < 5e-4


def test_fit_epoch_reports_what_bench_labels():
    """bench.py names its kernels from rnvp_last_dispatch: one C2 epoch slice through rnvp_fit_epoch leaves the record of its
    LAST batch (here the ragged one), and the launch count per step is what DESIGN.md states"""
    from probaforms_amd import _hip
    L, d, c, h = C2
    s, params, rng = _model(C2, 5)
    n, batch = 65536 + 16960, 65536
    shape = _hip.RnvpShape.make(L, d, c, (h,), "tanh", alt_masks=1)
    P = _hip.param_count(shape)
    x = _dev(rng.standard_normal((n, d)).astype(np.float32)); cc = _dev(rng.standard_normal((n, c)).astype(np.float32))
    perm = torch.randperm(n, device="cuda")
    p = _dev(params); m = torch.zeros(P, device="cuda"); v = torch.zeros(P, device="cuda"); g = torch.empty(P, device="cuda")
    hist = torch.zeros(2, device="cuda")
    ws = torch.empty(_hip.workspace_bytes(shape, _hip.OP_TRAIN, batch), dtype=torch.uint8, device="cuda")
    _hip.fit_epoch(shape, p, None, x, cc, perm, n, batch, g, hist, m, v, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, ws)
    torch.cuda.synchronize()
    got = _hip.last_dispatch(_hip.PROFILE_TRAIN)
    assert got["kernel"] == "k_mfma_train" and got["variant"] == "netsplit" and got["rows"] == 16960, got
    assert got["gemm1_fwd"] == "bx3"
    assert got["launches"] == 2                  # training kernel + finish (sum, Adam, re-pack): the pack launch ran once, before the first batch
    assert torch.isfinite(hist).all()


REPACK_CASES = [
    # (L, d, c, h, n, batch): every register-chained training variant, several steps in ONE rnvp_fit_epoch call
    (8, 16, 4, 128, 3 * 65536 + 777, 65536),      # C2: net-split kernel, ragged tail
    (8, 16, 4, 128, 2 * 70000, 70000),            # row-parallel (compact dW2)
    (8, 16, 4, 128, 5 * 32, 32),                  # tile split at the reference's default batch
    (3, 16, 0, 40, 3 * 5000, 5000),               # no condition (NF 2, CQ 0), padded hidden tiles
    (12, 32, 8, 256, 2 * 40000 + 100, 40000),     # C3: wide kernel
    (4, 32, 8, 64, 3 * 9000, 9000),               # NF 4 net-split / row-parallel
    (8, 64, 16, 128, 2 * 20000, 20000),           # C4's geometry (NF 8)
    (3, 5, 3, 40, 4 * 64, 64),                    # the reference's own test widths (d = 5, cdim = 3): padded d, cdim, hidden (40 units: not LDS-resident)
]


@pytest.mark.parametrize("L,d,c,h,n,batch", REPACK_CASES)
def test_fit_epoch_repack_equals_a_loop_of_train_steps_bit_for_bit(L, d, c, h, n, batch):
    """rnvp_fit_epoch packs the weight fragments ONCE and lets every step's finish kernel re-pack what Adam updated;
    rnvp_train_step packs at the head of every call.  Same kernels otherwise: parameters, both Adam moments and the batch losses
    must agree bit for bit after several steps -- a stale or misplaced fragment would change the second step's gradient."""
    from probaforms_amd import _hip
    shape = _hip.RnvpShape.make(L, d, c, (h,), "tanh", alt_masks=1)
    P = _hip.param_count(shape)
    gen = torch.Generator(device="cuda").manual_seed(L * 1000 + d + n)
    p0 = (torch.rand(P, device="cuda", generator=gen) - 0.5) * min(0.6, 3.0 / np.sqrt(h + d + c))
    x = torch.randn(n, d, device="cuda", generator=gen)
    cc = torch.randn(n, c, device="cuda", generator=gen) if c else None
    perm = torch.randperm(n, device="cuda", generator=gen)
    ws = torch.empty(_hip.workspace_bytes(shape, _hip.OP_TRAIN, batch), dtype=torch.uint8, device="cuda")
    nb = (n + batch - 1) // batch
    # (a) one library call for the epoch
    pa = p0.clone(); ma = torch.zeros(P, device="cuda"); va = torch.zeros(P, device="cuda"); ga = torch.empty(P, device="cuda")
    la = torch.zeros(nb, device="cuda")
    _hip.fit_epoch(shape, pa, None, x, cc, perm, n, batch, ga, la, ma, va, 1e-3, 0.9, 0.999, 1e-8, 0.01, 1, ws)
    assert _hip.last_dispatch(_hip.PROFILE_TRAIN)["launches"] == (2 if nb > 1 else 3)
    # (b) the same batches, one rnvp_train_step call each
    pb = p0.clone(); mb = torch.zeros(P, device="cuda"); vb = torch.zeros(P, device="cuda"); gb = torch.empty(P, device="cuda")
    lb = torch.zeros(nb, device="cuda")
    for k in range(nb):
        lo = k * batch; rows = min(batch, n - lo)
        _hip.train_step(shape, pb, None, x, cc, perm[lo:lo + rows].contiguous(), rows, 1.0 / rows, gb, lb[k:k + 1], mb, vb,
                        1e-3, 0.9, 0.999, 1e-8, 0.01, k + 1, ws)
    torch.cuda.synchronize()
    assert torch.isfinite(la).all() and torch.isfinite(pa).all()
    assert torch.equal(la, lb) and torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb)
    assert not torch.equal(pa, p0)
