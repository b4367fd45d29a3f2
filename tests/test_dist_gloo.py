"""Data-parallel fit over torch.distributed (gloo, world_size 2) on CPU.

The product computes on HIP only, so here the C-ABI calls are replaced -- in the TEST process,
by monkeypatching probaforms_amd._hip -- with the CPU oracle as a compute stand-in.  What is
under test is the host logic of SURVEY.md 8(e): identical permutation on every rank, contiguous
shards of each global batch (ragged last batch, possibly empty shards), gradients scaled by
1/B_global, ONE all-reduce(SUM) of the flat [grad | loss] buffer per step, identical Adam on every
rank.  Two ranks must reproduce the single-process result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _install_oracle_backend():
    """swap the HIP entry points for oracle-backed CPU stand-ins (test process only)"""
    os.environ["device"] = "cpu"
    sys.path.insert(0, ROOT)
    from oracle import Oracle, Shape
    from probaforms_amd import _engine, _hip
    import probaforms_amd.models.nflow as nflow
    import probaforms_amd.models.realnvp as realnvp
    o = Oracle(32)

    def oshape(s):
        return Shape.make(s.L, s.d, s.c, tuple(s.hidden[:s.n_hidden]), "tanh" if s.act == 0 else "relu")

    def loss_grad(shape, params, masks, x, c, row_index, n_rows, inv_B, grad_out, loss_out, ws):
        if n_rows == 0:
            grad_out.zero_(); loss_out.zero_(); return
        rows = row_index.numpy() if row_index is not None else np.arange(n_rows)
        X = x.numpy()[rows]; Cc = None if c is None else c.numpy()[rows]
        loss, g = o.loss_grad(oshape(shape), params.numpy(), X, Cc, masks.numpy(), inv_B=inv_B)
        grad_out.copy_(torch.from_numpy(g)); loss_out[0] = float(loss)

    def adam_step(params, grad, m, v, n, lr, b1, b2, eps, wd, step):
        o.adam(params.numpy(), grad.numpy(), m.numpy(), v.numpy(), step, lr=lr, betas=(b1, b2), eps=eps,
               weight_decay=wd)

    def train_step(shape, params, masks, x, c, row_index, n_rows, inv_B, grad_buf, loss_out, m, v,
                   lr, b1, b2, eps, wd, step, ws):
        loss_grad(shape, params, masks, x, c, row_index, n_rows, inv_B, grad_buf, loss_out, ws)
        adam_step(params, grad_buf, m, v, params.numel(), lr, b1, b2, eps, wd, step)

    def inverse(shape, params, masks, z, c, n_rows, x_out, ws):
        x_out.copy_(torch.from_numpy(o.sample(oshape(shape), params.numpy(), z.numpy(),
                                              None if c is None else c.numpy(), masks.numpy())))

    def fit_epoch(shape, params, masks, x, c, perm, n, batch_size, grad_buf, loss_hist, m, v, lr, b1, b2, eps, wd,
                  first_step, ws):
        for k, s0 in enumerate(range(0, n, batch_size)):
            rows = min(batch_size, n - s0)
            train_step(shape, params, masks, x, c, perm[s0:s0 + rows], rows, 1.0 / rows, grad_buf, loss_hist[k:k + 1],
                       m, v, lr, b1, b2, eps, wd, first_step + k, ws)

    def dp_finish_step(params, grad_loss, m, v, n, lr, b1, b2, eps, wd, step, loss_out):
        loss_out[0] = float(grad_loss[n])
        adam_step(params, grad_loss[:n], m, v, n, lr, b1, b2, eps, wd, step)

    def fit_epoch_dp_cb(all_reduce, rank, world, shape, params, masks, x, c, perm, n, batch_size, grad_loss, loss_hist, m, v,
                        lr, b1, b2, eps, wd, first_step, ws, chunks=1):
        """stand-in for rnvp_fit_epoch_dp_cb (csrc/rnvp_dp.hip), the loop RealNVP.fit runs under a gloo process group: per
        global batch this rank's contiguous share (remainder rows to the low ranks), gradients scaled by 1 / B_global, ONE
        exchange of [gradient | loss], loss read-out + the identical Adam step.  (The C loop itself runs with two ranks in
        tests/test_dist_gpu.py.)"""
        P = params.numel()
        for k, s0 in enumerate(range(0, n, batch_size)):
            rows = min(batch_size, n - s0)
            base, rem = divmod(rows, world)
            lo = s0 + rank * base + min(rank, rem)
            mine = base + (1 if rank < rem else 0)
            loss_grad(shape, params, masks, x, c, perm[lo:lo + mine], mine, 1.0 / rows, grad_loss[:P], grad_loss[P:P + 1], ws)
            if chunks <= 1:
                all_reduce(grad_loss[:P + 1], P + 1)
            else:       # rnvp_fit_epoch_dp_cb_chunked: the message in chunks of layers, last layers first, the loss with the first
                L, per = shape.L, P // shape.L
                for j in range(min(chunks, L)):
                    l1, l0 = L - L * j // min(chunks, L), L - L * (j + 1) // min(chunks, L)
                    cnt = (l1 - l0) * per + (1 if j == 0 else 0)
                    all_reduce(grad_loss[l0 * per:l0 * per + cnt], cnt)
            dp_finish_step(params, grad_loss, m, v, P, lr, b1, b2, eps, wd, first_step + k, loss_hist[k:k + 1])

    def prior_normal(seed, row_offset, n_rows, d, z_out):
        z_out.copy_(torch.from_numpy(o.prior_normal(seed, row_offset, n_rows, d)))

    def sample(shape, params, masks, c, n_rows, seed, row_offset, x_out, ws):
        z = torch.empty(n_rows, shape.d)
        prior_normal(seed, row_offset, n_rows, shape.d, z)
        inverse(shape, params, masks, z, c, n_rows, x_out, ws)

    _hip.loss_grad, _hip.adam_step, _hip.train_step, _hip.inverse = loss_grad, adam_step, train_step, inverse
    _hip.fit_epoch, _hip.dp_finish_step, _hip.prior_normal, _hip.sample = fit_epoch, dp_finish_step, prior_normal, sample
    _hip.fit_epoch_dp_cb = fit_epoch_dp_cb
    _hip.workspace_bytes = lambda shape, op, rows: 16
    for mod in (_engine, nflow, realnvp):
        mod.require_hip = lambda device: None
    return realnvp


def _data():
    rng = np.random.default_rng(3)
    return rng.normal(size=(75, 4)), rng.normal(size=(75, 2))        # 75 rows, bs 32 -> 32, 32, 11


def _fit(world_rank=None):
    realnvp = _install_oracle_backend()
    X, C = _data()
    # ranks other than 0 arrive with a different generator state: parameters and the epoch shuffles must
    # still be rank 0's (both are broadcast), so the run equals the single-process run seeded with 0
    torch.manual_seed(0 if not world_rank else 100 + world_rank)
    m = realnvp.RealNVP(n_layers=3, hidden=(6,), batch_size=32, n_epochs=2, lr=0.01, weight_decay=0.1)
    m.fit(X, C)
    flat = torch.cat([p.detach().reshape(-1) for p in m.nf.parameters()]).numpy().copy()
    hist = np.array([float(v) for v in m.loss_history])
    # sampling: shares per rank / gathered / replicated, all from the same generator state
    Cs = C[:37]
    out = {}
    for mode in (None, "shard", "gather"):
        torch.manual_seed(9)
        out[str(mode)] = m.sample(Cs, distributed=mode) if mode else m.sample(Cs)
    # counter-based 'device' prior: ranks arrive with DIFFERENT generator states (rank 0's seed is broadcast) and
    # each draws only its own global rows; the shares must still tile the single-process draw seeded like rank 0
    m.prior.host_rng = False
    for mode in (None, "shard", "gather"):
        torch.manual_seed(9 if not world_rank else 77 + world_rank)
        out["dev_" + str(mode)] = m.sample(Cs, distributed=mode) if mode else m.sample(Cs)
    return flat, hist, out


def _worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        flat, hist, smp = _fit(rank)
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), flat=flat, hist=hist, **{"s_" + k: v for k, v in smp.items()})
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_match_single_process(tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / "rank0.npz"); r1 = np.load(tmp_path / "rank1.npz")
    # every rank applies the identical update: replicas stay bit-identical
    assert np.array_equal(r0["flat"], r1["flat"]) and np.array_equal(r0["hist"], r1["hist"])
    assert r0["hist"].shape == (6,)                                   # 3 batches x 2 epochs
    # reference run in a fresh single process (world size 1 path: fused train_step)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_single, args=(q,)); p.start(); flat1, hist1, smp1 = q.get(timeout=240); p.join()
    # only the summation order of the two shards differs from the single-process gradient
    np.testing.assert_allclose(r0["hist"], hist1, rtol=2e-5, atol=2e-5)
    assert np.abs(r0["flat"] - flat1).max() < 5e-4 and np.abs(r0["flat"] - flat1).mean() < 2e-5
    # sampling: the two shares are the two halves of what every rank draws alone; 'gather' returns all of it
    for r in (r0, r1):
        assert r["s_None"].shape == (37, 4) and np.array_equal(r["s_gather"], r["s_None"])
    assert r0["s_shard"].shape == (19, 4) and r1["s_shard"].shape == (18, 4)
    assert np.array_equal(np.concatenate([r0["s_shard"], r1["s_shard"]]), r0["s_None"])
    assert smp1["shard"].shape == (37, 4)                            # world size 1: the keyword changes nothing
    # device prior: shares of the two ranks (seeded differently) tile the single-process draw, no duplicated rows
    dev_full = r0["s_dev_None"]                                      # rank 0 drawing all rows alone (same weights)
    shares = np.concatenate([r0["s_dev_shard"], r1["s_dev_shard"]])
    assert np.array_equal(shares, dev_full) and np.array_equal(r1["s_dev_gather"], dev_full)
    np.testing.assert_allclose(dev_full, smp1["dev_None"], atol=5e-3)   # single process: same draw, weights differ by summation order
    assert len({row.tobytes() for row in shares}) == 37
    assert np.abs(dev_full - r0["s_None"]).max() > 0.1               # and it is a different stream from the host prior


def _single(q):
    q.put(_fit(None))


def test_three_way_shards_cover_ragged_batch():
    from probaforms_amd._engine import batch_bounds, shard_bounds
    n, bs, world = 75, 32, 3
    seen = []
    for s, e in batch_bounds(n, bs):
        for r in range(world):
            lo, hi = shard_bounds(s, e, r, world)
            seen.extend(range(lo, hi))
    assert seen == list(range(n))
