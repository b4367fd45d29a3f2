"""Worker of tests/test_dist_gpu.py: one rank of a data-parallel RealNVP.fit on the HIP kernels.
Launched by torch.distributed.run; on a 1-GPU box every rank uses cuda:0 (env `device`) and the
collective runs over gloo -- the code path is the one RCCL serves on a multi-GPU node."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _chunked_epoch_matches(rank, world):
    from probaforms_amd import _engine
    from probaforms_amd.models import NormalizingFlow, RealNVPLayer, StandardNormalPrior
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(3)
    n, d, c = 20000, 16, 4
    X = torch.randn(n, d, generator=g).to(dev); C = torch.randn(n, c, generator=g).to(dev)
    perm = torch.randperm(n, generator=g).to(dev)
    results = []
    for chunks in (1, 3, 8):
        torch.manual_seed(0)
        layers = [RealNVPLayer(d, c, (torch.arange(d) + i) % 2, (128,), "tanh") for i in range(8)]
        nf = NormalizingFlow(layers, StandardNormalPrior(d, dev))
        for p in nf.parameters():
            p.data = p.data.to(dev)
        eng = nf.engine()
        opt = _engine.FlatAdam(eng.flat.numel(), dev, lr=1e-3, weight_decay=0.01)
        losses = torch.zeros(3, device=dev)
        eng.fit_epoch_dp(opt, None, X, C, perm, 8192, losses, exchange=lambda t, count: _engine.all_reduce_sum(t), rank=rank,
                         world=world, chunks=chunks)
        torch.cuda.synchronize()
        results.append(torch.cat([eng.flat.detach().reshape(-1), opt.exp_avg.reshape(-1), opt.exp_avg_sq.reshape(-1), losses]).clone())
    ok = all(torch.equal(results[0], r) for r in results[1:]) and bool(torch.isfinite(results[0]).all())
    both = [torch.empty_like(results[0]) for _ in range(world)]
    dist.all_gather(both, results[1])
    return bool(ok and all(torch.equal(both[0], b) for b in both))


def main():
    out = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    from probaforms_amd.models import RealNVP
    rng = np.random.default_rng(0)
    X = rng.standard_normal((1000, 5)); C = rng.standard_normal((1000, 3))
    torch.manual_seed(0 if rank == 0 else 1234 + rank)          # rank 0's init and shuffles must win
    m = RealNVP(n_layers=4, hidden=(16,), batch_size=96, n_epochs=2, lr=1e-2, weight_decay=0.05)
    m.fit(X, C)                                                  # 1000 rows, bs 96 -> last batch 40 rows
    flat = m.nf.engine().flat.detach().clone()
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    same = all(torch.equal(gathered[0], g) for g in gathered)
    torch.manual_seed(5)
    xs = m.sample(C[:64])
    smp = {}
    for mode in ("shard", "gather"):                             # row shares per rank, optional gather
        torch.manual_seed(5)
        smp[mode] = m.sample(C[:61], distributed=mode)
    torch.manual_seed(5)
    full61 = m.sample(C[:61])
    # counter-based device prior with the SAME seed on both ranks (the usual data-parallel convention): the shares
    # must tile the single-process draw and contain no repeated rows
    m.prior.host_rng = False
    dev = {}
    for mode in ("shard", "gather", "full"):
        torch.manual_seed(5)
        dev[mode] = m.sample(C[:61], distributed=mode) if mode != "full" else m.sample(C[:61])
    np.savez(out + ".rank%d.npz" % rank, shard=smp["shard"], gather=smp["gather"], full=full61,
             dev_shard=dev["shard"], dev_gather=dev["gather"], dev_full=dev["full"])
    # CVAE, data parallel: ranks seeded differently must still walk rank 0's shuffle and noise
    from probaforms_amd.models import CVAE
    torch.manual_seed(0 if rank == 0 else 999 + rank)
    cv = CVAE(latent_dim=2, hidden=(16,), batch_size=96, n_epochs=2, lr=1e-2)
    cv.fit(X, C)
    cflat = cv._core.flat.detach().clone()
    cg = [torch.empty_like(cflat) for _ in range(world)]
    dist.all_gather(cg, cflat)
    csame = all(torch.equal(cg[0], g) for g in cg)
    # the reference's C2 fit at its default batch size (tests/golden/c2_fit.npz), two ranks: 16 rows per rank per step
    f = np.load(os.path.join(ROOT, "tests", "golden", "c2_fit.npz"))
    torch.manual_seed(0 if rank == 0 else 77 + rank)
    m2 = RealNVP(n_layers=8, hidden=(128,), lr=0.001, n_epochs=2)
    m2.fit(f["X"], f["C"])
    c2_hist = np.array([float(v) for v in m2.loss_history])
    c2_flat = m2.nf.engine().flat.detach().cpu().numpy()
    if rank == 0:
        np.savez(out + ".c2fit.npz", hist=c2_hist, flat=c2_flat)
    # round 6: the step's exchange in chunks of layers (rnvp_fit_epoch_dp_cb_chunked) against the one-message loop, two ranks,
    # C2 flow, global batch 8192 (4096 rows per rank; ragged last batch): identical bits on both ranks
    chunk_same = _chunked_epoch_matches(rank, world)
    if rank == 0:
        np.savez(out + ".chunks.npz", same=chunk_same)
    if rank == 0:
        np.savez(out, flat=flat.cpu().numpy(), hist=np.array([float(v) for v in m.loss_history]), same=same, xs=xs,
                 cvae_flat=cflat.cpu().numpy(), cvae_hist=np.array([float(v) for v in cv.loss_history]), cvae_same=csame)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
