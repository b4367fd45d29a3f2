#!/usr/bin/env python3
"""bench.py -- RealNVP fit+sample throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], "C2"): n = 1M rows per GPU of make_moons-shaped tabular
data, d=16, cond=4, 8 coupling layers, hidden=(128,), float32.  One STEP = one pass of the hot
path over one batch: a training step (loss + gradient + Adam, realnvp.py:246-251) on a batch of
65 536 shuffled rows per GPU, plus sampling (inverse pass, realnvp.py:279-282) of 65 536 rows per
GPU.  Inputs are resident in HBM before the timed region.  With N > 1 every rank processes its
own 65 536-row shard of a global batch of N x 65 536 (weak scaling) and the flat
[gradient | loss] buffer is all-reduced over RCCL before Adam.

`value` = rows (fit rows + sampled rows) per second over all GPUs.  The JSON line also carries
  roofline     -- the dominant kernel (fused loss+gradient) against the f32 MFMA peak, from HIP
                  events around each launch inside the timed region;
  cpu_baseline -- the CPU oracle (oracle/, a C port of the reference's algorithm) on all host cores
                  of rank 0's box, on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# workload (C2)
N_ROWS, D, CDIM, LAYERS, HIDDEN = 1_000_000, 16, 4, 8, (128,)
BATCH = 65_536
F32_MFMA_PEAK_TFLOPS = 157.3        # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"


def moons_block(n, rng, noise=0.1):
    """two interleaving half circles + gaussian noise, shuffled (the make_moons recipe), with labels"""
    n_out = n // 2
    n_in = n - n_out
    t_out = np.linspace(0, np.pi, n_out); t_in = np.linspace(0, np.pi, n_in)
    x = np.concatenate([np.stack([np.cos(t_out), np.sin(t_out)], 1),
                        np.stack([1 - np.cos(t_in), 1 - np.sin(t_in) - 0.5], 1)])
    y = np.concatenate([np.zeros(n_out), np.ones(n_in)])
    p = rng.permutation(n)
    x = x[p] + rng.normal(scale=noise, size=(n, 2)); y = y[p]
    return x, y


def make_data(n, d, c, seed):
    """d/2 independent moons blocks side by side, standardised; conditions = first c moon labels
    (padded with normals if c > d/2)  -- SURVEY.md 8(d) 'Synthetic inputs'."""
    rng = np.random.default_rng(seed)
    cols, labels = [], []
    for _ in range(d // 2):
        x, y = moons_block(n, rng)
        cols.append(x); labels.append(y)
    X = np.concatenate(cols, 1)
    X = (X - X.mean(0)) / X.std(0)
    lab = np.stack(labels, 1)
    C = lab[:, :c] if c <= lab.shape[1] else np.concatenate([lab, rng.normal(size=(n, c - lab.shape[1]))], 1)
    return X.astype(np.float32), C.astype(np.float32)


def useful_flops_per_row(d, c, hidden, L, passes):
    """F_useful = 4 h (d + c) per row per layer (dead masked lanes removed)"""
    return 4 * hidden[0] * (d + c) * L * passes


def pmc_traffic(kernel_prefix):
    """HBM bytes per launch of the dominant kernel from the committed PMC profile of this same
    command (scripts/gpu_traffic.sh: separate --pmc passes for FETCH_SIZE and WRITE_SIZE, FETCH_SIZE
    doubled per MI355X_MICROARCH.md).  bench.py cannot collect counters itself; None if absent."""
    path = os.path.join(ROOT, "profiles", "r01_traffic_pmc.json")
    try:
        d = json.load(open(path))
    except OSError:
        return None
    for k, v in d.items():
        if k.startswith(kernel_prefix):
            return float(v["hbm_bytes_per_launch"])
    return None


def cpu_baseline(X, C, params, rows_per_thread=16384):
    """The oracle (C port of the reference's algorithm) on the host cores: one training step
    (loss + gradient, shards summed, Adam) + sampling, on rows_per_thread rows per core -- the same
    mix as one GPU step.  Threads call into the C library concurrently (ctypes releases the GIL);
    each computes the gradient of its shard exactly like a data-parallel rank would."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import Oracle, Shape
    cores = max(1, min(os.cpu_count() or 1, 64))
    rows = rows_per_thread * cores
    rows = min(rows, X.shape[0])
    o = Oracle(32)
    s = Shape.make(LAYERS, D, CDIM, HIDDEN, "tanh")
    p = params.copy(); m = np.zeros_like(p); v = np.zeros_like(p)
    z = np.random.default_rng(1).normal(size=(rows, D)).astype(np.float32)
    chunks = [(i * rows // cores, (i + 1) * rows // cores) for i in range(cores)]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        grads = list(ex.map(lambda ab: o.loss_grad(s, p, X[ab[0]:ab[1]], C[ab[0]:ab[1]], inv_B=1.0 / rows)[1], chunks))
        g = np.sum(grads, axis=0, dtype=np.float32)
        o.adam(p, g, m, v, 1, lr=1e-3)
        list(ex.map(lambda ab: o.sample(s, p, z[ab[0]:ab[1]], C[ab[0]:ab[1]]), chunks))
    dt = time.perf_counter() - t0
    return dict(value=2 * rows / dt, unit="rows/s", cores=cores, kind="port",
                sample="oracle/rnvp_oracle.c (float32, gcc -O2) on %d threads: 1 training step on %d rows + sampling "
                       "%d rows of the C2 workload, %.1f s" % (cores, rows, rows, dt))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # BENCH_ONE_GPU=1 (developer aid): run the N-rank code path on a 1-GPU box -- every rank on cuda:0,
    # gloo instead of RCCL.  Never used by the driver; numbers from it mean nothing.
    one_gpu = os.environ.get("BENCH_ONE_GPU") == "1"
    if one_gpu:
        local = 0
    # BENCH_FORCE_DIST=1 (developer aid): a single rank still initialises RCCL and runs the data-parallel step
    # (unfused loss+grad, all-reduce on RCCL's stream under the sampling kernel, Adam) -- exercises the N > 1
    # code path, RCCL included, on a 1-GPU box.
    force_dist = os.environ.get("BENCH_FORCE_DIST") == "1" and world == 1
    if force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    dp = world > 1 or force_dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    from probaforms_amd import _engine, _hip
    from probaforms_amd.models import NormalizingFlow, RealNVPLayer, StandardNormalPrior

    # model: random init of the C2 architecture (same seed on every rank -> identical replicas)
    torch.manual_seed(0)
    layers = [RealNVPLayer(D, CDIM, (torch.arange(D) + i) % 2, HIDDEN, "tanh") for i in range(LAYERS)]
    nf = NormalizingFlow(layers, StandardNormalPrior(D, dev))
    for p in nf.parameters():
        p.data = p.data.to(dev)
    eng = nf.engine()
    if dp:
        _engine.broadcast_(eng.flat, 0)
    opt = _engine.FlatAdam(eng.flat.numel(), dev, lr=1e-3, weight_decay=0.0)

    # data: resident in HBM before timing; each rank owns its own n rows (weak scaling)
    Xh, Ch = make_data(N_ROWS, D, CDIM, seed=rank)
    X = torch.from_numpy(Xh).to(dev); C = torch.from_numpy(Ch).to(dev)
    torch.manual_seed(1 + rank)
    perm = _engine.loader_permutation(N_ROWS).to(dev)
    bounds = _engine.batch_bounds(N_ROWS, BATCH)
    gen = torch.Generator(device=dev).manual_seed(rank)
    z = torch.randn(BATCH, D, device=dev, generator=gen)             # prior draws for the sampling leg
    xs = torch.empty_like(z)
    losses = torch.zeros(args.steps + args.warmup, device=dev)
    inv_B = 1.0 / (BATCH * world)
    P = eng.P

    def step(i, timed_idx=None):
        s, e = bounds[i % (len(bounds) - 1)]                          # full batches only
        rows = perm[s:e]
        g = eng.loss_grad(X, C, rows, e - s, inv_B)
        c_rows = C[s:e]                                               # conditions of the sampled rows
        if dp:
            # the gradient all-reduce (RCCL, its own stream) runs under the sampling kernel, which
            # does not depend on it; Adam waits for the reduced gradient
            work = dist.all_reduce(g[:P + 1], op=dist.ReduceOp.SUM, async_op=True)
            eng.inverse(z, c_rows, out=xs)
            work.wait()
            losses[i:i + 1].copy_(g[P:P + 1])
            eng.adam(opt)
        else:
            losses[i:i + 1].copy_(g[P:P + 1])
            eng.adam(opt)
            eng.inverse(z, c_rows, out=xs)

    def step1(i):
        """single GPU: the fused rnvp_train_step (loss + gradient + Adam), as RealNVP.fit uses it"""
        s, e = bounds[i % (len(bounds) - 1)]
        eng.train_step(opt, X, C, perm[s:e], e - s, inv_B, losses[i:i + 1])
        eng.inverse(z, C[s:e], out=xs)

    if not dp:
        step = lambda i, timed_idx=None: step1(i)

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    # HIP events around the dominant kernel (the fused forward+backward launch), recorded by the
    # library on the stream it launches on, for every step of the timed region
    _hip.profile_enable(args.steps)
    if dp:
        dist.barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k, k)
    torch.cuda.synchronize()
    if dp:
        dist.barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if dp:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    n_timed, tot_ms = _hip.profile_read()
    _hip.profile_enable(0)
    assert n_timed == args.steps, (n_timed, args.steps)
    kern_ms = tot_ms / n_timed
    final_loss = float(losses[args.warmup + args.steps - 1].item())
    assert np.isfinite(final_loss), "training diverged"

    if rank == 0:
        rows_per_step = 2 * BATCH * world
        fl = useful_flops_per_row(D, CDIM, HIDDEN, LAYERS, 3) * BATCH      # fwd + dgrad + wgrad
        achieved = fl / (kern_ms * 1e-3) / 1e12
        path = _hip.kernel_path(eng.shape, eng.masks_host, _hip.OP_TRAIN)
        out = {
            "metric": "RealNVP samples/sec (fit+sample)", "value": rows_per_step * args.steps / dt,
            "unit": "rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C2: RealNVP n=1M/GPU d=16 cond=4 L=8 hidden=(128,), per-GPU batch 65536: "
                                   "1 train step (loss+grad+Adam) + 65536 sampled rows per step",
                       "global_batch": BATCH * world, "parallelism": "dp%d" % world,
                       "kernel_path": "mfma" if path == _hip.PATH_MFMA else "generic", "final_loss": final_loss},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / F32_MFMA_PEAK_TFLOPS,
                         "traffic": pmc_traffic("k_mfma_train" if path == _hip.PATH_MFMA else "k_generic_train"),
                         "kernel": "k_mfma_train / k_generic_train (fused forward+backward), %.3f ms avg over %d launches, "
                                   "%d useful flop/row x %d rows" % (kern_ms, args.steps,
                                                                     useful_flops_per_row(D, CDIM, HIDDEN, LAYERS, 3), BATCH)},
        }
        if not args.no_cpu_baseline:
            params = eng.params.detach().cpu().numpy()
            out["cpu_baseline"] = cpu_baseline(Xh, Ch, params)
            # second half of BASELINE.json's metric: per-row log-prob MAE of the HIP path against the
            # CPU restatement of the reference (float32 oracle, and its float64 referee) on the
            # trained weights, 4096 rows
            from oracle import Oracle, Shape
            rows = 4096
            lp = nf.log_prob_samples(X[:rows], C[:rows]).cpu().numpy()
            sh = Shape.make(LAYERS, D, CDIM, HIDDEN, "tanh")
            _, lp32, _ = Oracle(32).log_prob(sh, params, Xh[:rows], Ch[:rows])
            _, lp64, _ = Oracle(64).log_prob(sh, params, Xh[:rows], Ch[:rows])
            out["logprob_mae"] = {"vs_oracle_f32": float(np.abs(lp - lp32).mean()),
                                  "vs_oracle_f64": float(np.abs(lp - lp64).mean()),
                                  "oracle_f32_vs_f64": float(np.abs(lp32 - lp64).mean()), "rows": rows,
                                  "target": 1e-5}
        print(json.dumps(out), flush=True)
    if dp:
        dist.barrier()                      # rank 0 may still be timing the CPU baseline
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
