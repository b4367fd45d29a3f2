#!/usr/bin/env python3
"""bench.py -- RealNVP fit+sample throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], "C2"): n = 1M rows per GPU of make_moons-shaped tabular data, d=16,
cond=4, 8 coupling layers, hidden=(128,).  One STEP = one pass of the hot path over the rank's data set:
  fit    -- one epoch of RealNVP.fit's batch loop (realnvp.py:235-254): 16 batches of the epoch's shuffled
            rows (15 x 65 536 + the ragged 16 960), each loss + gradient + Adam;
  sample -- RealNVP.sample for the same 1M conditions (realnvp.py:279-282; nflow.py:141-143): the prior
            draw (counter-based device prior, made inside the inverse kernel) + the inverse pass.
X, C and the epoch permutations are resident in HBM before the timed region.  With N > 1 every rank owns
its own 1M rows (weak scaling): per batch each rank computes the gradient of its 65 536-row shard of the
global batch of N x 65 536, the flat [gradient | loss] buffer is all-reduced over RCCL, every rank applies
the same Adam step; sampling shards the global N x 1M rows with no collective.

With --gpus N > 1 and no WORLD_SIZE in the environment this script starts its own N ranks (child processes
of `python -m torch.distributed.run`; nothing in the parent touches the GPU) and relays their output.

`value` = rows (fit rows + sampled rows) per second over all GPUs.  The JSON line also carries
  roofline        -- the dominant kernel (fused forward+backward) against the f32 MFMA peak, from HIP events
                     around every one of its launches inside the timed region;
  roofline_kernels-- the same for the sampling kernel (timed region) and the log-prob kernel (measured after it);
  cpu_baseline    -- the reference's CPU path (oracle/torch_cpu.py: eager PyTorch CPU ops in the reference's order,
                     validated against the reference in the build container) on the host cores of rank 0's box, on a
                     bounded sample of the same workload; cpu_baseline_c_oracle: the scalar C oracle on all cores;
  api_level       -- numpy in -> numpy out rates of RealNVP.fit / .sample on the same data (N = 1 only);
  logprob_mae     -- second half of the metric: per-row log-prob of the HIP path against the oracle.
"""
import argparse
import glob
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# workload (C2)
N_ROWS, D, CDIM, LAYERS, HIDDEN = 1_000_000, 16, 4, 8, (128,)
BATCH = 65_536
F32_MFMA_PEAK_TFLOPS = 157.3        # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r02_traffic_pmc.json")


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


def csrc_hash():
    """identifies the kernel sources a PMC profile was taken with (no .git on the GPU box)"""
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "probaforms_amd", "csrc", "*.hip")) +
                   glob.glob(os.path.join(ROOT, "probaforms_amd", "csrc", "*.h")) +
                   glob.glob(os.path.join(ROOT, "include", "*.h")))
    for f in files:
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel_prefix):
    """HBM bytes per launch of a kernel from the committed PMC profile of this same command
    (scripts/gpu_traffic.sh: separate --pmc passes for FETCH_SIZE and WRITE_SIZE, FETCH_SIZE doubled per
    MI355X_MICROARCH.md).  bench.py cannot collect counters itself; the file's number is reported only while
    it was taken with exactly these kernel sources (csrc_hash) and names this kernel, else None."""
    try:
        d = json.load(open(TRAFFIC_FILE))
    except (OSError, ValueError):
        return None
    if d.get("csrc_hash") != csrc_hash():
        return None
    for k, v in d.get("kernels", {}).items():
        if k.startswith(kernel_prefix):
            return float(v["hbm_bytes_per_launch"])
    return None


def cpu_baseline(X, C, rows=4 * BATCH):
    """The reference's CPU path on THIS box's host cores: oracle/torch_cpu.py, an eager-PyTorch
    loop that issues the reference's op sequence -- DataLoader(TensorDataset, shuffle=True) batches of 65 536 rows, per layer
    cat -> Linear -> Tanh -> Linear for both nets, exp, masked affine (int64 masks), MultivariateNormal prior, autograd
    backward, torch.optim.Adam, per-step loss read-back; sampling through the reversed layers -- validated in the build
    container against the reference itself (tests/golden/validate_torch_cpu.py: identical outputs, fit 0.94x / sample
    1.10x / combined 1.00x the reference's rows/s on the same 8 cores).  Bounded sample: one epoch over `rows` rows of the
    C2 data + sampling `rows` rows (the 1:1 mix of a GPU step), after a one-batch warm-up; two thread counts, best kept."""
    from oracle.torch_cpu import timed_fit_and_sample
    from probaforms_amd._engine import effective_cpus
    ncpu, grant = os.cpu_count() or 2, effective_cpus()               # CPUs shown / CPUs the cgroup quota grants
    cands = sorted({max(1, grant), max(1, grant // 2)})
    runs = [timed_fit_and_sample(LAYERS, D, CDIM, HIDDEN, X[:rows], C[:rows], BATCH, t) for t in cands]
    best = max(runs, key=lambda r: r["combined_rows_per_s"])
    import torch
    return dict(value=best["combined_rows_per_s"], unit="rows/s", cores=best["threads"], kind="port",
                sample="oracle/torch_cpu.py (eager PyTorch %s CPU ops in the reference's order, float32) with %d torch threads "
                       "on a %d-CPU host (CPU quota of this process: %d): 1 epoch of 65536-row batches over %d rows (%.1f s = %.1f k rows/s) + sampling %d rows "
                       "(%.1f s = %.1f k rows/s); thread counts tried: %s; in the build container this loop runs at 1.00x the "
                       "reference's own combined rate on the same 8 cores"
                       % (torch.__version__, best["threads"], ncpu, grant, rows, best["t_fit"], best["fit_rows_per_s"] / 1e3, rows,
                          best["t_sample"], best["sample_rows_per_s"] / 1e3,
                          ", ".join("%d: %.1f k" % (r["threads"], r["combined_rows_per_s"] / 1e3) for r in runs)))


def cpu_baseline_oracle(X, C, params, rows_per_thread=8192):
    """Secondary CPU figure: the C oracle (scalar port of the algorithm) on the host cores: one training step
    (loss + gradient, shards summed, Adam) + sampling, on rows_per_thread rows per core -- the same
    1:1 mix of fit rows and sampled rows as one GPU step.  Threads call into the C library concurrently
    (ctypes releases the GIL); each computes the gradient of its shard like a data-parallel rank would."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import Oracle, Shape
    from probaforms_amd._engine import effective_cpus
    cores = max(1, min(effective_cpus(), 64))
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
                sample="oracle/rnvp_oracle.c (float32, gcc -O2, scalar) on %d threads: 1 training step on %d rows + "
                       "sampling %d rows of the C2 workload, %.1f s; the eager-PyTorch reference itself reaches 72.7 k "
                       "(fit) / 199 k (sample) rows/s on 8 cores of the build container (BASELINE.md)" % (cores, rows, rows, dt))


def api_level(Xh, Ch, dev):
    """numpy -> numpy through the reference's class API: fit (upload, cast, host shuffles, epochs, loss history)
    and sample (prior draw, inverse, download).  Never `value`."""
    import torch
    from probaforms_amd.models import RealNVP
    out = {}
    epochs = 10                                           # the reference's default n_epochs (realnvp.py:161)
    for prior in ("host", "device"):
        torch.manual_seed(0)
        m = RealNVP(n_layers=LAYERS, hidden=HIDDEN, batch_size=BATCH, n_epochs=1, lr=1e-3, prior_rng=prior)
        m.fit(Xh, Ch)                                     # first call: model build, allocations
        m.n_epochs = epochs
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter(); m.fit(Xh, Ch); torch.cuda.synchronize(dev); t_fit = time.perf_counter() - t0
        m.sample(Ch)
        t_s = float("inf")
        for _ in range(3):                                # host-side noise (pinned allocations, other processes): best of 3
            t0 = time.perf_counter(); xs = m.sample(Ch); t_s = min(t_s, time.perf_counter() - t0)
        assert xs.shape == (N_ROWS, D) and np.isfinite(xs).all()
        if prior == "host":
            out["fit_rows_per_s"] = N_ROWS * epochs / t_fit
            out["fit_epochs"] = epochs
        out["sample_rows_per_s_%s_prior" % prior] = N_ROWS / t_s
    out["note"] = ("RealNVP(batch_size=65536).fit(X, C) / .sample(C) on the C2 arrays, numpy in -> numpy out; 'host' prior = "
                   "the reference's CPU randn stream, 'device' = counter-based draw inside the inverse kernel")
    return out


def secondary_configs(Xh, Ch, dev):
    """after the timed region, never `value`: BASELINE.json configs[4] (CVAE on the C2 arrays: one fused training step on
    65 536 rows, device resident) and the C2 flow at the reference's DEFAULT batch size of 32 (realnvp.py:161: a latency
    regime: one fused step = three launches)."""
    import torch
    from probaforms_amd import _engine, _hip
    from probaforms_amd.models import CVAE, NormalizingFlow, RealNVPLayer, StandardNormalPrior
    out = {}
    X = torch.from_numpy(Xh).to(dev); C = torch.from_numpy(Ch).to(dev)
    # C5
    torch.manual_seed(0)
    m = CVAE(latent_dim=2, hidden=HIDDEN, batch_size=BATCH, n_epochs=1, lr=1e-3, noise_rng="device")
    m.fit(Xh[:BATCH], Ch[:BATCH])
    core = m._core
    eps = torch.randn(BATCH, 2, device=dev); idx = torch.randperm(N_ROWS, device=dev)[:BATCH].contiguous()
    g = core.grads(); ws = core.workspace(BATCH)
    def cvae_step(t):
        _hip.cvae_loss_grad(core.shape, core.sync(), X, C, idx, eps, BATCH, 1.0 / BATCH, 0.001, g[:core.P], g[core.P:core.P + 1], ws)
        _hip.adam_step(core.sync(), g[:core.P], m.opt.exp_avg[:core.P], m.opt.exp_avg_sq[:core.P], core.P, 1e-3, 0.9, 0.999, 1e-8, 0.0, t)
    for t in range(3):
        cvae_step(t + 1)
    torch.cuda.synchronize(dev)
    _hip.profile_enable(64)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    K = 20
    e0.record()
    for t in range(K):
        cvae_step(t + 4)
    e1.record(); torch.cuda.synchronize(dev)
    n_k, k_ms = _hip.profile_read(_hip.PROFILE_TRAIN)
    _hip.profile_enable(0)
    flop_row = 3 * 2 * ((D + CDIM) * HIDDEN[0] + HIDDEN[0] * 4 + (2 + CDIM) * HIDDEN[0] + HIDDEN[0] * D)
    step_ms = e0.elapsed_time(e1) / K
    out["cvae_c5"] = {"workload": "CVAE latent 2, hidden (128,), d=16 cond=4: loss+grad+Adam on 65536 rows, device resident",
                      "ms_per_step": step_ms, "rows_per_s": BATCH / (step_ms * 1e-3),
                      "kernel_ms": k_ms / max(n_k, 1), "useful_flop_per_row": flop_row,
                      "roofline_frac_f32_mfma": flop_row * BATCH / (k_ms / max(n_k, 1) * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS}
    # C2 flow at batch_size 32
    torch.manual_seed(0)
    layers = [RealNVPLayer(D, CDIM, (torch.arange(D) + i) % 2, HIDDEN, "tanh") for i in range(LAYERS)]
    nf = NormalizingFlow(layers, StandardNormalPrior(D, dev, host_rng=False))
    for p in nf.parameters():
        p.data = p.data.to(dev)
    eng = nf.engine(); opt = _engine.FlatAdam(eng.flat.numel(), dev, lr=1e-3, weight_decay=0.0)
    for bs, nsteps in ((32, 512), (1024, 256)):
        n_small = bs * nsteps
        perm = torch.randperm(n_small, device=dev); losses = torch.zeros(nsteps, device=dev)
        eng.fit_epoch(opt, X[:n_small], C[:n_small], perm, bs, losses)
        torch.cuda.synchronize(dev)
        e0.record()
        eng.fit_epoch(opt, X[:n_small], C[:n_small], perm, bs, losses)
        e1.record(); torch.cuda.synchronize(dev)
        us = e0.elapsed_time(e1) / nsteps * 1e3
        out["c2_batch%d" % bs] = {"workload": "the C2 flow at batch_size=%d%s: %d fused steps in one rnvp_fit_epoch call (tile-split "
                                              "training kernel)" % (bs, " (the reference's default)" if bs == 32 else "", nsteps),
                                  "us_per_step": us, "rows_per_s": bs / (us * 1e-6)}
    return out


def self_launch(args):
    """--gpus N > 1 without a launcher: start the N ranks as children (never exec: the parent has not touched
    the GPU and stays alive only to relay the exit code)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-api-level", action="store_true")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    # stdout carries exactly ONE line, the JSON: libraries that chat on fd 1 (RCCL prints a five-line banner there when a
    # communicator is created) are sent to stderr for the whole run; rank 0 writes the result to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from probaforms_amd._engine import effective_cpus
    # torch sizes its CPU thread pool from os.cpu_count(); under a cgroup quota (16 of 256 CPUs on the MI355X boxes) one
    # parallel host op then gets the whole process throttled for tens of milliseconds
    torch.set_num_threads(max(1, min(torch.get_num_threads(), effective_cpus())))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # BENCH_ONE_GPU=1 (developer aid): run the N-rank code path on a 1-GPU box -- every rank on cuda:0,
    # gloo instead of RCCL.  Never used by the driver; numbers from it mean nothing.
    one_gpu = os.environ.get("BENCH_ONE_GPU") == "1"
    if one_gpu:
        local = 0
    # BENCH_FORCE_DIST=1 (developer aid): a single rank still initialises RCCL and runs the data-parallel step
    # (unfused loss+grad, all-reduce, Adam) -- exercises the N > 1 code path, RCCL included, on a 1-GPU box.
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
    assert world == args.gpus, "WORLD_SIZE=%d but --gpus %d" % (world, args.gpus)
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    from probaforms_amd import _engine, _hip
    from probaforms_amd.models import NormalizingFlow, RealNVPLayer, StandardNormalPrior

    # model: random init of the C2 architecture (same seed on every rank -> identical replicas)
    torch.manual_seed(0)
    layers = [RealNVPLayer(D, CDIM, (torch.arange(D) + i) % 2, HIDDEN, "tanh") for i in range(LAYERS)]
    nf = NormalizingFlow(layers, StandardNormalPrior(D, dev, host_rng=False))
    for p in nf.parameters():
        p.data = p.data.to(dev)
    eng = nf.engine()
    if dp:
        _engine.broadcast_(eng.flat, 0)
        # the communicator and the all-reduce of exactly this message size set themselves up here, outside any timing
        # (with --warmup 0 the first timed step would otherwise carry RCCL's lazy initialisation)
        dist.all_reduce(torch.zeros(eng.P + 1, device=dev), op=dist.ReduceOp.SUM)
        torch.cuda.synchronize()
    opt = _engine.FlatAdam(eng.flat.numel(), dev, lr=1e-3, weight_decay=0.0)

    # data: resident in HBM before timing; each rank owns its own n rows (weak scaling)
    Xh, Ch = make_data(N_ROWS, D, CDIM, seed=rank)
    X = torch.from_numpy(Xh).to(dev); C = torch.from_numpy(Ch).to(dev)
    n_steps = args.steps + args.warmup
    bounds = _engine.batch_bounds(N_ROWS, BATCH)
    nb = len(bounds)
    gen = torch.Generator(device=dev).manual_seed(1 + rank)
    perms = [torch.randperm(N_ROWS, device=dev, generator=gen) for _ in range(n_steps)]    # one shuffle per epoch
    xs = torch.empty(N_ROWS, D, dtype=torch.float32, device=dev)
    losses = torch.zeros(n_steps, nb, device=dev)
    P = eng.P

    def step_dp(i):
        """N ranks: per batch the shard's loss + gradient, the all-reduce of [gradient | loss] on RCCL's stream, and --
        under that all-reduce, which it does not depend on -- the sampling of this batch's 65 536 rows (the 1M sampled
        rows of the step are drawn batch by batch; the counter-based prior makes the chunks tile the one-shot draw);
        then loss read-out + Adam in one launch."""
        perm = perms[i]
        for k, (s, e) in enumerate(bounds):
            g = eng.loss_grad(X, C, perm[s:e], e - s, 1.0 / ((e - s) * world))
            work = dist.all_reduce(g[:P + 1], op=dist.ReduceOp.SUM, async_op=True)
            eng.sample(e - s, C[s:e], 1000 + i, row_offset=rank * N_ROWS + s, out=xs[s:e])
            work.wait()
            eng.finish_dp_step(opt, losses[i, k:k + 1])

    def step_1(i):
        """single GPU: rnvp_fit_epoch (a fused loss + gradient + Adam step per batch, looped in the library),
        then the fused prior draw + inverse, as RealNVP.fit / .sample(prior_rng='device') issue them"""
        eng.fit_epoch(opt, X, C, perms[i], BATCH, losses[i])
        eng.sample(N_ROWS, C, 1000 + i, row_offset=0, out=xs)

    step = step_dp if dp else step_1
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    # HIP events around the hot kernels, recorded by the library on the stream it launches on, for every
    # launch of the timed region
    _hip.profile_enable(args.steps * nb + 8)
    if dp:
        dist.barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k)
    torch.cuda.synchronize()
    if dp:
        dist.barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if dp:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    n_train, train_ms = _hip.profile_read(_hip.PROFILE_TRAIN)
    n_inv, inv_ms = _hip.profile_read(_hip.PROFILE_INVERSE)
    assert n_train == args.steps * nb and n_inv == args.steps * (nb if dp else 1), (n_train, n_inv, args.steps)
    final_loss = float(losses[n_steps - 1, nb - 1].item())
    assert np.isfinite(final_loss), "training diverged"

    if rank == 0:
        # log-prob kernel (not part of fit+sample): measured here, after the timed region
        for _ in range(2):
            eng.forward(X, C, want_z=False, want_logp=True)
        torch.cuda.synchronize()
        _hip.profile_read(_hip.PROFILE_FORWARD)
        for _ in range(5):
            eng.forward(X, C, want_z=False, want_logp=True)
        torch.cuda.synchronize()
        n_fwd, fwd_ms = _hip.profile_read(_hip.PROFILE_FORWARD)
    _hip.profile_enable(0)

    if rank == 0:
        rows_per_step = 2 * N_ROWS * world
        f1 = useful_flops_per_row(D, CDIM, HIDDEN, LAYERS, 1)
        train_tf = 3 * f1 * N_ROWS * args.steps / (train_ms * 1e-3) / 1e12          # fwd + dgrad + wgrad
        inv_tf = f1 * N_ROWS * args.steps / (inv_ms * 1e-3) / 1e12
        fwd_tf = f1 * N_ROWS * n_fwd / (fwd_ms * 1e-3) / 1e12
        path = _hip.kernel_path(eng.shape, eng.masks_host, _hip.OP_TRAIN)
        kname = "k_mfma_train" if path == _hip.PATH_MFMA else "k_generic_train"
        out = {
            "metric": "RealNVP samples/sec (fit+sample)", "value": rows_per_step * args.steps / dt,
            "unit": "rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C2: RealNVP n=1M/GPU d=16 cond=4 L=8 hidden=(128,); one step = one fit epoch over the "
                                   "rank's 1M rows (16 batches of 65536 incl. the ragged one: loss+grad+Adam each) + "
                                   "sampling 1M rows (counter-based prior draw inside the inverse kernel)" +
                                   ("; N > 1: the 1M rows are sampled batch by batch under each gradient all-reduce" if dp else ""),
                       "global_batch": BATCH * world, "parallelism": "dp%d" % world,
                       "kernel_path": "mfma" if path == _hip.PATH_MFMA else "generic", "final_loss": final_loss},
            "roofline": {"bound": "mfma", "achieved": train_tf, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": train_tf / F32_MFMA_PEAK_TFLOPS, "traffic": pmc_traffic(kname),
                         "kernel": "%s (fused forward+backward): %d launches in the timed region, %.3f ms avg (15 of every 16 "
                                   "on 65536 rows, 1 on 16960), %d useful flop/row x %d rows per epoch"
                                   % (kname, n_train, train_ms / n_train, 3 * f1, N_ROWS)},
            "roofline_kernels": {
                "sample (k_mfma_flow inverse, prior drawn in-kernel)": {
                    "bound": "mfma", "achieved": inv_tf, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": inv_tf / F32_MFMA_PEAK_TFLOPS, "ms_per_launch": inv_ms / n_inv,
                    "rows_per_launch": N_ROWS * args.steps // n_inv, "launches": n_inv, "where": "timed region"},
                "log_prob (k_mfma_flow forward)": {
                    "bound": "mfma", "achieved": fwd_tf, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": fwd_tf / F32_MFMA_PEAK_TFLOPS, "ms_per_launch": fwd_ms / n_fwd, "rows_per_launch": N_ROWS,
                    "launches": n_fwd, "where": "after the timed region"}},
            "device_resident": {"fit_rows_per_s": None, "sample_rows_per_s": N_ROWS * world / (inv_ms / args.steps * 1e-3),
                                "note": "sample: kernel time only; fit: ms_per_step minus the sampling kernels' time"},
        }
        out["device_resident"]["fit_rows_per_s"] = N_ROWS * world / max(dt / args.steps - inv_ms / args.steps * 1e-3, 1e-9)
        params = eng.params.detach().cpu().numpy()
        if world == 1 and not force_dist and not args.no_api_level:
            out["api_level"] = api_level(Xh, Ch, dev)
            out["secondary_configs"] = secondary_configs(Xh, Ch, dev)
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(Xh, Ch)
            out["cpu_baseline_c_oracle"] = cpu_baseline_oracle(Xh, Ch, params)
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
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dp:
        dist.barrier()                      # rank 0 may still be timing the CPU baseline
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
