#!/usr/bin/env python3
"""bench.py -- RealNVP fit+sample throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], "C2"): n = 1M rows per GPU of make_moons-shaped tabular data, d=16,
cond=4, 8 coupling layers, hidden=(128,).  One STEP = one pass of the hot path over the rank's data set:
  fit    -- one epoch of RealNVP.fit's batch loop (realnvp.py:235-254): 16 batches of the epoch's shuffled
            rows (15 x 65 536 + the ragged 16 960), each loss + gradient + Adam;
  sample -- RealNVP.sample for the same 1M conditions (realnvp.py:279-282; nflow.py:141-143): the prior
            draw (counter-based device prior, made inside the inverse kernel) + the inverse pass.
X, C and the epoch permutations (the reference's DataLoader shuffles) are resident in HBM before the timed region.  With N > 1 every rank owns
its own 1M rows (weak scaling): per batch each rank computes the gradient of its 65 536-row shard of the
global batch of N x 65 536, the flat [gradient | loss] buffer is all-reduced over RCCL, every rank applies
the same Adam step; sampling shards the global N x 1M rows with no collective.

Batch regimes with N > 1: by default every rank's share of a step stays 65 536 rows (--per-rank-batch; the global batch grows with N);
--global-batch 65536 shares the reference's ONE batch_size (realnvp.py:237; SURVEY.md 8(d)/(e)'s headline) out over the ranks -- 8 192 rows
per rank at 8 GPUs.  A default run with N > 1 also times the global-batch regime after the timed region
(config.global_batch_65536_regime), and exits with code 3 (line still printed, with "error") if the library's RCCL communicator
does not span all ranks or the replicas' parameters differ afterwards.
Timing: exactly --steps steps between barrier + synchronize form a block; when a block is shorter than a second it is repeated up to
about six seconds (the driver's GPU-busy sampler has a ~5 s period) and the MEDIAN block is reported (timed_blocks, block_seconds).

With --gpus N > 1 and no WORLD_SIZE in the environment this script starts its own N ranks (child processes
of `python -m torch.distributed.run`; nothing in the parent touches the GPU) and relays their output.

`value` = rows (fit rows + sampled rows) per second over all GPUs.  The JSON line also carries
  roofline        -- the dominant kernel (fused forward+backward) against the f32 MFMA peak, from HIP events
                     around every one of its launches inside the timed region;
  roofline_kernels-- the same for the sampling kernel (timed region) and the log-prob kernel (measured after it);
  cpu_baseline    -- the reference's CPU path (oracle/torch_cpu.py: eager PyTorch CPU ops in the reference's order,
                     validated against the reference in the build container) on the host cores of rank 0's box, on a
                     bounded sample of the same workload; cpu_baseline_c_oracle: the scalar C oracle on all cores;
  api_level       -- numpy in -> numpy out rates of RealNVP.fit / .sample on the same data (N = 1 only); combined_rows_per_s = one fit
                     epoch + one sample(1M) with the REFERENCE-EXACT prior stream: the number to read beside `value`;
  logprob_mae     -- second half of the metric: per-row log-prob of the HIP path against the oracle;
  secondary_configs -- measured after the timed region, never `value` (N = 1, workload c2 only): the CVAE of configs[4], the C2 flow at
                     the reference's default batch size, the reference's default networks, the C2 arrays through a flow with two
                     hidden layers (any-shape kernels), the f32 / bx3 A/B, C3 and C4.
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

# workloads (BASELINE.json configs[1..3]); per GPU: rows, d, cond, layers, hidden, what one step does.  C3 is quoted on 8M rows
# over 8 GPUs = 1M rows per GPU, C4 on 16M draws over 8 GPUs = 2M per GPU (L and hidden as SURVEY.md 8 reads them).
WORKLOADS = {
    "c2": dict(n=1_000_000, d=16, c=4, L=8, hidden=(128,), fit=True, sample=True, label="C2"),
    "c3": dict(n=1_000_000, d=32, c=8, L=12, hidden=(256,), fit=True, sample=True, label="C3"),
    "c4": dict(n=2_000_000, d=64, c=16, L=8, hidden=(128,), fit=False, sample=True, label="C4"),
}
# the default workload (C2, the configuration BASELINE.json's metric is quoted on for one GPU); main() rebinds these
N_ROWS, D, CDIM, LAYERS, HIDDEN = 1_000_000, 16, 4, 8, (128,)
WORKLOAD_FIT = True
BATCH = 65_536
F32_MFMA_PEAK_TFLOPS = 157.3        # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
# split-bf16 GEMM1 (precision 'bx3'): six bf16 products per f32 product on the dense bf16 MFMA peak (16x the f32 one)
BX3_EFFECTIVE_TFLOPS = 16.0 * F32_MFMA_PEAK_TFLOPS / 6.0
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r06_traffic_pmc.json")             # the C2 step (this command, default workload)
TRAFFIC_FILE_C3C4 = os.path.join(ROOT, "profiles", "r06_traffic_pmc_c3c4.json")   # the C3 / C4 kernels (scripts/bench_kernels.py)


def mixed_bound_seconds_per_row(d, c, hidden, L, passes=1):
    """roofline time per row of a bx3 forward / inverse kernel: GEMM1 (useful 4 h (d/2 + c) flop per layer) on the
    six-product bf16 form, GEMM2 (4 h d/2) on f32 MFMA -- the bound such a kernel should be priced against, beside the
    all-f32 one"""
    h = hidden[0]
    g1 = 4 * h * (d / 2 + c) * L * passes
    g2 = 4 * h * (d / 2) * L * passes
    return g1 / (BX3_EFFECTIVE_TFLOPS * 1e12) + g2 / (F32_MFMA_PEAK_TFLOPS * 1e12)


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


def train_mixed_bound_seconds_per_row(d, c, hidden, L):
    """roofline time per row of a training step whose FORWARD GEMM1 runs on split-bf16 MFMA (the backward, incl. its GEMM1
    recompute, and GEMM2 stay on f32 MFMA): the forward GEMM1's useful flops on the six-product bf16 form, the other
    3 F - that on the f32 peak"""
    h = hidden[0]
    g1 = 4 * h * (d / 2 + c) * L
    rest = 3 * 4 * h * (d + c) * L - g1
    return g1 / (BX3_EFFECTIVE_TFLOPS * 1e12) + rest / (F32_MFMA_PEAK_TFLOPS * 1e12)


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


def tile_geometry(d, c):
    """(NF, CQ) of the register-chained kernels' template arguments (csrc/rnvp_mfma.h pick_tiles)"""
    if d <= 16 and c == 0: return 2, 0
    if d <= 16 and c <= 4: return 2, 1
    if d <= 32 and c <= 8: return 4, 2
    return 8, 4


def kernel_prefix(disp, d, c, inverse=None):
    """the instantiation rocprofv3 lists for a dispatch record: name<NF, CQ, row tiles[, INVERSE]"""
    nf, cq = tile_geometry(d, c)
    p = "%s<%d, %d, %d" % (disp["kernel"], nf, cq, disp["row_tiles"])
    return p if inverse is None else p + (", true" if inverse else ", false")


def pmc_traffic(kernel_prefix, traffic_file=None):
    """HBM bytes per launch of a kernel from the committed PMC profile of this same command
    (scripts/gpu_traffic.sh: separate --pmc passes for FETCH_SIZE and WRITE_SIZE, FETCH_SIZE doubled per
    MI355X_MICROARCH.md).  bench.py cannot collect counters itself; the file's number is reported only while
    it was taken with exactly these kernel sources (csrc_hash) and names this kernel, else None."""
    try:
        d = json.load(open(traffic_file or TRAFFIC_FILE))
    except (OSError, ValueError):
        return None
    if d.get("csrc_hash") != csrc_hash():
        return None
    # several instantiations share a prefix (the ragged batch runs another row-tile count): the one with the most launches
    hits = [(v.get("launches", 0), float(v["hbm_bytes_per_launch"])) for k, v in d.get("kernels", {}).items() if k.startswith(kernel_prefix)]
    return max(hits)[1] if hits else None


def cpu_baseline(X, C, rows=4 * BATCH):
    """The reference's CPU path on THIS box's host cores: oracle/torch_cpu.py, an eager-PyTorch
    loop that issues the reference's op sequence -- DataLoader(TensorDataset, shuffle=True) batches of 65 536 rows, per layer
    cat -> Linear -> Tanh -> Linear for both nets, exp, masked affine (int64 masks), MultivariateNormal prior, autograd
    backward, torch.optim.Adam, per-step loss read-back; sampling through the reversed layers -- validated in the build
    container against the reference itself (tests/golden/validate_torch_cpu.py: identical outputs; rows/s 1.00x the
    reference's in round 4, 1.22x in the judge's round-5 run, 1.45x in round 6 on the same 8 shared cores -- never slower, so
    this is a GENEROUS stand-in for the reference's CPU path: DESIGN.md section 7).  Bounded sample: one epoch over `rows` rows of the
    C2 data + sampling `rows` rows (the 1:1 mix of a GPU step), after a one-batch warm-up; two thread counts, best kept."""
    from oracle.torch_cpu import timed_fit_and_sample
    from probaforms_amd._engine import effective_cpus
    ncpu, grant = os.cpu_count() or 2, effective_cpus()               # CPUs shown / CPUs the cgroup quota grants
    cands = sorted({max(1, grant), max(1, grant // 2)})
    runs = [timed_fit_and_sample(LAYERS, D, CDIM, HIDDEN, X[:rows], C[:rows], BATCH, t) for t in cands]
    key = "combined_rows_per_s" if WORKLOAD_FIT else "sample_rows_per_s"        # C4: sampling only
    best = max(runs, key=lambda r: r[key])
    import torch
    return dict(value=best[key], unit="rows/s", cores=best["threads"], kind="port",
                sample="oracle/torch_cpu.py (eager PyTorch %s CPU ops in the reference's order, float32) with %d torch threads "
                       "on a %d-CPU host (CPU quota of this process: %d): 1 epoch of 65536-row batches over %d rows (%.1f s = %.1f k rows/s) + sampling %d rows "
                       "(%.1f s = %.1f k rows/s); thread counts tried: %s; in the build container this loop produces the reference's outputs at 1.00-1.45x the "
                       "reference's own combined rate on the same 8 cores (never slower: a generous baseline)"
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
        t_fit = float("inf")
        for _ in range(3):                                # host-side noise (worker threads starting, allocations): best of 3, as for sample
            t0 = time.perf_counter(); m.fit(Xh, Ch); torch.cuda.synchronize(dev); t_fit = min(t_fit, time.perf_counter() - t0)
        m.sample(Ch)
        t_s = float("inf")
        for _ in range(3):                                # host-side noise (pinned allocations, other processes): best of 3
            t0 = time.perf_counter(); xs = m.sample(Ch); t_s = min(t_s, time.perf_counter() - t0)
        assert xs.shape == (N_ROWS, D) and np.isfinite(xs).all()
        if prior == "host":
            out["fit_rows_per_s"] = N_ROWS * epochs / t_fit
            out["fit_epochs"] = epochs
        out["sample_rows_per_s_%s_prior" % prior] = N_ROWS / t_s
    # the metric's fit+sample at API level with the REFERENCE-EXACT prior stream (the default RealNVP()): rows of one epoch + rows
    # of one sample call over the time both take -- the number to put next to `value`, which times the device-resident path with
    # the counter-based prior
    out["combined_rows_per_s"] = 2 * N_ROWS / (N_ROWS / out["fit_rows_per_s"] + N_ROWS / out["sample_rows_per_s_host_prior"])
    out["note"] = ("RealNVP(batch_size=65536).fit(X, C) / .sample(C) on the C2 arrays, numpy in -> numpy out; 'host' prior = "
                   "the reference's CPU randn stream, 'device' = counter-based draw inside the inverse kernel; combined = one fit "
                   "epoch + one sample(1M) call with the reference-exact prior, 2 x rows / (their times)")
    return out


def reference_stream_sampling(eng, C, dev):
    """the sampling path with the REFERENCE's prior stream (prior_rng='host', the default): torch.randn's bits drawn on the device
    (k_mt19937_uniform + jump-ahead + k_normal_fill_16: nflow.HostStreamOnDevice) + the inverse kernel -- device resident, after
    the timed region; None where this torch build's CPU randn is not the one the device restates"""
    import torch
    from probaforms_amd import _hip
    from probaforms_amd.models.nflow import HostStreamOnDevice
    if not HostStreamOnDevice.usable(dev):
        return None
    g = torch.Generator(); g.manual_seed(123)
    z = torch.empty(N_ROWS, D, dtype=torch.float32, device=dev); xs = torch.empty_like(z)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    best = None
    for _ in range(20):             # warm clocks first (the generator set-up above idles the chip)
        eng.inverse(z, C, out=xs)
    for rep in range(6):
        hs = HostStreamOnDevice(dev, g).begin()
        torch.cuda.synchronize(dev)
        _hip.profile_read(_hip.PROFILE_INVERSE)
        ev[0].record(); hs.draw(z); ev[1].record(); eng.inverse(z, C, out=xs); ev[2].record()
        hs.end()
        torch.cuda.synchronize(dev)
        nk, kms = _hip.profile_read(_hip.PROFILE_INVERSE)
        cur = (ev[0].elapsed_time(ev[2]), ev[0].elapsed_time(ev[1]), kms / max(nk, 1))
        if rep and (best is None or cur[0] < best[0]):
            best = cur
    f1 = useful_flops_per_row(D, CDIM, HIDDEN, LAYERS, 1)
    tot, draw, inv = best
    return {"bound": "mfma", "achieved": f1 * N_ROWS / (inv * 1e-3) / 1e12, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": f1 * N_ROWS / (inv * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS, "ms_per_launch": inv, "rows_per_launch": N_ROWS,
            "prior_draw_ms": draw, "prior_numbers_per_s": N_ROWS * D / (draw * 1e-3),
            "prior_write_GBps": N_ROWS * D * 4 / (draw * 1e-3) / 1e9,
            "draw_plus_inverse_ms": tot, "rows_per_s": N_ROWS / (tot * 1e-3), "dispatch": _hip.last_dispatch(_hip.PROFILE_INVERSE),
            "where": "after the timed region; frac = the inverse kernel alone; the draw (mt19937 on 32 workgroups by jump-ahead + "
                     "torch's Box-Muller blocks) is bound by instruction issue, not by the 64 MB it writes"}


def rank_steps_of_8(dev):
    """SURVEY.md 8(d)/(e)'s data-parallel configuration on ONE GPU: global batch 65 536 over 8 ranks = 8 192 rows per rank and
    step.  What one rank executes per step -- its loss + gradient, the [gradient | loss] all-reduce on the library's RCCL
    communicator (here a ONE-rank communicator: the collective's launch and copy, not its latency across 8 GPUs), loss read-out +
    Adam + re-pack -- through rnvp_fit_epoch_dp, the call RealNVP.fit makes per epoch under torch.distributed; next to the same
    flow's one-GPU step on the whole 65 536-row batch.  8 x 8192 rows per rank-step against 65 536 rows per one-GPU step is the
    scaling this regime can reach before the collective's cross-GPU latency."""
    import torch
    from probaforms_amd import _engine, _hip
    from probaforms_amd.models import NormalizingFlow, RealNVPLayer, StandardNormalPrior
    out = {}
    try:
        comm = _hip.dp_init(_hip.dp_unique_id(), 0, 1)
    except Exception as e:          # librccl not loadable
        return {"error": "no RCCL communicator: %s" % e}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for key in ("c2", "c3"):
        w = WORKLOADS[key]
        d, c, L, hidden = w["d"], w["c"], w["L"], w["hidden"]
        torch.manual_seed(0)
        layers = [RealNVPLayer(d, c, (torch.arange(d) + i) % 2, hidden, "tanh") for i in range(L)]
        nf = NormalizingFlow(layers, StandardNormalPrior(d, dev, host_rng=False))
        for p in nf.parameters():
            p.data = p.data.to(dev)
        eng = nf.engine(); opt = _engine.FlatAdam(eng.flat.numel(), dev, lr=1e-3, weight_decay=0.0)
        gen = torch.Generator(device=dev).manual_seed(5)
        nsteps, rb = 64, 8192
        n = nsteps * rb
        X = torch.randn(n, d, device=dev, generator=gen); C = torch.randn(n, c, device=dev, generator=gen)
        perm = torch.randperm(n, device=dev, generator=gen); losses = torch.zeros(nsteps, device=dev)
        res = {}
        def chunked(kc):          # the exchange in kc chunks of layers, each all-reduce on the communicator's side stream (rnvp_dp_set_chunks)
            _hip.dp_set_chunks(comm, kc)
            eng.fit_epoch_dp(opt, comm, X, C, perm, rb, losses)
            _hip.dp_set_chunks(comm, 1)
        for name, fn, steps, rows in (
                ("rank_step_8192_rows", lambda: eng.fit_epoch_dp(opt, comm, X, C, perm, rb, losses), nsteps, rb),
                ("rank_step_8192_rows_exchange_in_4_chunks", lambda: chunked(4), nsteps, rb),
                ("one_gpu_step_65536_rows", lambda: eng.fit_epoch(opt, X, C, perm, BATCH, losses), n // BATCH, BATCH)):
            for _ in range(3):          # (an epoch of 64 steps is 6-17 ms: three of them before the timed one, for the clock ramp)
                fn()
            torch.cuda.synchronize(dev)
            _hip.profile_enable(4 * nsteps)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize(dev)
            nk, kms = _hip.profile_read(_hip.PROFILE_TRAIN)
            _hip.profile_enable(0)
            us = e0.elapsed_time(e1) / steps * 1e3
            res[name] = {"us_per_step": us, "kernel_us": kms / max(nk, 1) * 1e3, "rows_per_s": rows / (us * 1e-6),
                         "dispatch": _hip.last_dispatch(_hip.PROFILE_TRAIN),
                         "roofline_frac_f32_mfma": 3 * useful_flops_per_row(d, c, hidden, L, 1) * rows / (kms / max(nk, 1) * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS}
        res["eight_rank_steps_over_one_gpu_step"] = 8 * res["rank_step_8192_rows"]["rows_per_s"] / res["one_gpu_step_65536_rows"]["rows_per_s"]
        res["workload"] = "%s: d=%d cond=%d L=%d hidden=%r" % (w["label"], d, c, L, hidden)
        out[key] = res
        del X, C, nf, eng
        torch.cuda.empty_cache()
    try:
        _hip.dp_destroy(comm)
    except Exception:
        pass
    out["note"] = ("per-rank step of the 8-GPU data-parallel configuration (global batch 65536 = 8192 rows per rank) measured on one GPU "
                   "with a one-rank RCCL communicator; the cross-GPU latency of the [gradient | loss] all-reduce is not in it.  "
                   "`..._exchange_in_4_chunks`: the same step with the message cut into 4 groups of layers whose all-reduces run on a side "
                   "stream under the next group's partial sums (rnvp_dp_set_chunks) -- on ONE rank there is no latency to hide, so the "
                   "difference to the line above is what the extra launches cost; the cut exists for the multi-GPU job")
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
    K = 15                                                   # full batches per call
    eps = torch.randn(K * BATCH, 2, device=dev); idx = torch.randperm(N_ROWS, device=dev)[:K * BATCH].contiguous()
    g = core.grads(); ws = core.workspace(BATCH)
    lossK = torch.zeros(K, device=dev)
    def cvae_epoch(t):       # what CVAE.fit issues per epoch on one GPU: cvae_fit_epoch -- per batch the step kernel + one finish launch
        _hip.cvae_fit_epoch(core.shape, core.sync(), X, C, idx, eps, K * BATCH, BATCH, 0.001, g[:core.P], lossK,
                            m.opt.exp_avg[:core.P], m.opt.exp_avg_sq[:core.P], 1e-3, 0.9, 0.999, 1e-8, 0.0, t, ws)
    # (one call is 0.9 ms of GPU work: 40 warm-up calls first -- a single call from an idle chip runs under its clock ramp -- then 8 timed)
    for w_ in range(40):
        cvae_epoch(1 + w_ * K)
    torch.cuda.synchronize(dev)
    _hip.profile_enable(8 * K + 8)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    NCALL = 8
    e0.record()
    for w_ in range(NCALL):
        cvae_epoch(1 + (40 + w_) * K)
    e1.record(); torch.cuda.synchronize(dev)
    n_k, k_ms = _hip.profile_read(_hip.PROFILE_TRAIN)
    _hip.profile_enable(0)
    flop_row = 3 * 2 * ((D + CDIM) * HIDDEN[0] + HIDDEN[0] * 4 + (2 + CDIM) * HIDDEN[0] + HIDDEN[0] * D)
    step_ms = e0.elapsed_time(e1) / (K * NCALL)
    out["cvae_c5"] = {"workload": "CVAE latent 2, hidden (128,), d=16 cond=4: loss+grad+Adam per 65536-row batch, 15 batches in one cvae_fit_epoch call (the call CVAE.fit makes), device resident",
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
        for _ in range(3):          # (warm-up: ~25-40 ms per call)
            eng.fit_epoch(opt, X[:n_small], C[:n_small], perm, bs, losses)
        torch.cuda.synchronize(dev)
        e0.record()
        eng.fit_epoch(opt, X[:n_small], C[:n_small], perm, bs, losses)
        e1.record(); torch.cuda.synchronize(dev)
        us = e0.elapsed_time(e1) / nsteps * 1e3
        out["c2_batch%d" % bs] = {"workload": "the C2 flow at batch_size=%d%s: %d fused steps in one rnvp_fit_epoch call (tile-split "
                                              "training kernel)" % (bs, " (the reference's default)" if bs == 32 else "", nsteps),
                                  "us_per_step": us, "rows_per_s": bs / (us * 1e-6)}
    # the reference's DEFAULT network on 2-d data (README example, BASELINE.json configs[0]): hidden=(10,), 8 layers, batch 32 --
    # small enough that rnvp_fit_epoch runs the whole epoch as one persistent launch (rnvp_resident.hip)
    torch.manual_seed(0)
    d1, c1 = 2, 1
    layers = [RealNVPLayer(d1, c1, (torch.arange(d1) + i) % 2, (10,), "tanh") for i in range(8)]
    nf1 = NormalizingFlow(layers, StandardNormalPrior(d1, dev, host_rng=False))
    for p in nf1.parameters():
        p.data = p.data.to(dev)
    eng1 = nf1.engine(); opt1 = _engine.FlatAdam(eng1.flat.numel(), dev, lr=1e-3, weight_decay=0.0)
    bs, nsteps = 32, 2048
    X1 = X[:bs * nsteps, :d1].contiguous(); C1 = C[:bs * nsteps, :c1].contiguous()
    perm = torch.randperm(bs * nsteps, device=dev); losses = torch.zeros(nsteps, device=dev)
    eng1.fit_epoch(opt1, X1, C1, perm, bs, losses)
    torch.cuda.synchronize(dev)
    e0.record()
    eng1.fit_epoch(opt1, X1, C1, perm, bs, losses)
    e1.record(); torch.cuda.synchronize(dev)
    us = e0.elapsed_time(e1) / nsteps * 1e3
    out["c1_defaults_batch32"] = {"workload": "the reference's default network (d=2, cond=1, hidden=(10,), 8 layers) at its default "
                                              "batch_size=32: %d steps in one rnvp_fit_epoch call" % nsteps,
                                  "one_launch_per_epoch": bool(_hip.fit_epoch_resident(eng1.shape, bs)),
                                  "us_per_step": us, "rows_per_s": bs / (us * 1e-6), "final_loss": float(losses[-1])}
    # CVAE() defaults (cvae.py:145: hidden=(10,), latent 2, batch 32) on the same 2-d rows: cvae_fit_epoch, one launch per epoch
    shape_v = _hip.CvaeShape.make(d1, c1, 2, (10,), "tanh")
    Pv = _hip.cvae_param_count(shape_v)
    gen = torch.Generator(device=dev).manual_seed(3)
    pv = (torch.rand(Pv, device=dev, generator=gen) - 0.5) * 0.4; mv = torch.zeros(Pv, device=dev); vv = torch.zeros(Pv, device=dev)
    epsv = torch.randn(bs * nsteps, 2, device=dev, generator=gen); gv = torch.empty(Pv, device=dev)
    wsv = torch.empty(_hip.cvae_workspace_bytes(shape_v, bs), dtype=torch.uint8, device=dev)
    def cvae_epoch(first):
        _hip.cvae_fit_epoch(shape_v, pv, X1, C1, perm, epsv, bs * nsteps, bs, 0.001, gv, losses, mv, vv, 1e-3, 0.9, 0.999, 1e-8, 0.0, first, wsv)
    cvae_epoch(1)
    torch.cuda.synchronize(dev)
    e0.record()
    cvae_epoch(1 + nsteps)
    e1.record(); torch.cuda.synchronize(dev)
    us = e0.elapsed_time(e1) / nsteps * 1e3
    out["cvae_defaults_batch32"] = {"workload": "the reference's default CVAE (hidden=(10,), latent 2) on d=2, cond=1 rows at its default "
                                                "batch_size=32: %d steps in one cvae_fit_epoch call" % nsteps,
                                    "one_launch_per_epoch": bool(_hip.cvae_fit_epoch_resident(shape_v, bs)),
                                    "us_per_step": us, "rows_per_s": bs / (us * 1e-6), "final_loss": float(losses[-1])}
    # the C2 arrays through a flow with TWO hidden layers (realnvp.py:22-38 builds any depth): outside the register-chained kernels,
    # served by the any-shape MFMA kernels -- the training call on 64-row blocks with in-kernel weight gradients (rnvp_lmm64.hip)
    torch.manual_seed(0)
    hid2 = (128, 128)
    layers = [RealNVPLayer(D, CDIM, (torch.arange(D) + i) % 2, hid2, "tanh") for i in range(LAYERS)]
    nf2 = NormalizingFlow(layers, StandardNormalPrior(D, dev, host_rng=False))
    for p in nf2.parameters():
        p.data = p.data.to(dev)
    eng2 = nf2.engine(); opt2 = _engine.FlatAdam(eng2.flat.numel(), dev, lr=1e-3, weight_decay=0.0)
    K2 = 4
    perm = torch.randperm(K2 * BATCH, device=dev); losses = torch.zeros(K2, device=dev)
    eng2.fit_epoch(opt2, X[:K2 * BATCH], C[:K2 * BATCH], perm, BATCH, losses)
    torch.cuda.synchronize(dev)
    _hip.profile_enable(64)
    e0.record()
    eng2.fit_epoch(opt2, X[:K2 * BATCH], C[:K2 * BATCH], perm, BATCH, losses)
    e1.record(); torch.cuda.synchronize(dev)
    n_k, k_ms = _hip.profile_read(_hip.PROFILE_TRAIN)
    _hip.profile_enable(0)
    flop_row = 3 * 2 * 2 * LAYERS * ((D + CDIM) * hid2[0] + hid2[0] * hid2[1] + hid2[1] * D)      # dense: the any-shape kernels read masks from a table
    step_ms = e0.elapsed_time(e1) / K2
    out["hidden_128x128"] = {"workload": "the C2 arrays, 8 coupling layers with hidden=(128, 128): loss+grad+Adam per 65536-row batch, %d batches "
                                         "in one rnvp_fit_epoch call, device resident (any-shape MFMA kernels)" % K2,
                             "dispatch": _hip.last_dispatch(_hip.PROFILE_TRAIN), "ms_per_step": step_ms, "rows_per_s": BATCH / (step_ms * 1e-3),
                             "kernel_ms": k_ms / max(n_k, 1), "dense_flop_per_row": flop_row,
                             "roofline_frac_f32_mfma": flop_row * BATCH / (k_ms / max(n_k, 1) * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS}
    out["c2_precision_ab"] = precision_ab(Xh, Ch, dev)
    out.update(secondary_c3_c4(dev))
    out["dp8_rank_steps"] = rank_steps_of_8(dev)
    return out


def precision_ab(Xh, Ch, dev):
    """BASELINE.json configs[1] says "bf16": the C2 flow with rnvp_shape.precision = f32 against bx3 (split-bf16 first Linear
    in forward / inverse / sampling and in the forward phase of the training kernel; rnvp_split.h) -- kernel times from the
    library's HIP events and the per-row log-prob MAE of each against the float64 oracle on the same weights.  Plain bf16
    inputs would miss the 1e-5 parity bar by three orders; the split form keeps float32-level accuracy."""
    import torch
    from oracle import Oracle, Shape
    from probaforms_amd import _engine, _hip
    from probaforms_amd.models import NormalizingFlow, RealNVPLayer, StandardNormalPrior
    X = torch.from_numpy(Xh).to(dev); C = torch.from_numpy(Ch).to(dev)
    out = {}
    rows = 4096
    for prec in ("f32", "bx3"):
        torch.manual_seed(0)
        layers = [RealNVPLayer(D, CDIM, (torch.arange(D) + i) % 2, HIDDEN, "tanh") for i in range(LAYERS)]
        nf = NormalizingFlow(layers, StandardNormalPrior(D, dev, host_rng=False), precision=prec)
        for p in nf.parameters():
            p.data = p.data.to(dev)
        eng = nf.engine()
        opt = _engine.FlatAdam(eng.flat.numel(), dev, lr=1e-3, weight_decay=0.0)
        idx = torch.randperm(N_ROWS, device=dev)[:BATCH].contiguous(); loss = torch.zeros(1, device=dev)
        xs = torch.empty(N_ROWS, D, device=dev)
        res = {}
        _hip.profile_enable(64)
        for name, kind, fn, reps, warm in (("train_step_65536_rows", _hip.PROFILE_TRAIN, lambda: eng.train_step(opt, X, C, idx, BATCH, 1.0 / BATCH, loss), 40, 200),
                                           ("log_prob_1M_rows", _hip.PROFILE_FORWARD, lambda: eng.forward(X, C, want_z=False, want_logp=True), 10, 50),
                                           ("sample_1M_rows", _hip.PROFILE_INVERSE, lambda: eng.sample(N_ROWS, C, 5, row_offset=0, out=xs), 10, 50)):
            for _ in range(warm):                  # ~50 ms of the same launches first: the chip's clock ramp ends before the timed loop (launches past the 64 event slots go unrecorded)
                fn()
            torch.cuda.synchronize(dev); _hip.profile_read(kind)
            for _ in range(reps):
                fn()
            torch.cuda.synchronize(dev)
            nk, kms = _hip.profile_read(kind)
            res[name + "_kernel_ms"] = kms / max(nk, 1)
        _hip.profile_enable(0)
        with torch.no_grad():
            lp = nf.log_prob_samples(X[:rows], C[:rows]).cpu().numpy()
        params = eng.params.detach().cpu().numpy()
        _, lp64, _ = Oracle(64).log_prob(Shape.make(LAYERS, D, CDIM, HIDDEN, "tanh"), params, Xh[:rows], Ch[:rows])
        res["logprob_mae_vs_oracle_f64"] = float(np.abs(lp - lp64).mean())
        out[prec] = res
    return out


def secondary_c3_c4(dev):
    """BASELINE.json configs[2..3] on ONE GPU, after the timed region, never `value`: C3 (d=32, cond=8, L=12, hidden 256) --
    one fused training step (loss + gradient + Adam, rnvp_train_step) on 65 536 rows, log_prob and sampling on 1M rows;
    C4 (d=64, cond=16, L=8, hidden 128) -- rnvp_sample of 2M draws (one rank's share of 16M).  Kernel times from the
    library's own HIP events; roofline fractions against the f32 MFMA peak and, for the kernels whose first Linear runs
    on split-bf16 MFMA (precision auto = bx3 for these shapes), against the mixed bound."""
    import torch
    from probaforms_amd import _engine, _hip
    from probaforms_amd.models import NormalizingFlow, RealNVPLayer, StandardNormalPrior
    out = {}
    for key in ("c3", "c4"):
        w = WORKLOADS[key]
        n, d, c, L, hidden = w["n"], w["d"], w["c"], w["L"], w["hidden"]
        torch.manual_seed(0)
        layers = [RealNVPLayer(d, c, (torch.arange(d) + i) % 2, hidden, "tanh") for i in range(L)]
        nf = NormalizingFlow(layers, StandardNormalPrior(d, dev, host_rng=False))
        for p in nf.parameters():
            p.data = p.data.to(dev)
        eng = nf.engine()
        gen = torch.Generator(device=dev).manual_seed(3)
        X = torch.randn(n, d, device=dev, generator=gen); C = torch.randn(n, c, device=dev, generator=gen)
        xs = torch.empty(n, d, device=dev)
        f1 = useful_flops_per_row(d, c, hidden, L, 1)
        res = {"workload": "%s: d=%d cond=%d L=%d hidden=%r, one GPU" % (w["label"], d, c, L, hidden)}
        _hip.profile_enable(64)

        def timed(kind, fn, reps, rows, passes):
            # warm-up of 2 x reps calls (tens of ms): these figures come after seconds of host-side data generation, and a
            # dozen launches from an idle chip run under its clock ramp (the C3 training kernel read 1.25 ms here and 1.15 ms
            # inside the fit epoch of --workload c3 with two warm-up calls)
            for _ in range(2 * reps):
                fn()
            torch.cuda.synchronize(dev)
            _hip.profile_read(kind)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record(); torch.cuda.synchronize(dev)
            nk, kms = _hip.profile_read(kind)
            kms /= max(nk, 1)
            r = {"rows": rows, "ms_per_call": e0.elapsed_time(e1) / reps, "kernel_ms": kms,
                 "rows_per_s": rows / (e0.elapsed_time(e1) / reps * 1e-3), "useful_flop_per_row": passes * f1,
                 "roofline_frac_f32_mfma": passes * f1 * rows / (kms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS}
            r["dispatch"] = _hip.last_dispatch(kind)
            if r["dispatch"]["gemm1_fwd"] == "bx3":     # priced against the bound of what ran as well
                bound = (train_mixed_bound_seconds_per_row(d, c, hidden, L) if kind == _hip.PROFILE_TRAIN
                         else mixed_bound_seconds_per_row(d, c, hidden, L, passes))
                r["roofline_frac_mixed_bound"] = bound * rows / (kms * 1e-3)
            return r

        if w["fit"]:
            opt = _engine.FlatAdam(eng.flat.numel(), dev, lr=1e-3, weight_decay=0.0)
            idx = torch.randperm(n, device=dev, generator=gen)[:BATCH].contiguous()
            loss = torch.zeros(1, device=dev)
            res["train_step_65536_rows"] = timed(_hip.PROFILE_TRAIN,
                                                  lambda: eng.train_step(opt, X, C, idx, BATCH, 1.0 / BATCH, loss), 10, BATCH, 3)
            res["log_prob_1M_rows"] = timed(_hip.PROFILE_FORWARD, lambda: eng.forward(X, C, want_z=False, want_logp=True), 5, n, 1)
        res["sample_%dM_rows" % (n // 1_000_000)] = timed(_hip.PROFILE_INVERSE, lambda: eng.sample(n, C, 77, row_offset=0, out=xs), 5, n, 1)
        _hip.profile_enable(0)
        out[key] = res
        del X, C, xs, nf, eng
        torch.cuda.empty_cache()
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
    ap.add_argument("--steps", type=int, default=100)      # ~0.53 s of timed GPU work at C2 (20 steps left the driver's 1 Hz sampler nothing to see)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-api-level", action="store_true")
    ap.add_argument("--global-batch", type=int, default=None,
                    help="the reference's semantics (realnvp.py:237: ONE batch_size): a global batch of this many rows sharded over "
                         "the ranks -- 65536 is SURVEY.md 8(d)/(e)'s headline (8192 rows per rank at 8 GPUs); every rank still owns "
                         "its own data rows, so an epoch has N times as many steps")
    ap.add_argument("--per-rank-batch", type=int, default=None,
                    help="rows per rank and step (default 65536): the global batch grows with N")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c2",
                    help="c2 (default: the configuration the metric is quoted on), c3 (fit + sample, d=32 cond=8 L=12 hidden 256, "
                         "1M rows per GPU) or c4 (sampling only, d=64 cond=16, 2M draws per GPU)")
    args = ap.parse_args()
    if args.global_batch is not None and args.per_rank_batch is not None:
        ap.error("--global-batch and --per-rank-batch exclude each other")
    global N_ROWS, D, CDIM, LAYERS, HIDDEN, WORKLOAD_FIT
    wl = WORKLOADS[args.workload]
    N_ROWS, D, CDIM, LAYERS, HIDDEN, WORKLOAD_FIT = wl["n"], wl["d"], wl["c"], wl["L"], wl["hidden"], wl["fit"]

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
    comm = None
    if dp:
        _engine.broadcast_(eng.flat, 0)
        # the library's RCCL communicator (what RealNVP.fit uses under torch.distributed over RCCL) and one all-reduce of
        # exactly this message size, outside any timing; BENCH_FORCE_DIST: a one-rank communicator; gloo jobs: none (the
        # per-batch loop over torch.distributed, as RealNVP.fit does there)
        comm = _hip.dp_init(_hip.dp_unique_id(), 0, 1) if force_dist else _engine.dp_communicator(dev)
        if comm is not None:
            _hip.dp_all_reduce(comm, torch.zeros(eng.P + 1, device=dev), eng.P + 1)
        torch.cuda.synchronize()
    opt = _engine.FlatAdam(eng.flat.numel(), dev, lr=1e-3, weight_decay=0.0)

    # data: resident in HBM before timing; each rank owns its own n rows (weak scaling)
    Xh, Ch = make_data(N_ROWS, D, CDIM, seed=rank)
    X = torch.from_numpy(Xh).to(dev); C = torch.from_numpy(Ch).to(dev)
    n_steps = args.steps + args.warmup
    do_fit, do_sample = wl["fit"], wl["sample"]
    # batch geometry: a GLOBAL batch of gbatch rows, of which this rank takes shard_bounds(...) -- by default gbatch = 65536 x N
    # (--per-rank-batch: every rank's share stays 65536 rows), with --global-batch G the reference's one batch_size, shared out
    if args.global_batch is not None:
        if args.global_batch < world:
            raise SystemExit("--global-batch smaller than the number of ranks")
        gbatch, batch_mode = int(args.global_batch), "global"
    else:
        gbatch, batch_mode = int(args.per_rank_batch or BATCH) * world, "per_rank"
    gbounds = _engine.batch_bounds(N_ROWS * world, gbatch)
    shares = [_engine.shard_bounds(s, e, rank, world) for (s, e) in gbounds]       # this rank's slice of every global batch
    assert sum(hi - lo for lo, hi in shares) == N_ROWS or world > 1
    rank_batch = shares[0][1] - shares[0][0]
    nb = len(gbounds) if do_fit else 0
    # one shuffle per epoch, the reference's: DataLoader(shuffle=True)'s RandomSampler seed from the global CPU generator, then
    # torch.randperm of a private generator -- drawn on the device with the host's bits where that is validated (_engine.DeviceShuffle,
    # what RealNVP.fit uses for its first epochs), on the host otherwise; all of them before the timed region
    torch.manual_seed(1 + rank)
    dev_shuffle = _engine.DeviceShuffle.usable(dev)
    perms = []
    if do_fit:
        for _ in range(n_steps):
            seed = _engine.draw_loader_seed()
            p_loc = _engine.DeviceShuffle.draw(N_ROWS, seed, dev) if dev_shuffle else _engine.permutation_from_seed(N_ROWS, seed).to(dev)
            if world == 1:
                perms.append(p_loc)
                continue
            # N ranks, weak scaling: the global data set is the N ranks' 1M-row blocks and a global batch holds 65 536 rows of
            # each.  RealNVP.fit's data-parallel call takes the permutation of the GLOBAL epoch and reads only this rank's
            # share of every batch, so the local shuffle is placed at those positions (the other ranks' entries are
            # never read) and the local block plays the data set.
            p_glob = torch.zeros(N_ROWS * world, dtype=torch.int64, device=dev)
            take = min(sum(hi - lo for lo, hi in shares), N_ROWS)
            pos = torch.cat([torch.arange(lo, hi, device=dev) for lo, hi in shares])[:take]
            p_glob[pos] = p_loc[:take]
            perms.append(p_glob)
    xs = torch.empty(N_ROWS, D, dtype=torch.float32, device=dev)
    losses = torch.zeros(n_steps, max(nb, 1), device=dev)
    P = eng.P

    def step(i):
        """what RealNVP.fit and .sample(prior_rng='device') issue for one epoch: _engine.run_epoch -- one GPU:
        rnvp_fit_epoch (a fused loss + gradient + Adam step per batch, looped in the library); N ranks over RCCL:
        rnvp_fit_epoch_dp (per batch this rank's rows, the all-reduce of [gradient | loss] on the same stream, loss
        read-out + Adam) -- then the fused prior draw + inverse over the rank's rows"""
        if do_fit:
            _engine.run_epoch(eng, opt, comm, X, C, perms[i], gbounds, gbatch, rank, world, losses[i])
        if do_sample:
            eng.sample(N_ROWS, C, 1000 + i, row_offset=rank * N_ROWS, out=xs)
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    # HIP events around the hot kernels, recorded by the library on the stream it launches on, for every
    # launch of the timed region
    MAX_BLOCKS = 512
    TIMED_SECONDS = 6.0                                     # of GPU work between the first and the last clock reading (the driver's GPU-busy sampler has a ~5 s period)
    _hip.profile_enable(args.steps * (nb + 1) + 8)          # one block's launches; read (and reset) after every block
    prof = {"n_train": 0, "train_ms": 0.0, "n_inv": 0, "inv_ms": 0.0}

    def timed_block():
        """EXACTLY args.steps steps between barrier + synchronize on both sides; the MAX over the ranks"""
        torch.cuda.synchronize()
        if dp:
            dist.barrier()
        t0 = time.perf_counter()
        for k in range(args.steps):
            step(args.warmup + k)
        torch.cuda.synchronize()
        if dp:
            dist.barrier()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        if dp:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        a, b = _hip.profile_read(_hip.PROFILE_TRAIN); prof["n_train"] += a; prof["train_ms"] += b      # outside the clock
        a, b = _hip.profile_read(_hip.PROFILE_INVERSE); prof["n_inv"] += a; prof["inv_ms"] += b
        return float(t.item())

    # A block of --steps steps can be far shorter than the driver can see (20 steps of C2 = 0.11 s): the block is then repeated until
    # about TIMED_SECONDS of GPU work have been timed and the MEDIAN block is reported (`steps` stays as passed, `timed_blocks` says
    # how many).  Every rank derives the count from the same max-reduced first block.
    block_s = [timed_block()]
    n_blocks = max(1, min(MAX_BLOCKS, int(np.ceil(TIMED_SECONDS / max(block_s[0], 1e-6)))))
    for _ in range(n_blocks - 1):
        block_s.append(timed_block())
    dt = float(np.median(block_s))
    timed_steps = n_blocks * args.steps

    # N > 1 self-check, after the clock stopped: every rank applied the identical Adam update to the identical all-reduced
    # gradient, so the replicas must still hold the same bits -- a checksum of the flat parameter buffer (and the last batch
    # loss) gathered over the ranks; the first multi-GPU run validates its own exchange this way
    replicas_identical, rccl_ranks = None, 0
    if dp:
        chk = torch.stack([eng.flat.double().sum(), eng.flat.double().abs().sum(),
                           (losses[n_steps - 1, nb - 1].double() if do_fit else torch.zeros((), dtype=torch.float64, device=dev))])
        allchk = [torch.zeros_like(chk) for _ in range(world)]
        if world > 1:
            dist.all_gather(allchk, chk)
        else:
            allchk = [chk]
        replicas_identical = bool(all(torch.equal(a, allchk[0]) for a in allchk))
        rccl_ranks = world if comm is not None else 0
    # the first real multi-GPU run must not print a line for a job that fell back to the per-batch loop or whose replicas drifted
    dp_failure = None
    if world > 1 and not one_gpu:
        if rccl_ranks != world:
            dp_failure = "the library's RCCL communicator spans %d of %d ranks (fell back to the per-batch loop)" % (rccl_ranks, world)
        elif not replicas_identical:
            dp_failure = "the ranks' parameters / last loss differ after the timed region"
    n_train, train_ms, n_inv, inv_ms = prof["n_train"], prof["train_ms"], prof["n_inv"], prof["inv_ms"]
    disp_last_timed = _hip.last_dispatch(_hip.PROFILE_TRAIN) if do_fit else None     # (read before the secondary regime below launches anything)
    # N > 1, default regime: the OTHER regime -- the reference's one global batch_size of 65 536 shared out over the ranks (SURVEY.md
    # 8(d)/(e)) -- is timed as well, after the clock stopped and outside `value`: a few fit epochs of N x 16 steps on the same
    # resident rows, so that one scaling run yields both curves
    other_regime = None
    if world > 1 and do_fit and batch_mode == "per_rank":
        g2 = BATCH
        gb2 = _engine.batch_bounds(N_ROWS * world, g2)
        sh2 = [_engine.shard_bounds(s, e, rank, world) for (s, e) in gb2]
        take2 = min(sum(hi - lo for lo, hi in sh2), N_ROWS)
        pos2 = torch.cat([torch.arange(lo, hi, device=dev) for lo, hi in sh2])[:take2]
        pg2 = torch.zeros(N_ROWS * world, dtype=torch.int64, device=dev)
        pg2[pos2] = torch.randperm(N_ROWS, device=dev)[:take2]
        losses2 = torch.zeros(len(gb2), device=dev)
        k2 = max(1, min(3, args.steps))
        _hip.profile_read(_hip.PROFILE_TRAIN)
        _engine.run_epoch(eng, opt, comm, X, C, pg2, gb2, g2, rank, world, losses2)           # warm-up (workspace for the new batch size)
        torch.cuda.synchronize(); dist.barrier()
        _hip.profile_read(_hip.PROFILE_TRAIN)
        t0 = time.perf_counter()
        for _ in range(k2):
            _engine.run_epoch(eng, opt, comm, X, C, pg2, gb2, g2, rank, world, losses2)
        torch.cuda.synchronize(); dist.barrier()
        t2 = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        dist.all_reduce(t2, op=dist.ReduceOp.MAX)
        nk2, ms2 = _hip.profile_read(_hip.PROFILE_TRAIN)
        other_regime = {"regime": "global batch %d shared out over %d ranks (%d rows per rank and step), fit epochs only, after the timed region"
                                  % (g2, world, sh2[0][1] - sh2[0][0]),
                        "steps_per_epoch": len(gb2), "epochs_timed": k2, "ms_per_fit_epoch": float(t2.item()) / k2 * 1e3,
                        "us_per_step": float(t2.item()) / k2 / len(gb2) * 1e6, "fit_rows_per_s": N_ROWS * world * k2 / float(t2.item()),
                        "training_kernel_us": ms2 / max(nk2, 1) * 1e3, "final_loss": float(losses2[-1].item())}
    # what the library launched for the timed region's last batch / sampling call (rnvp_last_dispatch: written by the launch
    # sites themselves): the kernel names, variants and arithmetic below are the library's statement, not a copy of its rules
    disp_train = disp_last_timed                                                       # the epoch's LAST batch: the ragged one
    disp_inv = _hip.last_dispatch(_hip.PROFILE_INVERSE) if do_sample else None
    if do_fit:
        # ... so the dispatch of a FULL batch is read from one more, untimed, gradient call of that size (same launch rules;
        # no parameter update); the launches per batch stay those of the fit loop
        eng.loss_grad(X, C, perms[0][shares[0][0]:shares[0][1]].contiguous(), rank_batch, 1.0 / gbatch)
        torch.cuda.synchronize()
        full = _hip.last_dispatch(_hip.PROFILE_TRAIN)
        full["launches"] = disp_train["launches"]; full["ragged_batch"] = {k: disp_train[k] for k in ("kernel", "variant", "row_tiles", "grid", "rows")}
        disp_train = full
    assert n_train == timed_steps * nb and n_inv == timed_steps * (1 if do_sample else 0), (n_train, n_inv, timed_steps)
    final_loss = float(losses[n_steps - 1, nb - 1].item()) if do_fit else None
    assert final_loss is None or np.isfinite(final_loss), "training diverged"

    if rank == 0:
        # log-prob kernel (not part of fit+sample): measured here, after the timed region
        for _ in range(20):         # (tens of ms of warm-up: the dispatch queries above let the chip's clock fall)
            eng.forward(X, C, want_z=False, want_logp=True)
        torch.cuda.synchronize()
        _hip.profile_read(_hip.PROFILE_FORWARD)
        for _ in range(10):
            eng.forward(X, C, want_z=False, want_logp=True)
        torch.cuda.synchronize()
        n_fwd, fwd_ms = _hip.profile_read(_hip.PROFILE_FORWARD)
        disp_fwd = _hip.last_dispatch(_hip.PROFILE_FORWARD)
    _hip.profile_enable(0)

    if rank == 0:
        rows_per_step = (int(do_fit) + int(do_sample)) * N_ROWS * world
        f1 = useful_flops_per_row(D, CDIM, HIDDEN, LAYERS, 1)
        train_tf = 3 * f1 * N_ROWS * timed_steps / (train_ms * 1e-3) / 1e12 if do_fit else None     # fwd + dgrad + wgrad
        inv_tf = f1 * N_ROWS * timed_steps / (inv_ms * 1e-3) / 1e12
        fwd_tf = f1 * N_ROWS * n_fwd / (fwd_ms * 1e-3) / 1e12
        path = _hip.kernel_path(eng.shape, eng.masks_host, _hip.OP_TRAIN)
        kname = disp_train["kernel"] if do_fit else None
        train_bx3 = do_fit and disp_train["gemm1_fwd"] == "bx3"
        flow_bx3 = (disp_inv or disp_fwd)["gemm1_fwd"] == "bx3"
        flow_kernel = (disp_inv or disp_fwd)["kernel"]
        fwd_bx3, fwd_kernel = disp_fwd["gemm1_fwd"] == "bx3", disp_fwd["kernel"]
        mixed = mixed_bound_seconds_per_row(D, CDIM, HIDDEN, LAYERS)
        last_share = shares[-1][1] - shares[-1][0]
        batches_text = ("%d of every %d on %d rows, 1 on %d" % (nb - 1, nb, rank_batch, last_share) if last_share != rank_batch
                        else "every launch on %d rows" % rank_batch)
        if do_fit:
            roof = {"bound": "mfma", "achieved": train_tf, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": train_tf / F32_MFMA_PEAK_TFLOPS,
                    "frac_mixed_bound": (train_mixed_bound_seconds_per_row(D, CDIM, HIDDEN, LAYERS) * N_ROWS * timed_steps / (train_ms * 1e-3)
                                         if train_bx3 else None),
                    "traffic": pmc_traffic(kernel_prefix(disp_train, D, CDIM), TRAFFIC_FILE if args.workload == "c2" else TRAFFIC_FILE_C3C4),
                    "dispatch": disp_train,
                    "kernel": "%s (fused forward+backward; variant %s, %d row tiles per wave, %d waves per workgroup; GEMM1 of the "
                              "forward phase on %s, everything else on f32 MFMA): %d launches in the timed region, %.3f ms avg (%s), "
                              "%d useful flop/row x %d rows per epoch; %d kernel launches per "
                              "batch; `frac` prices the useful flops against the f32 MFMA peak%s"
                              % (kname, disp_train["variant"], disp_train["row_tiles"], disp_train["waves"],
                                 "3-term split-bf16 MFMA (float32-level accuracy)" if train_bx3 else "f32 MFMA", n_train,
                                 train_ms / n_train, batches_text, 3 * f1, N_ROWS, disp_train["launches"],
                                 ", `frac_mixed_bound` against the bound of what runs (forward GEMM1 as six bf16 products per f32 "
                                 "product on the dense bf16 peak)" if train_bx3 else "")}
        else:           # sampling only (C4): the inverse kernel is the dominant one
            roof = {"bound": "mfma", "achieved": inv_tf, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": inv_tf / F32_MFMA_PEAK_TFLOPS,
                    # the committed PMC profile has this kernel at 1M rows per launch; scaled to this launch's rows
                    "traffic": (lambda t: None if t is None else t * N_ROWS / 1048576.0)(pmc_traffic(kernel_prefix(disp_inv, D, CDIM, True), TRAFFIC_FILE_C3C4)),
                    "frac_mixed_bound": mixed * N_ROWS * timed_steps / (inv_ms * 1e-3) if flow_bx3 else None,
                    "kernel": "%s inverse with the prior drawn in-kernel: %d launches of %d rows in the timed region, %.3f ms avg, "
                              "%d useful flop/row; `frac` prices it against the f32 MFMA peak, `frac_mixed_bound` against the "
                              "bound of what it executes (GEMM1 as six bf16 products per f32 product on the bf16 peak, "
                              "GEMM2 on f32 MFMA)" % (flow_kernel, n_inv, N_ROWS, inv_ms / n_inv, f1)}
        sample_entry = {"bound": "mfma", "achieved": inv_tf, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": inv_tf / F32_MFMA_PEAK_TFLOPS, "ms_per_launch": inv_ms / max(n_inv, 1),
                        "rows_per_launch": N_ROWS, "launches": n_inv, "where": "timed region"}
        fwd_entry = {"bound": "mfma", "achieved": fwd_tf, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": fwd_tf / F32_MFMA_PEAK_TFLOPS, "ms_per_launch": fwd_ms / n_fwd, "rows_per_launch": N_ROWS,
                     "launches": n_fwd, "where": "after the timed region"}
        sample_entry["dispatch"], fwd_entry["dispatch"] = disp_inv, disp_fwd
        if do_sample:       # HBM bytes per launch of the sampling kernel from the committed PMC profile (None when the sources moved)
            t = pmc_traffic(kernel_prefix(disp_inv, D, CDIM, True), TRAFFIC_FILE if args.workload == "c2" else TRAFFIC_FILE_C3C4)
            sample_entry["traffic"] = None if t is None else t * N_ROWS / 1048576.0
            sample_entry["algorithmic_bytes"] = 4 * N_ROWS * (CDIM + D)     # conditions in, samples out: the prior is drawn in the kernel
        if flow_bx3 and do_sample:
            sample_entry["frac_mixed_bound"] = mixed * N_ROWS * timed_steps / (inv_ms * 1e-3)
        if fwd_bx3:
            fwd_entry["frac_mixed_bound"] = mixed * N_ROWS * n_fwd / (fwd_ms * 1e-3)
        gemm1_text = "3-term split-bf16 MFMA, float32-level accuracy"
        dtype = "f32"
        if train_bx3 or flow_bx3:
            dtype = "f32 (first Linear of the s/t nets on %s in: %s; everything else f32)" % (
                gemm1_text, ", ".join(t for t, on in (("the training kernel's forward phase", train_bx3), ("sampling", flow_bx3 and do_sample),
                                                      ("log-prob", fwd_bx3)) if on))
        step_text = ("one fit epoch over the rank's %d rows (%d batches of %d rows per rank incl. the ragged one: loss+grad+Adam each) + "
                     % (N_ROWS, nb, rank_batch) if do_fit else "") + \
                    "sampling %d rows (counter-based prior draw inside the inverse kernel)" % N_ROWS
        if dp and do_fit:
            step_text += ("; N ranks: _engine.run_epoch, the call RealNVP.fit makes -- " +
                          ("rnvp_fit_epoch_dp on the library's RCCL communicator (loss+grad, all-reduce, Adam per batch on one stream)"
                           if comm is not None else "per-batch loop over torch.distributed (no RCCL communicator in this job)"))
        out = {
            "metric": "RealNVP samples/sec (fit+sample)", "value": rows_per_step * args.steps / dt,
            "unit": "rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "timed_blocks": n_blocks,
            "block_seconds": {"median": dt, "min": float(min(block_s)), "max": float(max(block_s))},
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": "%s: RealNVP n=%d/GPU d=%d cond=%d L=%d hidden=%r; one step = %s"
                                   % (wl["label"], N_ROWS, D, CDIM, LAYERS, HIDDEN, step_text),
                       "global_batch": gbatch if do_fit else None, "rank_batch": rank_batch if do_fit else None,
                       "batch_mode": ("global batch fixed (--global-batch): the reference's one batch_size shared out over the ranks"
                                      if batch_mode == "global" else "per-rank batch fixed (default): the global batch grows with N"),
                       "parallelism": "dp%d" % world, "global_batch_65536_regime": other_regime,
                       "rccl_ranks": rccl_ranks, "replicas_identical": replicas_identical,
                       "kernel_path": "mfma" if path == _hip.PATH_MFMA else "generic", "final_loss": final_loss},
            "roofline": roof,
            "roofline_kernels": {
                "sample (%s inverse, prior drawn in-kernel)" % flow_kernel: sample_entry,
                "log_prob (%s forward)" % fwd_kernel: fwd_entry},
            "device_resident": {"fit_rows_per_s": None, "sample_rows_per_s": N_ROWS * world / (inv_ms / timed_steps * 1e-3),
                                "note": "sample: kernel time only; fit: ms_per_step minus the sampling kernels' time"},
        }
        if do_fit:
            out["device_resident"]["fit_rows_per_s"] = N_ROWS * world / max(dt / args.steps - inv_ms / timed_steps * 1e-3, 1e-9)
        if do_sample and not args.no_api_level:
            _hip.profile_enable(64)
            ref_stream = reference_stream_sampling(eng, C, dev)
            _hip.profile_enable(0)
            if ref_stream is not None:
                out["roofline_kernels"]["sample with the reference's prior stream (mt19937 + Box-Muller on the device, then %s inverse)" % flow_kernel] = ref_stream
        params = eng.params.detach().cpu().numpy()
        if world == 1 and not force_dist and not args.no_api_level and args.workload == "c2":
            out["api_level"] = api_level(Xh, Ch, dev)
            out["secondary_configs"] = secondary_configs(Xh, Ch, dev)
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(Xh, Ch)
            if args.workload == "c2":
                out["cpu_baseline_c_oracle"] = cpu_baseline_oracle(Xh, Ch, params)
        # second half of BASELINE.json's metric (every workload, whatever else is switched off): per-row log-prob MAE of
        # the HIP path against the CPU restatement of the reference (float32 oracle, and its float64 referee) on the
        # weights the run ends with (trained where the workload fits, the random init for c4), 4096 rows
        from oracle import Oracle, Shape
        rows = 4096
        lp = nf.log_prob_samples(X[:rows], C[:rows]).detach().cpu().numpy()
        sh = Shape.make(LAYERS, D, CDIM, HIDDEN, "tanh")
        _, lp32, _ = Oracle(32).log_prob(sh, params, Xh[:rows], Ch[:rows])
        _, lp64, _ = Oracle(64).log_prob(sh, params, Xh[:rows], Ch[:rows])
        out["logprob_mae"] = {"vs_oracle_f32": float(np.abs(lp - lp32).mean()),
                              "vs_oracle_f64": float(np.abs(lp - lp64).mean()),
                              "oracle_f32_vs_f64": float(np.abs(lp32 - lp64).mean()), "rows": rows,
                              "mean_abs_logp": float(np.abs(lp64).mean()),
                              # north_star: 1e-5.  For the d >= 32 shapes |log p| runs to several hundred, one float32 ulp there is
                              # 3e-5 ... 6e-5, and the float32 oracle ITSELF sits 1.2e-5 from its float64 referee (oracle_f32_vs_f64):
                              # the tests' bar for those shapes is 3e-5 AND no further from float64 than 3x the reference's own distance
                              "target": 1e-5 if args.workload == "c2" else 3e-5}
        # the two batch regimes, where nobody can miss them: `value` is measured in `batch_regime`; the other one is reported beside
        # it (N > 1: timed after the clock stopped; N = 1: what ONE rank of an 8-GPU job executes per step, secondary_configs)
        out["batch_regime"] = ("weak batch: %d rows per rank and step, the global batch grows with N" % rank_batch if batch_mode == "per_rank"
                               else "strong batch: the reference's ONE global batch of %d rows shared out over the %d ranks" % (gbatch, world)) if do_fit else None
        if other_regime is not None:
            out["strong_batch_regime_global_65536"] = other_regime
        d8 = out.get("secondary_configs", {}).get("dp8_rank_steps")
        if d8:
            out["strong_batch_8gpu_projection"] = {
                "eight_rank_steps_over_one_gpu_step": {k: v["eight_rank_steps_over_one_gpu_step"] for k, v in d8.items() if isinstance(v, dict) and "eight_rank_steps_over_one_gpu_step" in v},
                "note": "global batch 65536 over 8 ranks = 8192 rows per rank: 8 x (rows/s of one rank's step, measured on this GPU through "
                        "rnvp_fit_epoch_dp on a one-rank RCCL communicator) / (rows/s of the one-GPU 65536-row step); the cross-GPU all-reduce "
                        "latency comes on top -- north_star's >= 6x is NOT projected in this regime, the weak-batch regime keeps the one-GPU step"}
        if dp_failure:
            out["error"] = dp_failure
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dp:
        dist.barrier()                      # rank 0 may still be timing the CPU baseline
        dist.destroy_process_group()
    if dp_failure:
        sys.stderr.write("bench.py: " + dp_failure + "\n")
        sys.exit(3)


if __name__ == "__main__":
    main()
