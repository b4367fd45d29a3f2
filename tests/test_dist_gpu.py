"""Data-parallel fit on the real HIP kernels: two ranks launched with
torch.distributed.run share cuda:0 over gloo and must reproduce the single-process fit; bench.py's
N-rank path runs the same way (BENCH_ONE_GPU=1)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(nproc, script_args, extra_env):
    env = dict(os.environ, **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + script_args
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)


@pytest.mark.timeout(900)
def test_two_ranks_on_hip_kernels_match_single_process(tmp_path):
    out = str(tmp_path / "dp.npz")
    r = _launch(2, [os.path.join(ROOT, "tests", "_dist_gpu_worker.py"), out], {"device": "cuda:0"})
    assert r.returncode == 0, r.stderr[-3000:]
    dp = np.load(out)
    assert bool(dp["same"])                                  # replicas bit-identical after training
    assert dp["hist"].shape == (22,)                         # 11 batches x 2 epochs
    from probaforms_amd.models import RealNVP
    rng = np.random.default_rng(0)
    X = rng.standard_normal((1000, 5)); C = rng.standard_normal((1000, 3))
    torch.manual_seed(0)
    m = RealNVP(n_layers=4, hidden=(16,), batch_size=96, n_epochs=2, lr=1e-2, weight_decay=0.05)
    m.fit(X, C)
    hist = np.array([float(v) for v in m.loss_history])
    flat = m.nf.engine().flat.detach().cpu().numpy()
    # only the summation order of the two shards differs from the single-process gradient
    np.testing.assert_allclose(dp["hist"], hist, rtol=2e-5, atol=2e-5)
    assert np.abs(dp["flat"] - flat).max() < 5e-4 and np.abs(dp["flat"] - flat).mean() < 2e-5
    torch.manual_seed(5)
    xs = m.sample(C[:64])
    assert np.abs(xs - dp["xs"]).max() < 5e-3
    # CVAE data parallel (ranks seeded differently): replicas identical, equal to the single process seeded like rank 0
    from probaforms_amd.models import CVAE
    assert bool(dp["cvae_same"])
    torch.manual_seed(0)
    cv = CVAE(latent_dim=2, hidden=(16,), batch_size=96, n_epochs=2, lr=1e-2)
    cv.fit(X, C)
    np.testing.assert_allclose(dp["cvae_hist"], np.array([float(v) for v in cv.loss_history]), rtol=5e-4, atol=5e-4)
    cf = cv._core.flat.detach().cpu().numpy()
    assert np.abs(dp["cvae_flat"] - cf).max() < 2e-3 and np.abs(dp["cvae_flat"] - cf).mean() < 5e-5
    # the reference's own fit of the C2 flow at batch_size=32 (tests/golden/c2_fit.npz), here on two ranks
    ref = np.load(os.path.join(ROOT, "tests", "golden", "c2_fit.npz")); got = np.load(out + ".c2fit.npz")
    assert got["hist"].shape == (16,)
    np.testing.assert_allclose(got["hist"][:4], ref["loss_history"][:4], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(got["hist"], ref["loss_history"], rtol=1e-3, atol=1e-3)
    assert np.abs(got["flat"][:ref["params_after"].size] - ref["params_after"]).max() < 2e-3
    # round 6: the exchange cut into chunks of layers gives the one-message loop's bits on both ranks
    assert bool(np.load(out + ".chunks.npz")["same"])
    # sharded sampling: rank shares are consecutive blocks of the replicated draw; 'gather' rebuilds all of it
    s0, s1 = np.load(out + ".rank0.npz"), np.load(out + ".rank1.npz")
    assert s0["shard"].shape == (31, 5) and s1["shard"].shape == (30, 5)
    for s_ in (s0, s1):
        np.testing.assert_array_equal(s_["gather"], s_["full"])
    np.testing.assert_array_equal(np.concatenate([s0["shard"], s1["shard"]]), s0["full"])
    # device prior, both ranks seeded alike: the shares tile the single-process device draw bit for bit; no row repeats
    shares = np.concatenate([s0["dev_shard"], s1["dev_shard"]])
    np.testing.assert_array_equal(shares, s0["dev_full"])
    np.testing.assert_array_equal(s1["dev_gather"], s0["dev_full"])
    assert len({row.tobytes() for row in shares}) == 61
    assert not np.array_equal(s0["dev_shard"][:30], s1["dev_shard"])


@pytest.mark.timeout(900)
def test_bench_self_launches_two_ranks_and_prints_one_json_line():
    """`python3 bench.py --gpus 2` with NO launcher (the driver's form): the script starts its own ranks"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["BENCH_ONE_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["scaling"] == "weak" and j["value"] > 0
    assert j["config"]["global_batch"] == 2 * 65536 and j["roofline"]["frac"] > 0
    assert len(j["roofline_kernels"]) >= 2 and all(v["frac"] > 0 for v in j["roofline_kernels"].values())
    assert j["config"]["rank_batch"] == 65536 and j["config"]["replicas_identical"] is True and j["timed_blocks"] >= 1
    o = j["config"]["global_batch_65536_regime"]            # the other regime is timed too, outside `value`
    assert o["steps_per_epoch"] == 31 and o["ms_per_fit_epoch"] > 0 and np.isfinite(o["final_loss"])


@pytest.mark.timeout(900)
def test_bench_global_batch_mode_shares_one_batch_size_out_over_the_ranks():
    """--global-batch 65536 (the reference's ONE batch_size, realnvp.py:237; SURVEY 8(d)/(e)'s headline): every rank takes
    65536 / N rows of each step, the replicas stay bit-identical, the line says which regime it timed"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["BENCH_ONE_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                        "--global-batch", "65536", "--no-cpu-baseline", "--no-api-level"], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert j["n_gpus"] == 2 and j["config"]["global_batch"] == 65536 and j["config"]["rank_batch"] == 32768
    assert j["config"]["replicas_identical"] is True and j["config"]["batch_mode"].startswith("global batch fixed")
    assert j["roofline"]["dispatch"]["rows"] == 32768


@pytest.mark.timeout(900)
def test_bench_under_an_external_launcher():
    r = _launch(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"],
                {"BENCH_ONE_GPU": "1"})
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2


@pytest.mark.timeout(900)
def test_bench_data_parallel_step_over_rccl_single_rank():
    """the N-rank step of bench.py with the real collective library: one rank, RCCL initialised, gradient
    all-reduce on RCCL's stream under the sampling kernel (BENCH_FORCE_DIST=1)"""
    env = dict(os.environ, BENCH_FORCE_DIST="1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert j["n_gpus"] == 1 and j["value"] > 0 and j["roofline"]["frac"] > 0


@pytest.mark.parametrize("cfg", ["c2", "c3"])
def test_chunked_exchange_is_the_one_message_loop_bit_for_bit(cfg):
    """rnvp_dp_set_chunks / rnvp_fit_epoch_dp_cb_chunked (round 6): per batch the partial sums, the exchange and Adam + re-pack run
    per chunk of layers -- on the library's one-rank RCCL communicator with the chunks' all-reduces on its side stream, and through
    the caller's exchange -- and must leave exactly the parameters, moments and losses of the unchunked loop (and of
    rnvp_fit_epoch, the single-GPU call)"""
    from probaforms_amd import _engine, _hip
    from probaforms_amd.models import NormalizingFlow, RealNVPLayer, StandardNormalPrior
    L, d, c, h = (8, 16, 4, 128) if cfg == "c2" else (12, 32, 8, 256)
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(4)
    n = 3 * 8192 + 1000
    X = torch.randn(n, d, generator=g).to(dev); C = torch.randn(n, c, generator=g).to(dev)
    perm = torch.randperm(n, generator=g).to(dev)
    comm = _hip.dp_init(_hip.dp_unique_id(), 0, 1)

    def run(mode, chunks):
        torch.manual_seed(0)
        layers = [RealNVPLayer(d, c, (torch.arange(d) + i) % 2, (h,), "tanh") for i in range(L)]
        nf = NormalizingFlow(layers, StandardNormalPrior(d, dev))
        for p in nf.parameters():
            p.data = p.data.to(dev)
        eng = nf.engine()
        opt = _engine.FlatAdam(eng.flat.numel(), dev, lr=1e-3, weight_decay=0.0)
        losses = torch.zeros(4, device=dev)
        for _ in range(2):                       # two epochs: the re-packed fragments of one feed the next
            if mode == "single":
                eng.fit_epoch(opt, X, C, perm, 8192, losses)
            elif mode == "rccl":
                _hip.dp_set_chunks(comm, chunks)
                eng.fit_epoch_dp(opt, comm, X, C, perm, 8192, losses)
            else:
                eng.fit_epoch_dp(opt, None, X, C, perm, 8192, losses, exchange=lambda t, count: None, rank=0, world=1, chunks=chunks)
        torch.cuda.synchronize()
        assert bool(torch.isfinite(losses).all())
        return torch.cat([eng.flat.detach().reshape(-1), opt.exp_avg.reshape(-1), opt.exp_avg_sq.reshape(-1), losses]).clone()

    try:
        base = run("rccl", 1)
        for mode, chunks in (("rccl", 2), ("rccl", 4), ("rccl", 8), ("cb", 1), ("cb", 3), ("single", 1)):
            got = run(mode, chunks)
            assert torch.equal(got, base), (mode, chunks, float((got - base).abs().max()))
    finally:
        _hip.dp_set_chunks(comm, 1)
        _hip.dp_destroy(comm)
