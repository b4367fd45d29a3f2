"""CPU-side contracts: init order (G1), state_dict layout, shuffle (G5), prior stream (G6),
sharding arithmetic, and that the C-ABI library exports every symbol of include/rnvp_hip.h."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch
from cases import CASES
from conftest import GOLDEN, ROOT, load_case

from probaforms_amd import _engine, _hip
from probaforms_amd.models import GenModel, RealNVP, RealNVPLayer, StandardNormalPrior, gen_network


def test_library_exports_every_declared_symbol():
    hdr = "".join(open(os.path.join(ROOT, "include", f)).read() for f in ("rnvp_hip.h", "cvae_hip.h"))
    declared = set(re.findall(r"\b((?:rnvp|cvae)_[a-z_]+)\s*\(", hdr))
    assert declared == set(_hip.EXPORTS), declared ^ set(_hip.EXPORTS)
    lib = ctypes.CDLL(_hip.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert _hip.lib().rnvp_version() == _hip.ABI_VERSION
    assert int(re.search(r"#define RNVP_HIP_VERSION (\d+)", hdr).group(1)) == _hip.ABI_VERSION
    assert b"workspace" in _hip.lib().rnvp_status_string(-3)


def test_param_count_matches_reference_shapes():
    from cases import param_count
    for name, (L, d, c, hidden, act, _) in CASES.items():
        s = _hip.RnvpShape.make(L, d, c, hidden, act)
        assert _hip.param_count(s) == param_count(L, d, c, hidden)
        for op in (_hip.OP_FORWARD, _hip.OP_INVERSE, _hip.OP_TRAIN):
            assert _hip.workspace_bytes(s, op, 4096) >= 0


def test_missing_device_fails_loudly():
    """no CPU fallback: CPU tensors are rejected before any kernel is launched"""
    s = _hip.RnvpShape.make(1, 2, 0, (4,), "tanh")
    x = torch.zeros(3, 2)
    with pytest.raises(RuntimeError, match="HIP device"):
        _hip.forward_logprob(s, torch.zeros(_hip.param_count(s)), torch.zeros(2, dtype=torch.uint8), x, None,
                             None, 3, torch.empty(3, 2), None, None, None, None)
    with pytest.raises(RuntimeError, match="HIP device only"):
        _engine.require_hip(torch.device("cpu"))


@pytest.mark.parametrize("name", [n for n in CASES if CASES[n][5] == "torch"])
def test_init_order_matches_reference(name):
    """G1: same nn.Linear creation order (nn_t then nn_s, layer by layer) => same seeded init."""
    L, d, c, hidden, act, _ = CASES[name]
    torch.manual_seed(0)
    layers = [RealNVPLayer(d, c, (torch.arange(d) + i) % 2, hidden, act) for i in range(L)]
    flat = torch.cat([p.detach().reshape(-1) for l in layers for p in l.parameters()]).numpy()
    gold = np.load(os.path.join(GOLDEN, "case_%s.npz" % name))
    assert np.array_equal(flat, gold["G1_params"])
    masks = np.stack([l.mask.numpy() for l in layers])
    assert masks.dtype == np.int64 and np.array_equal(masks.astype(np.uint8), gold["masks"])


def test_state_dict_keys_and_layout():
    layer = RealNVPLayer(5, 3, torch.arange(5) % 2, (10,), "tanh")
    keys = list(layer.state_dict())
    assert keys == ["nn_t.0.weight", "nn_t.0.bias", "nn_t.2.weight", "nn_t.2.bias",
                    "nn_s.0.weight", "nn_s.0.bias", "nn_s.2.weight", "nn_s.2.bias"]
    assert layer.nn_t[0].weight.shape == (10, 8) and layer.nn_s[2].weight.shape == (5, 10)
    assert "mask" not in layer.state_dict()                       # plain attribute, realnvp.py:68
    net = gen_network(4, 2, (7, 9), "relu")
    assert [type(m).__name__ for m in net] == ["Linear", "ReLU", "Linear", "ReLU", "Linear"]
    net = gen_network(4, 2, (7,), "anything-else")
    assert type(net[1]).__name__ == "ReLU"                        # realnvp.py:34-37


def test_constructor_defaults_and_registry():
    m = RealNVP()
    assert (m.n_layers, m.hidden, m.activation, m.batch_size, m.n_epochs, m.lr, m.weight_decay, m.verbose) == \
        (8, (10,), 'tanh', 32, 10, 0.0001, 0, 0)
    assert m.prior is None and m.nf is None and m.opt is None and m.loss_history == []
    assert RealNVP in GenModel.__subclasses__()
    assert GenModel().fit(None, None) is None and GenModel().sample(None) is None


def test_loader_permutation_bit_exact():
    """G5: identical batch composition and identical consumption of the global CPU generator."""
    f = np.load(os.path.join(GOLDEN, "loader_indices.npz"))
    for seed in (0, 7):
        for n in (100, 103, 1000):
            torch.manual_seed(seed)
            for e in range(3):
                perm = _engine.loader_permutation(n)
                assert perm.dtype == torch.int64
                assert np.array_equal(perm.numpy(), f["seed%d_n%d" % (seed, n)][e])
            assert np.array_equal(torch.randn(4).numpy(), f["seed%d_n%d_next" % (seed, n)])


def test_batch_and_shard_bounds():
    b = _engine.batch_bounds(1000, 32)
    assert len(b) == 32 and b[-1] == (992, 1000) and b[0] == (0, 32)          # 31 x 32 + 8 (A8)
    assert _engine.batch_bounds(64, 32) == [(0, 32), (32, 64)]
    for (s, e) in [(0, 32), (992, 1000), (5, 6), (10, 10)]:
        for world in (1, 2, 3, 8):
            parts = [_engine.shard_bounds(s, e, r, world) for r in range(world)]
            assert parts[0][0] == s and parts[-1][1] == e
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


def test_prior_stream_and_logprob():
    """G6: prior.sample == randn on the global CPU generator; log_prob == closed form."""
    f = np.load(os.path.join(GOLDEN, "prior.npz"))
    for seed, n, d in ((0, 7, 2), (3, 33, 5), (0, 16, 16), (1, 5, 1)):
        torch.manual_seed(seed)
        pr = StandardNormalPrior(d, "cpu")
        assert np.array_equal(pr.sample((n,)).numpy(), f["seed%d_n%d_d%d" % (seed, n, d)])
        z = torch.from_numpy(f["logprob_in_d%d" % d])
        np.testing.assert_allclose(pr.log_prob(z).numpy(), f["logprob_out_d%d" % d], rtol=1e-6, atol=1e-6)


def test_flatten_parameters_keeps_state_dict_live():
    layer = RealNVPLayer(4, 2, torch.arange(4) % 2, (8,), "tanh")
    plist = list(layer.parameters())
    before = [p.detach().clone() for p in plist]
    flat = _engine.flatten_parameters(plist, "cpu")
    assert _engine.is_flat(plist, flat) and flat.numel() % 4 == 0
    for p, b in zip(plist, before):
        assert torch.equal(p, b)
    flat.zero_()
    assert all(float(p.abs().sum()) == 0 for p in layer.state_dict().values())   # same memory
    layer.nn_t[0].weight.data = torch.ones(8, 6)                                   # user replaces a tensor
    assert not _engine.is_flat(plist, flat)


def test_flatten_parameters_one_gather_pads_and_casts():
    """freshly built (host) parameters are gathered and moved in one copy: reference order, float32, zero padding to a
    multiple of four floats, every parameter a live view"""
    ps = [torch.nn.Parameter(torch.arange(3, dtype=torch.float64).reshape(3, 1) / 7), torch.nn.Parameter(torch.tensor([0.5, -2.0]))]
    flat = _engine.flatten_parameters(ps, "cpu")
    assert flat.dtype == torch.float32 and flat.numel() == 8 and _engine.is_flat(ps, flat)
    assert torch.equal(flat, torch.tensor([0.0, 1 / 7, 2 / 7, 0.5, -2.0, 0.0, 0.0, 0.0], dtype=torch.float32))
    assert ps[0].shape == (3, 1) and ps[1].shape == (2,)
    flat[3] = 9.0
    assert float(ps[1][0]) == 9.0


def test_install_as_probaforms_import_path():
    """drop-in name: after install_as_probaforms() the reference's import line resolves to the build"""
    import subprocess, sys
    code = ("import probaforms_amd; probaforms_amd.install_as_probaforms();"
            "from probaforms.models import RealNVP, CVAE;"
            "from probaforms.models.realnvp import RealNVPLayer;"
            "from probaforms.models.interfaces import GenModel;"
            "import probaforms_amd.models as m;"
            "assert RealNVP is m.RealNVP and CVAE is m.CVAE and issubclass(RealNVP, GenModel); print('ok')")
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr


def test_chunked_host_randn_is_the_same_stream():
    """NormalizingFlow.sample_to_host draws the prior chunk by chunk (nflow.row_chunks): the stream of
    one randn(n, d) (the reference's prior.sample, nflow.py:141) is unchanged, ragged tails included,
    and so is the generator state left behind."""
    import torch
    from probaforms_amd.models.nflow import row_chunks
    for d, rows, n in ((5, 16, 1007), (64, 4096, 10000), (2, 96, 1000), (1, 16, 100), (1, 16, 97), (3, 32, 65),
                       (7, 48, 48 * 3 + 15), (16, 16, 16), (2, 16, 5)):
        ch = row_chunks(n, rows)
        assert sum(m for _, m in ch) == n and all(lo == sum(m for _, m in ch[:i]) for i, (lo, _) in enumerate(ch))
        assert all(m <= rows + 15 for _, m in ch)
        torch.manual_seed(0)
        a = torch.randn(n, d); a_next = torch.randn(4)
        torch.manual_seed(0)
        b = torch.cat([torch.randn(m, d) for _, m in ch]); b_next = torch.randn(4)
        assert torch.equal(a, b) and torch.equal(a_next, b_next), (d, rows, n)


def test_pipelined_rows_policy():
    import torch
    from probaforms_amd.models import RealNVPLayer
    from probaforms_amd.models.nflow import NormalizingFlow, StandardNormalPrior
    layer = lambda d: [RealNVPLayer(d, 0, torch.arange(d) % 2, (4,), "tanh")]
    nf = NormalizingFlow(layer(64), StandardNormalPrior(64, "cpu"))
    rows = nf.pipelined_rows(16_000_000)
    # d = 64: the row floor decides (a chunk must still fill the chip); d = 2: the byte size does
    assert rows % 16 == 0 and rows == nf.PIPELINE_MIN_ROWS
    nf_small = NormalizingFlow(layer(2), StandardNormalPrior(2, "cpu"))
    assert nf_small.pipelined_rows(10 ** 8) * 2 * 4 == nf_small.PIPELINE_CHUNK_BYTES
    assert nf.pipelined_rows(2 * rows) == 0 and nf.pipelined_rows(2 * rows + 1) == rows
    nf2 = NormalizingFlow(layer(64), object())                # custom prior: one-shot path only
    assert nf2.pipelined_rows(10 ** 9) == 0


def test_bench_traffic_is_gated_on_the_kernel_source_hash(tmp_path, monkeypatch):
    """bench.py reports roofline.traffic from a committed PMC profile only while that profile was taken with exactly
    the kernel sources in the tree and names the kernel; otherwise null (the number must not go stale)"""
    import json
    import bench
    h = bench.csrc_hash()
    assert len(h) == 16 and h == bench.csrc_hash()
    f = tmp_path / "traffic.json"
    monkeypatch.setattr(bench, "TRAFFIC_FILE", str(f))
    assert bench.pmc_traffic("k_mfma_train") is None                                    # no file
    f.write_text(json.dumps({"csrc_hash": h, "kernels": {"k_pack_weights": {"hbm_bytes_per_launch": 1.0},
                                                         "k_mfma_train<2, 1, 4, 1, 0>": {"hbm_bytes_per_launch": 185e6}}}))
    assert bench.pmc_traffic("k_mfma_train") == 185e6
    assert bench.pmc_traffic("k_generic_train") is None                                 # kernel not in the profile
    f.write_text(json.dumps({"csrc_hash": "0" * 16, "kernels": {"k_mfma_train<2, 1, 4, 1, 0>": {"hbm_bytes_per_launch": 185e6}}}))
    assert bench.pmc_traffic("k_mfma_train") is None                                    # sources changed since the profile


def test_shape_struct_matches_the_c_header():
    """the ctypes mirror of rnvp_shape has the header's size and field order (compiled with gcc as C)"""
    import ctypes, os, subprocess, tempfile
    from conftest import ROOT
    from probaforms_amd import _hip
    src = ('#include <stdio.h>\n#include <stddef.h>\n#include "include/rnvp_hip.h"\n#include "include/cvae_hip.h"\n'
           'int main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu", sizeof(rnvp_shape), offsetof(rnvp_shape, hidden), '
           'offsetof(rnvp_shape, alt_masks), offsetof(rnvp_shape, precision), offsetof(rnvp_shape, small_calls), '
           'offsetof(rnvp_shape, family), sizeof(cvae_shape), offsetof(cvae_shape, family));return 0;}\n')
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "s.c"); exe = os.path.join(td, "s")
        open(c, "w").write(src)
        subprocess.check_call(["gcc", "-std=c99", "-I", ROOT, c, "-o", exe])
        size, o_hidden, o_alt, o_prec, o_small, o_fam, csize, co_fam = map(int, subprocess.check_output([exe]).split())
    S = _hip.RnvpShape
    assert (size, o_hidden, o_alt, o_prec, o_small, o_fam) == (ctypes.sizeof(S), S.hidden.offset, S.alt_masks.offset,
                                                               S.precision.offset, S.small_calls.offset, S.family.offset)
    assert (csize, co_fam) == (ctypes.sizeof(_hip.CvaeShape), _hip.CvaeShape.family.offset)
    fam = ('#include <stdio.h>\n#include "include/rnvp_hip.h"\nint main(void){printf("%d %d %d %d %d", RNVP_FAMILY_AUTO, RNVP_FAMILY_VALU, '
           'RNVP_FAMILY_LMM, RNVP_FAMILY_LMM16, RNVP_FAMILY_LMM64);return 0;}\n')
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "f.c"); exe = os.path.join(td, "f")
        open(c, "w").write(fam)
        subprocess.check_call(["gcc", "-std=c99", "-I", ROOT, c, "-o", exe])
        values = list(map(int, subprocess.check_output([exe]).split()))
    assert _hip.FAMILIES == dict(zip(["auto", "valu", "lmm", "lmm16", "lmm64"], values)) == {"auto": 0, "valu": 1, "lmm": 2, "lmm16": 3, "lmm64": 4}
    assert _hip.SMALL_CALLS == {"invariant": 0, "latency": 1}
    assert _hip.PRECISIONS == {"auto": 0, "f32": 1, "bx3": 2}


def test_kernel_selection_policy_is_host_logic():
    """which kernels serve a shape is decided on the host and needs no GPU: the resident (one launch per epoch) fit for
    the reference's default network, the CVAE families, invalid shapes refused"""
    from probaforms_amd import _hip
    R = _hip.RnvpShape.make
    default = R(8, 2, 1, (10,), "tanh", alt_masks=1)                  # RealNVP() defaults on 2-d data with one condition
    assert _hip.fit_epoch_resident(default, 32) and _hip.fit_epoch_resident(default, 1) and _hip.fit_epoch_resident(default, 64)
    assert not _hip.fit_epoch_resident(default, 129)                  # more than 8 waves of 16 rows
    assert not _hip.fit_epoch_resident(R(8, 2, 1, (10,), "tanh", family="valu"), 32)
    assert not _hip.fit_epoch_resident(R(8, 16, 4, (128,), "tanh", alt_masks=1), 32)      # C2: 75 k parameters do not fit LDS
    assert _hip.fit_epoch_resident(R(8, 2, 1, (10, 20, 15), "tanh"), 32)                  # the docstring network: two tiles per hidden layer
    assert not _hip.fit_epoch_resident(R(8, 2, 1, (10, 33), "tanh"), 32)
    assert _hip.fit_epoch_resident(R(8, 2, 1, (10, 10), "tanh"), 32) and _hip.fit_epoch_resident(R(8, 2, 1, (10, 16, 15), "relu"), 32)
    assert not _hip.fit_epoch_resident(R(8, 2, 1, (4, 4, 4, 4), "tanh"), 32)
    assert not _hip.fit_epoch_resident(R(8, 2, 1, (33,), "tanh"), 32)
    assert _hip.fit_epoch_resident(R(8, 2, 1, (32,), "relu"), 32) and not _hip.fit_epoch_resident(R(4, 16, 4, (32,), "relu"), 32)
    assert not _hip.fit_epoch_resident(R(17, 2, 1, (10,), "tanh"), 32) and _hip.fit_epoch_resident(R(16, 2, 1, (10,), "tanh"), 32)
    assert not _hip.fit_epoch_resident(R(2, 17, 0, (10,), "tanh"), 32) and not _hip.fit_epoch_resident(R(2, 16, 16, (10,), "tanh"), 32)
    Cv = _hip.CvaeShape.make
    assert _hip.cvae_kernel_path(Cv(16, 4, 2, (128,), "tanh")) == _hip.PATH_MFMA
    assert _hip.cvae_kernel_path(Cv(16, 4, 2, (128,), "tanh", family="lmm")) == _hip.PATH_LMM
    assert _hip.cvae_kernel_path(Cv(16, 4, 2, (128,), "tanh", family="generic")) == _hip.PATH_GENERIC
    assert _hip.cvae_kernel_path(Cv(4, 2, 3, (7, 9), "relu")) == _hip.PATH_LMM
    assert _hip.cvae_kernel_path(Cv(10, 5, 10, (700, 700), "relu")) == _hip.PATH_GENERIC
    bad = Cv(4, 2, 3, (7, 9), "relu"); bad.family = 7
    assert _hip.cvae_kernel_path(bad) not in (_hip.PATH_MFMA, _hip.PATH_LMM, _hip.PATH_GENERIC)       # RNVP_EINVAL


def test_permutation_prefetcher_host_only_keeps_what_was_submitted_and_close_drops_the_rest():
    """ADVICE round 4 (low / medium): host_only() must not throw away futures start() already submitted, and close() leaves nothing
    behind; the permutations are the serial DataLoader's whichever path produced them"""
    n, epochs = 70_000, 5
    torch.manual_seed(17)
    want = [_engine.loader_permutation(n) for _ in range(epochs)]
    torch.manual_seed(17)
    pf = _engine.PermutationPrefetcher(n, epochs, workers=2, lookahead=1, device=None).start()
    submitted = dict(pf.futs)
    assert submitted and pf.dev_epochs == 0
    pf.host_only()
    assert all(pf.futs.get(e) is f for e, f in submitted.items())            # the same future objects: nothing resubmitted
    got = [pf.get(e) for e in range(epochs)]
    assert all(torch.equal(a, b) for a, b in zip(got, want))
    pf.close()
    assert not pf.futs and not pf._early and pf._ws is None


def test_library_write_counts_die_with_their_storage():
    """the per-storage count of the library's raw-pointer writes must not outlive the storage (a later
    allocation at the same address would inherit it: a false "parameters modified")"""
    import gc
    t = torch.zeros(1024)
    key = t.untyped_storage().data_ptr()
    _engine.note_param_write(t); _engine.note_param_write(t)
    assert _engine._STORAGE_WRITES[key] == 2
    p = torch.nn.Parameter(t)
    assert _engine.param_state([p])[0][2] == 2
    del p, t
    gc.collect()
    assert key not in _engine._STORAGE_WRITES and key not in _engine._STORAGE_WATCHED


def test_bench_batch_geometry_of_both_regimes():
    """bench.py's two batch regimes: the per-rank share of every global batch tiles the rank's rows exactly"""
    rows, world = 1_000_000, 8
    for gbatch, rank_batch, nb in ((65_536, 8_192, 123), (65_536 * world, 65_536, 16)):
        gb = _engine.batch_bounds(rows * world, gbatch)
        assert len(gb) == nb
        for rank in range(world):
            shares = [_engine.shard_bounds(s, e, rank, world) for (s, e) in gb]
            assert shares[0][1] - shares[0][0] == rank_batch
            assert sum(hi - lo for lo, hi in shares) == rows
