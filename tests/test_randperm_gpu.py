"""The reference's epoch shuffle on the device (rnvp_randperm_torch_cpu, csrc/rnvp_randperm.hip) against torch.randperm itself:
/root/reference/probaforms/models/realnvp.py:235 DataLoader(shuffle=True) -> RandomSampler -> torch.randperm(n, generator=g)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _device_randperm(n, seed):
    from probaforms_amd import _hip
    from probaforms_amd.models.nflow import HostStreamOnDevice
    g = torch.Generator(); g.manual_seed(seed)
    st, mt = HostStreamOnDevice._unpack(g)
    mtd = torch.from_numpy(mt.copy()).cuda()
    out = torch.empty(n, dtype=torch.int64, device="cuda")
    ws = torch.empty(_hip.randperm_workspace_bytes(n), dtype=torch.uint8, device="cuda")
    _hip.randperm_torch_cpu(mtd, n, out, ws)
    HostStreamOnDevice._pack(g, st, mtd.cpu().numpy())
    return out.cpu(), g


@pytest.mark.parametrize("n", [1, 2, 3, 7, 100, 623, 624, 625, 1000, 2049, 4097, 65536, 100003, 638976, 638977, 1000000, 3000001])
def test_device_randperm_is_torch_randperm(n):
    for seed in (0, 12345, 2 ** 31 + 5, 2 ** 40 + 17, 2 ** 63 - 1):
        if n > 200000 and seed not in (12345, 2 ** 63 - 1):
            continue
        ref_g = torch.Generator(); ref_g.manual_seed(seed)
        ref = torch.randperm(n, generator=ref_g)
        got, g = _device_randperm(n, seed)
        assert torch.equal(got, ref), (n, seed)
        # the generator ends where torch left its own: whatever is drawn next agrees
        assert torch.equal(torch.randperm(17, generator=g), torch.randperm(17, generator=ref_g))


def test_device_randperm_repeats_itself_and_refuses_what_torch_does_otherwise():
    from probaforms_amd import _hip
    a, _ = _device_randperm(300001, 7)
    b, _ = _device_randperm(300001, 7)
    assert torch.equal(a, b) and sorted(a.tolist()) == list(range(300001))
    mt = torch.zeros(625, dtype=torch.int32, device="cuda")
    with pytest.raises(Exception):
        _hip.randperm_torch_cpu(mt, 2 ** 32 // 20, torch.empty(4, dtype=torch.int64, device="cuda"), torch.empty(1024, dtype=torch.uint8, device="cuda"))


def test_fit_with_device_drawn_first_permutations_equals_the_host_shuffle(monkeypatch):
    """RealNVP.fit on 70 000 rows: the first two epochs' permutations come from the device (PermutationPrefetcher), the third from the
    worker threads; losses, parameters and the global generator's final state equal those of a fit on host permutations only"""
    from probaforms_amd import _engine
    from probaforms_amd.models import RealNVP
    rng = np.random.default_rng(0)
    n = 70000
    X = rng.standard_normal((n, 5)).astype(np.float32); C = rng.standard_normal((n, 3)).astype(np.float32)
    assert _engine.DeviceShuffle.usable("cuda:0")
    out = {}
    for mode in ("device", "host"):
        monkeypatch.setenv("RNVP_HOST_SHUFFLE_ON_DEVICE", "1" if mode == "device" else "0")
        torch.manual_seed(11)
        m = RealNVP(n_layers=4, hidden=(16,), batch_size=8192, n_epochs=3, lr=1e-3)
        m.fit(X, C)
        out[mode] = (torch.stack([l.reshape(()) for l in m.loss_history]).cpu(), torch.cat([p.detach().reshape(-1).cpu() for p in m.nf.parameters()]),
                     torch.get_rng_state().clone())
    for a, b in zip(out["device"], out["host"]):
        assert torch.equal(a, b)
    monkeypatch.setenv("RNVP_HOST_SHUFFLE_ON_DEVICE", "1")
    p = _engine.PermutationPrefetcher(n, 3, device="cuda:0")
    assert p.dev_epochs == (3 if _engine.effective_cpus() < 4 else 2)
    assert p.get(0).is_cuda and torch.equal(p.get(1).cpu(), _engine.permutation_from_seed(n, p.seeds[1]))
    p.close()
    monkeypatch.setenv("RNVP_HOST_SHUFFLE_ON_DEVICE", "0")
    assert _engine.PermutationPrefetcher(n, 3, device="cuda:0").dev_epochs == 0
