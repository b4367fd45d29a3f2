#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE (hse-cs/probaforms @ 2024_10_08).

Run ONLY in the build container, where the reference is mounted read-only:

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference cannot travel to the GPU box, so its outputs are committed as
small fixtures (data: inputs + expected outputs).  Fixture groups follow
SURVEY.md 8(c): G1 init state_dict + masks, G2 forward per layer / log-prob,
G3 inverse per layer, G4 loss, gradients and Adam trajectories, G5 DataLoader
batch indices, G6 prior samples, G7 seeded end-to-end fit + sample, G8 ragged
batch, G9 relu / multi-hidden.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from cases import CASES, GRAD_STRIDE, inputs, numpy_params, param_count  # noqa: E402

from probaforms.models import RealNVP  # the reference  # noqa: E402
from probaforms.models.realnvp import RealNVPLayer  # noqa: E402
from probaforms.models.nflow import NormalizingFlow  # noqa: E402

assert "/root/reference" in sys.modules["probaforms"].__file__, "must import the reference"
torch.set_num_threads(1)


def build_model(name, seed=0, **kw):
    L, d, c, hidden, act, _ = CASES[name]
    m = RealNVP(n_layers=L, hidden=hidden, activation=act, **kw)
    torch.manual_seed(seed)
    X = np.zeros((2, d), np.float32)
    C = np.zeros((2, c), np.float32) if c > 0 else None
    m._model_init(X, C)          # realnvp.py:180-207 (consumes the global CPU generator)
    return m


def flat(nf):
    return torch.cat([p.detach().reshape(-1) for p in nf.parameters()]).numpy().copy()


def flat_grad(nf):
    return torch.cat([p.grad.detach().reshape(-1) for p in nf.parameters()]).numpy().copy()


def load_flat(nf, vec):
    off = 0
    with torch.no_grad():
        for p in nf.parameters():
            n = p.numel()
            p.copy_(torch.from_numpy(vec[off:off + n]).view_as(p))
            off += n
    assert off == vec.size


def per_sample_logp(nf, X, C):
    """replay of nflow.py:109-115 without the .mean()"""
    ll = None
    outs, lds = [], []
    for layer in nf.layers:
        X, ch = layer.f(X, C)
        outs.append(X.detach().numpy().copy()); lds.append(ch.detach().numpy().copy())
        ll = ch if ll is None else ll + ch
    ll = ll + nf.prior.log_prob(X)
    return outs, lds, X.detach().numpy().copy(), ll.detach().numpy().copy()


def make_case(name):
    L, d, c, hidden, act, wsrc = CASES[name]
    out = {}
    m = build_model(name, seed=0)
    nf = m.nf
    masks = np.stack([l.mask.numpy() for l in nf.layers])
    assert masks.dtype == np.int64
    out["masks"] = masks.astype(np.uint8)                                   # G1 (bit-exact)
    if wsrc == "torch":
        out["G1_params"] = flat(nf)                                          # G1 init order
        params = out["G1_params"]
        # spread the weights a little so that exp(s) is not ~1 everywhere
        params2 = (params * (1.0 if act == "relu" else 1.5)).astype(np.float32)
    else:
        params2 = numpy_params(name, scale=1.0)
        assert params2.size == param_count(L, d, c, hidden)
    load_flat(nf, params2)
    if wsrc == "torch":
        out["params"] = params2
    n = 64 if wsrc == "torch" else 32
    X, C, Z = inputs(name, n)
    Xt = torch.from_numpy(X); Ct = torch.from_numpy(C) if C is not None else None
    with torch.no_grad():
        outs, lds, z, lp = per_sample_logp(nf, Xt, Ct)
        out["G2_mean"] = np.float32(nf.log_prob(Xt, Ct).item())
    out["G2_layer_out"] = np.stack(outs); out["G2_layer_ld"] = np.stack(lds)
    out["G2_z"] = z; out["G2_logp"] = lp
    # G3 inverse, layer by layer in sampling order (nflow.py:142-143)
    with torch.no_grad():
        cur = torch.from_numpy(Z); gouts = []
        for layer in nf.layers[::-1]:
            cur = layer.g(cur, Ct); gouts.append(cur.numpy().copy())
    out["G3_layer_out"] = np.stack(gouts); out["G3_x"] = gouts[-1]
    with torch.no_grad():
        back = torch.from_numpy(z)
        for layer in nf.layers[::-1]:
            back = layer.g(back, Ct)
    out["G3_roundtrip_maxerr"] = np.float32((back - Xt).abs().max().item())
    # G4 / G8: loss + gradient for a full batch and for a ragged 8-row batch
    for tag, rows in (("G4", n), ("G8", 8)):
        nf.zero_grad()
        loss = -nf.log_prob(Xt[:rows], None if Ct is None else Ct[:rows])   # realnvp.py:246
        loss.backward()
        g = flat_grad(nf)
        out[tag + "_loss"] = np.float32(loss.item())
        if wsrc == "torch":
            out[tag + "_grad"] = g
        else:   # large shapes: every GRAD_STRIDE-th entry + norms keep the fixture small
            out[tag + "_grad_sub"] = g[::GRAD_STRIDE].copy()
            out[tag + "_grad_l2"] = np.float64(np.sqrt((g.astype(np.float64) ** 2).sum()))
            out[tag + "_grad_sum"] = np.float64(g.astype(np.float64).sum())
    # G4 Adam trajectories (3 steps) for weight_decay 0 and 0.2 (forecast.ipynb cell 23)
    if wsrc == "torch":
        for wd in (0.0, 0.2):
            load_flat(nf, params2)
            opt = torch.optim.Adam(nf.parameters(), lr=0.01, weight_decay=wd)  # realnvp.py:205-207
            traj_p, traj_m, traj_v, traj_loss = [], [], [], []
            for _step in range(3):
                loss = -nf.log_prob(Xt, Ct)
                opt.zero_grad(); loss.backward(); opt.step()
                traj_loss.append(loss.item()); traj_p.append(flat(nf))
                traj_m.append(torch.cat([opt.state[p]["exp_avg"].reshape(-1) for p in nf.parameters()]).numpy().copy())
                traj_v.append(torch.cat([opt.state[p]["exp_avg_sq"].reshape(-1) for p in nf.parameters()]).numpy().copy())
            k = "G4_adam_wd%g" % wd
            out[k + "_p"] = np.stack(traj_p); out[k + "_m"] = np.stack(traj_m)
            out[k + "_v"] = np.stack(traj_v); out[k + "_loss"] = np.array(traj_loss, np.float32)
    np.savez_compressed(os.path.join(HERE, "case_%s.npz" % name), **out)
    print(name, {k: getattr(v, "shape", None) for k, v in out.items() if k.startswith("G2")},
          "roundtrip", out["G3_roundtrip_maxerr"])


def make_loader_indices():
    """G5: batch index sequences of the real DataLoader (realnvp.py:237)."""
    from torch.utils.data import DataLoader, TensorDataset
    out = {}
    for seed in (0, 7):
        for n in (100, 103, 1000):
            torch.manual_seed(seed)
            ds = TensorDataset(torch.arange(n))
            epochs = []
            for _e in range(3):
                idx = torch.cat([b[0] for b in DataLoader(ds, batch_size=32, shuffle=True)])
                epochs.append(idx.numpy())
            out["seed%d_n%d" % (seed, n)] = np.stack(epochs).astype(np.int64)
            # a draw AFTER the three epochs pins how much of the global stream was consumed
            out["seed%d_n%d_next" % (seed, n)] = torch.randn(4).numpy()
    np.savez_compressed(os.path.join(HERE, "loader_indices.npz"), **out)
    print("G5", list(out)[:4])


def make_prior_samples():
    """G6: prior.sample((n,)) of MultivariateNormal(0, I) (nflow.py:141) per (seed, n, d)."""
    out = {}
    for seed, n, d in ((0, 7, 2), (3, 33, 5), (0, 16, 16), (1, 5, 1)):
        torch.manual_seed(seed)
        prior = torch.distributions.MultivariateNormal(torch.zeros(d), torch.eye(d))
        out["seed%d_n%d_d%d" % (seed, n, d)] = prior.sample((n,)).numpy()
        zz = torch.randn(6, d)
        out["logprob_in_d%d" % d] = zz.numpy(); out["logprob_out_d%d" % d] = prior.log_prob(zz).numpy()
    np.savez_compressed(os.path.join(HERE, "prior.npz"), **out)
    print("G6", list(out))


def make_moons_fit():
    """G7: seeded end-to-end fit + sample on make_moons (README.md:51-59)."""
    from sklearn.datasets import make_moons
    X, y = make_moons(n_samples=1000, noise=0.1, random_state=0)
    C = y.reshape(-1, 1)
    out = {"X": X, "C": C.astype(np.float64)}
    for L in (8, 4):
        torch.manual_seed(0)
        m = RealNVP(n_layers=L, lr=0.01, n_epochs=2)
        m.fit(X, C)
        out["L%d_loss_history" % L] = np.array([float(v) for v in m.loss_history], np.float32)
        out["L%d_params_after" % L] = flat(m.nf)
        st = torch.get_rng_state()
        out["L%d_z" % L] = torch.randn(1000, 2).numpy()
        torch.set_rng_state(st)
        out["L%d_sample" % L] = m.sample(C)
        # second fit continues training (warm start, realnvp.py:189-193)
        m.fit(X[:64], C[:64])
        out["L%d_loss_history_len_after_refit" % L] = np.int64(len(m.loss_history))
    np.savez_compressed(os.path.join(HERE, "moons_fit.npz"), **out)
    print("G7", {k: getattr(v, "shape", None) for k, v in out.items()})


def make_c2_fit():
    """G7 on the C2 architecture at the reference's DEFAULT batch size (realnvp.py:161, 237-254): d=16, c=4, L=8,
    hidden=(128,), n=256 rows, batch_size=32, 2 epochs = 16 steps.  Pins, end to end against the reference, the kernels
    that serve small batches of a wide flow (three or more hidden tiles per net)."""
    rng = np.random.default_rng(42)
    X = rng.normal(size=(256, 16)); C = rng.normal(size=(256, 4))
    out = {"X": X, "C": C}
    torch.manual_seed(0)
    m = RealNVP(n_layers=8, hidden=(128,), lr=0.001, n_epochs=2)          # batch_size = 32: the default
    m.fit(X, C)
    out["loss_history"] = np.array([float(v) for v in m.loss_history], np.float32)
    out["params_after"] = flat(m.nf)
    out["sample"] = m.sample(C)
    with torch.no_grad():
        out["logp_after"] = per_sample_logp(m.nf, torch.tensor(X, dtype=torch.float32), torch.tensor(C, dtype=torch.float32))[3]
    np.savez_compressed(os.path.join(HERE, "c2_fit.npz"), **out)
    print("G7/c2", {k: getattr(v, "shape", None) for k, v in out.items()}, out["loss_history"][:4], out["loss_history"][-2:])


if __name__ == "__main__":
    only = sys.argv[1:]
    for name in CASES:
        if not only or name in only:
            make_case(name)
    if not only:
        make_loader_indices(); make_prior_samples(); make_moons_fit()
    if not only or "c2_fit" in only:
        make_c2_fit()
