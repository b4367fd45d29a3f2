#!/usr/bin/env python3
"""Golden fixtures for the CVAE row, produced by running the REFERENCE.

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_cvae.py

Per case: seeded init (Encoder then Decoder, cvae.py:164-175), X, C, the eps the reference draws
inside sample_z (cvae.py:187, re-derived from the saved generator state), loss and gradients of
compute_loss, a 3-step Adam trajectory, encoder/decoder outputs, and one seeded end-to-end
fit + sample."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
from probaforms.models import CVAE  # the reference

assert "/root/reference" in sys.modules["probaforms"].__file__
torch.set_num_threads(1)

CASES = {   # name: (d, c, latent, hidden, activation, KL_weight)
    "default": (5, 3, 2, (10,), "tanh", 0.001),          # tests/test_models.py shape, CVAE defaults
    "nocond": (5, 0, 2, (10,), "tanh", 0.001),
    "c5": (16, 4, 2, (128,), "tanh", 0.001),               # BASELINE.json configs[4] shape
    "relu_mh": (4, 2, 3, (7, 9), "relu", 0.5),
}


def state_flat(m):
    """oracle flat order: trunk, [W_mu, W_ls, b_mu, b_ls], decoder"""
    e, d = m.encoder.state_dict(), m.decoder.state_dict()
    nh = len(m.hidden)
    parts = []
    for k in range(nh):
        parts += [e["model.%d.weight" % (2 * k)], e["model.%d.bias" % (2 * k)]]
    parts += [e["mu.weight"], e["log_sigma.weight"], e["mu.bias"], e["log_sigma.bias"]]
    for k in range(nh + 1):
        parts += [d["model.%d.weight" % (2 * k)], d["model.%d.bias" % (2 * k)]]
    return torch.cat([p.detach().reshape(-1) for p in parts]).numpy().copy()


def grad_flat(m):
    e, d = dict(m.encoder.named_parameters()), dict(m.decoder.named_parameters())
    nh = len(m.hidden)
    parts = []
    for k in range(nh):
        parts += [e["model.%d.weight" % (2 * k)], e["model.%d.bias" % (2 * k)]]
    parts += [e["mu.weight"], e["log_sigma.weight"], e["mu.bias"], e["log_sigma.bias"]]
    for k in range(nh + 1):
        parts += [d["model.%d.weight" % (2 * k)], d["model.%d.bias" % (2 * k)]]
    return torch.cat([p.grad.detach().reshape(-1) for p in parts]).numpy().copy()


def make_case(name):
    d, c, lat, hidden, act, klw = CASES[name]
    rng = np.random.default_rng(len(name) * 7 + d)
    n = 48
    X = rng.normal(size=(n, d)).astype(np.float32)
    C = rng.normal(size=(n, c)).astype(np.float32) if c else None
    out = {"X": X}
    if C is not None:
        out["C"] = C
    m = CVAE(latent_dim=lat, hidden=hidden, activation=act, KL_weight=klw, lr=0.01, weight_decay=0.0)
    torch.manual_seed(0)
    m._model_init(X, C)                                   # cvae.py:164-184, consumes the global generator
    m.encoder.to("cpu"); m.decoder.to("cpu")
    out["init_params"] = state_flat(m)
    Xt = torch.from_numpy(X); Ct = None if C is None else torch.from_numpy(C)
    with torch.no_grad():
        mu, ls = m.encoder(Xt, Ct)
    out["mu"] = mu.numpy(); out["log_sigma"] = ls.numpy()
    # loss + gradient with the eps the reference itself draws
    torch.manual_seed(11)
    st = torch.get_rng_state()
    eps = torch.randn(n, lat)
    torch.set_rng_state(st)
    loss = m.compute_loss(Xt, Ct)                          # cvae.py:195-203
    m.opt.zero_grad(); loss.backward()
    out["eps"] = eps.numpy(); out["loss"] = np.float32(loss.item()); out["grad"] = grad_flat(m)
    # decoder on given latent draws (cvae.py:284-290)
    Z = torch.randn(n, lat)
    with torch.no_grad():
        out["Z"] = Z.numpy(); out["decoded"] = m.decoder(Z, Ct).numpy()
    # 3 optimizer steps on the full batch, eps drawn by the reference each step
    torch.manual_seed(21)
    epss, ps, losses = [], [], []
    for _ in range(3):
        st = torch.get_rng_state(); epss.append(torch.randn(n, lat).numpy()); torch.set_rng_state(st)
        loss = m.compute_loss(Xt, Ct)
        m.opt.zero_grad(); loss.backward(); m.opt.step()
        losses.append(loss.item()); ps.append(state_flat(m))
    out["adam_eps"] = np.stack(epss); out["adam_p"] = np.stack(ps); out["adam_loss"] = np.array(losses, np.float32)
    # seeded end-to-end fit + sample through the public API (RNG: init, then per epoch shuffle draws,
    # per step eps, per epoch full-data eps; then torch.normal for sampling)
    m2 = CVAE(latent_dim=lat, hidden=hidden, activation=act, KL_weight=klw, lr=0.01, n_epochs=3, batch_size=16)
    import probaforms.models.cvae as refmod
    refmod.DEVICE = torch.device("cpu")
    torch.manual_seed(5)
    m2.fit(X, C)
    out["fit_loss_history"] = np.array([float(v) for v in m2.loss_history], np.float32)
    out["fit_params"] = state_flat(m2)
    out["fit_sample"] = m2.sample(C if C is not None else n)
    np.savez_compressed(os.path.join(HERE, "cvae_%s.npz" % name), **out)
    print(name, {k: getattr(v, "shape", None) for k, v in out.items()})


if __name__ == "__main__":
    for name in CASES:
        make_case(name)
