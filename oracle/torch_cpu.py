"""Eager-PyTorch CPU restatement of the reference's RealNVP fit / sample loop -- TEST / MEASUREMENT INFRASTRUCTURE ONLY.

This is the "same-box CPU rate" of SURVEY.md 8(d)(ii) / BASELINE.md section 3: the reference's Python files never travel
to the GPU box, so bench.py's `cpu_baseline` leg times THIS loop there -- the same sequence of torch CPU ops the
reference issues, written from the published semantics, nothing copied:
  per coupling layer (/root/reference/probaforms/models/realnvp.py:91-101,120-129): cat([x*mask, c]) -> Linear -> Tanh ->
  Linear for the t net and the s net, exp, the masked affine update, sum of the log-det terms;
  flow (nflow.py:109-117,141-143): layers in order + prior term + batch mean; sampling: prior draw, layers reversed;
  fit (realnvp.py:235-254): a shuffled pass over the rows in batches; loss = -log_prob; zero_grad, backward (autograd),
  torch.optim.Adam.step; the loss read back per step.
To run at the reference's speed the loop also keeps the reference's inefficiencies: the masks are int64 tensors (so
every `x * mask` is a mixed-dtype elementwise op), the prior is a torch.distributions.MultivariateNormal object (a
triangular solve per log_prob call) and batches come out of torch's DataLoader(TensorDataset) row by row.  With float
masks, the closed-form prior and index slicing the same loop is ~4.6x faster on 8 cores.
tests/golden/validate_torch_cpu.py (build container only: it imports the reference) checks outputs and rows/s against the
reference itself.  Only bench.py's cpu_baseline leg and that script import this file; nothing under probaforms_amd/ does.
"""
import time

import torch
import torch.nn as nn


def _net(n_in, n_out, hidden, activation):
    mods, w = [], n_in
    for h in hidden:
        mods += [nn.Linear(w, h), nn.Tanh() if activation == "tanh" else nn.ReLU()]
        w = h
    mods.append(nn.Linear(w, n_out))
    return nn.Sequential(*mods)


class EagerFlow(nn.Module):
    def __init__(self, L, d, c, hidden, activation="tanh"):
        super().__init__()
        self.d, self.c = d, c
        self.masks = [(torch.arange(d) + i) % 2 for i in range(L)]                 # int64, as realnvp.py:199
        self.prior = torch.distributions.MultivariateNormal(torch.zeros(d), torch.eye(d))
        self.nets_t = nn.ModuleList([_net(d + c, d, hidden, activation) for _ in range(L)])
        self.nets_s = nn.ModuleList([_net(d + c, d, hidden, activation) for _ in range(L)])

    def load_flat(self, flat):
        """flat parameter vector in the reference's nf.parameters() order: per layer net t then net s"""
        off = 0
        with torch.no_grad():
            for t, s in zip(self.nets_t, self.nets_s):
                for p in list(t.parameters()) + list(s.parameters()):
                    n = p.numel()
                    p.copy_(torch.as_tensor(flat[off:off + n]).view_as(p)); off += n
        assert off == len(flat)

    def log_prob_rows(self, X, C):
        x, ld = X, torch.zeros(X.shape[0])
        for m, nt, ns in zip(self.masks, self.nets_t, self.nets_s):
            xc = torch.cat([x * m, C], dim=1) if C is not None else x * m
            T, S = nt(xc), ns(xc)
            x = (x * torch.exp(S) + T) * (1 - m) + x * m
            ld = ld + (S * (1 - m)).sum(dim=-1)
        return ld + self.prior.log_prob(x), x

    def inverse_rows(self, Z, C):
        """x = g(z, c): the layers' inverses, last layer first (nflow.py:142-143, realnvp.py:120-128)"""
        x = Z
        for m, nt, ns in reversed(list(zip(self.masks, self.nets_t, self.nets_s))):
            xc = torch.cat([x * m, C], dim=1) if C is not None else x * m
            T, S = nt(xc), ns(xc)
            x = ((x - T) * torch.exp(-S)) * (1 - m) + x * m
        return x

    def sample(self, C, n=None):
        n = len(C) if C is not None else n
        return self.inverse_rows(self.prior.sample((n,)), C)


def fit_epoch(flow, opt, X, C, batch_size):
    """one epoch through torch's own DataLoader(TensorDataset, shuffle=True), as realnvp.py:235-240 builds it: the
    per-row dataset indexing and collation of a 65 536-row batch is a large part of the reference's step time"""
    from torch.utils.data import DataLoader, TensorDataset
    data = TensorDataset(X, C) if C is not None else TensorDataset(X)
    hist = []
    for batch in DataLoader(data, batch_size=batch_size, shuffle=True):
        xb, cb = (batch[0], batch[1]) if C is not None else (batch[0], None)
        loss = -flow.log_prob_rows(xb, cb)[0].mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        hist.append(loss.detach().cpu())
    return hist


def timed_fit_and_sample(L, d, c, hidden, Xn, Cn, batch_size, threads, lr=1e-3, flat=None):
    """one warm-up batch, then one timed epoch over Xn and one timed sampling of len(Xn) rows; returns a dict"""
    torch.set_num_threads(int(threads))
    X = torch.from_numpy(Xn).to(torch.float32); C = None if Cn is None else torch.from_numpy(Cn).to(torch.float32)
    torch.manual_seed(0)
    flow = EagerFlow(L, d, c, hidden)
    if flat is not None:
        flow.load_flat(flat)
    opt = torch.optim.Adam(flow.parameters(), lr=lr)
    fit_epoch(flow, opt, X[:batch_size], None if C is None else C[:batch_size], batch_size)          # warm-up
    t0 = time.perf_counter()
    hist = fit_epoch(flow, opt, X, C, batch_size)
    t_fit = time.perf_counter() - t0
    with torch.no_grad():
        flow.sample(None if C is None else C[:1024], 1024)                                               # warm-up
    t0 = time.perf_counter()
    xs = flow.sample(C, len(X))           # graph recorded, as in the reference (no no_grad there: realnvp.py:279-282)
    t_s = time.perf_counter() - t0
    n = len(X)
    return dict(fit_rows_per_s=n / t_fit, sample_rows_per_s=n / t_s, combined_rows_per_s=2 * n / (t_fit + t_s),
                t_fit=t_fit, t_sample=t_s, rows=n, threads=int(threads), final_loss=float(hist[-1]),
                finite=bool(torch.isfinite(xs).all()))
