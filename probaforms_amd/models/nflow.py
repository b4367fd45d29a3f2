"""Flow container and invertible-layer protocol, MI355X build.

Mirrors `probaforms.models.nflow` (/root/reference/probaforms/models/nflow.py:15-145):
``InvertibleLayer`` is the protocol (``f`` forward + log-det, ``g`` inverse) and
``NormalizingFlow`` chains layers under a prior.  Here a flow made of RealNVP coupling
layers is not evaluated layer by layer: the whole stack runs as one fused HIP kernel
(probaforms_amd/csrc) through ``FlowEngine``.
"""
import math

import torch
import torch.nn as nn

from .._engine import FlowEngine, default_device, require_hip

DEVICE = default_device()


class InvertibleLayer(nn.Module):
    """Protocol of one invertible transform (nflow.py:15-67).

    f(X, C) -> (X_new [B, var_size], log_det [B]);   g(X, C) -> X_new [B, var_size].
    """

    def __init__(self, var_size):
        super().__init__()
        self.var_size = var_size

    def f(self, X, C):
        return None

    def g(self, X, C):
        return None


class StandardNormalPrior:
    """N(0, I) over var_size dimensions: what the reference builds as
    ``MultivariateNormal(zeros(d), eye(d))`` (realnvp.py:189-191).  ``log_prob`` is the closed
    form -0.5 (d ln 2pi + |z|^2) and ``sample`` is ``randn`` on the global CPU generator -- both
    bit-equal to the reference's distribution object (tests/golden/prior.npz) --
    which lets the kernels fold the prior term into the forward pass."""

    def __init__(self, var_size, device, host_rng=True):
        self.var_size = int(var_size)
        self.device = torch.device(device)
        self.host_rng = host_rng
        self.loc = torch.zeros(var_size)
        self.covariance_matrix = torch.eye(var_size)

    def log_prob(self, z):
        return -0.5 * ((z * z).sum(-1) + self.var_size * math.log(2.0 * math.pi))

    def sample(self, sample_shape=()):
        shape = tuple(sample_shape) + (self.var_size,)
        if not self.host_rng:
            return torch.randn(shape, device=self.device)  # throughput option: not the reference's stream
        return torch.randn(shape).to(self.device)          # host generator: the reference's CPU stream


class NormalizingFlow(nn.Module):
    """Layers + prior (nflow.py:71-145).

    log_prob(X, C) -> 0-dim tensor: mean over the batch of [sum_l log_det_l + prior.log_prob(z)].
    sample(C)      -> [n, var_size]: prior draw pushed through the layers' inverses, last layer first.
    log_prob_samples(X, C) -> [n] per-row log-density (build-only addition).
    """

    def __init__(self, layers, prior):
        super().__init__()
        self.layers = nn.ModuleList(layers)
        self.prior = prior
        self._engine_obj = None

    # -- engine ------------------------------------------------------------------------------
    def engine(self):
        dev = None
        for p in self.parameters():
            dev = p.device
            break
        if dev is None or dev.type != "cuda":
            dev = DEVICE
        require_hip(dev)
        if self._engine_obj is None or self._engine_obj.device != torch.device(dev) \
                or len(self._engine_obj.layers) != len(self.layers):
            self._engine_obj = FlowEngine(list(self.layers), dev)
        return self._engine_obj

    def _fused_prior(self):
        return isinstance(self.prior, StandardNormalPrior)

    def _on_device(self, t, eng):
        if t is None:
            return None
        return torch.as_tensor(t, dtype=torch.float32).to(eng.device).contiguous()

    # -- reference API -------------------------------------------------------------------------
    def log_prob(self, X, C):
        eng = self.engine()
        X, C = self._on_device(X, eng), self._on_device(C, eng)
        if self._fused_prior():
            _, _, _, tot = eng.forward(X, C, want_z=False, want_logp=False, want_sum=True)
            return (tot / X.shape[0]).reshape(())
        z, ld, _, _ = eng.forward(X, C, want_z=True, want_logdet=True, want_logp=False)
        return (ld + self.prior.log_prob(z)).mean()

    def log_prob_samples(self, X, C=None):
        eng = self.engine()
        X, C = self._on_device(X, eng), self._on_device(C, eng)
        if self._fused_prior():
            return eng.forward(X, C, want_z=False, want_logp=True)[2]
        z, ld, _, _ = eng.forward(X, C, want_z=True, want_logdet=True, want_logp=False)
        return ld + self.prior.log_prob(z)

    def sample(self, C):
        eng = self.engine()
        if type(C) == type(1):            # python int only, as nflow.py:135 (np.int64 is not an int)
            n, C = C, None
        else:
            n = len(C)
            C = self._on_device(C, eng)
        z = self.prior.sample((n,))
        z = torch.as_tensor(z, dtype=torch.float32).to(eng.device).contiguous()
        return eng.inverse(z, C, out=z)
