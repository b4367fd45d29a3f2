"""Conditional RealNVP, MI355X build.

Mirrors `probaforms.models.realnvp` (/root/reference/probaforms/models/realnvp.py): same
constructor arguments and defaults, same attributes (`nf`, `opt`, `prior`, `loss_history`, ...),
same `state_dict` keys and `[out, in]` weight layout, same RNG consumption (parameter init,
per-epoch shuffle, prior draws) -- but the coupling stack, its backward and Adam run as
hand-written HIP kernels (probaforms_amd/csrc, C ABI in include/rnvp_hip.h).
"""
import numpy as np
import torch
import torch.nn as nn

from .. import _hip
from .._engine import (FlatAdam, FlowFunction, InverseFunction, PermutationPrefetcher, broadcast_, default_device, dist_info, fit_epochs,
                       flatten_parameters, shard_bounds, is_flat, require_hip)
from .interfaces import GenModel
from .nflow import InvertibleLayer, NormalizingFlow, StandardNormalPrior

DEVICE = default_device()


def _to_device_f32(a, device):
    if torch.is_tensor(a):
        return a.to(device).to(torch.float32).contiguous()
    a = np.asarray(a)
    if a.ndim > 0 and not a.flags.c_contiguous:
        a = np.ascontiguousarray(a)              # (a 0-d input stays 0-d: len() then raises TypeError as in the reference)
    if a.dtype.kind not in "fiub" or a.dtype.itemsize > 8:      # (numpy's longdouble: a width torch.from_numpy does not take)
        a = a.astype(np.float64)
    return torch.from_numpy(a).to(device).to(torch.float32).contiguous()


def gen_network(n_inputs, n_outputs, hidden=(10,), activation='tanh'):
    """The s or t net: Linear(n_inputs, hidden[0]), act, ..., Linear(hidden[-1], n_outputs)
    (realnvp.py:19-43).  Built from stock ``nn.Linear`` modules in the reference's order so the
    default initialisation consumes the global generator identically; 'tanh' selects Tanh, any
    other string ReLU."""
    widths = [n_inputs] + list(hidden)
    net = nn.Sequential()
    for w_in, w_out in zip(widths[:-1], widths[1:]):
        net.append(nn.Linear(w_in, w_out))
        net.append(nn.Tanh() if activation == 'tanh' else nn.ReLU())
    net.append(nn.Linear(widths[-1], n_outputs))
    return net


class _LayerView:
    """what FlowFunction needs of an engine, for ONE layer evaluated on its own (shape with L = 1)"""

    def __init__(self, layer, shape, mask, params, device):
        self.shape, self.masks, self.params, self.flat = shape, mask, params, params
        self.param_list = list(layer.parameters())
        self.P = params.numel()
        self.device = device

    def workspace(self, op, rows):
        return RealNVPLayer._ws(self.shape, self.device, op, max(int(rows), 1))


class RealNVPLayer(InvertibleLayer):
    """One affine coupling layer (realnvp.py:47-129).

    With m = mask, u = [X*m || C]:  T = nn_t(u), S = nn_s(u),
      f:  X_new = (X*exp(S) + T)*(1-m) + X*m,   log_det = sum_j (1-m_j) S_j
      g:  X_new = ((X - T)*exp(-S))*(1-m) + X*m
    `mask` is a {0,1} tensor of length var_size kept as a plain attribute (not in state_dict).
    `f`/`g` run the HIP kernels on this layer alone; inside a NormalizingFlow the whole stack
    is fused instead.  `f` is differentiable like the reference's (one autograd node whose backward
    is the HIP backward, rnvp_backward with L = 1) whenever grad mode is on and a parameter, X or C
    requires grad; so is `g` (backward = rnvp_inverse_backward).  Under torch.no_grad() both return plain tensors.
    """

    def __init__(self, var_size, cond_size, mask, hidden=(10,), activation='tanh'):
        super().__init__(var_size=var_size)
        self.cond_size = cond_size
        self.hidden = tuple(hidden)
        self.activation = activation
        self.mask = mask            # stays a host tensor; the kernels take a uint8 copy
        self.nn_t = gen_network(var_size + cond_size, var_size, hidden, activation)   # order matters:
        self.nn_s = gen_network(var_size + cond_size, var_size, hidden, activation)   # t first, then s
        self._flat = None

    def _layer_params(self, device):
        plist = list(self.parameters())
        if is_flat(plist, self._flat) and self._flat.device == device:
            return self._flat
        # inside a fused flow the parameters already sit contiguously in the flow's buffer
        first = plist[0]
        if first.device == device and first.dtype == torch.float32:
            off, ok = first.data_ptr(), True
            for p in plist:
                ok = ok and p.data_ptr() == off and p.is_contiguous() and p.dtype == torch.float32
                off += p.numel() * 4
            if ok:
                n = sum(p.numel() for p in plist)
                base = first.data.view(-1)
                return torch.as_strided(base, (n,), (1,))
        self._flat = flatten_parameters(plist, device)
        return self._flat

    def _prep(self, X, C):
        dev = X.device if isinstance(X, torch.Tensor) and X.is_cuda else DEVICE
        require_hip(dev)
        X = torch.as_tensor(X, dtype=torch.float32).to(dev).contiguous()
        if self.cond_size > 0:
            if C is None:
                raise RuntimeError("mat1 and mat2 shapes cannot be multiplied: layer built with cond_size=%d needs C"
                                   % self.cond_size)
            C = torch.as_tensor(C, dtype=torch.float32).to(dev).contiguous()
        elif C is not None:
            raise RuntimeError("mat1 and mat2 shapes cannot be multiplied: layer built with cond_size=0 got C")
        host_mask = self.mask.detach().to("cpu").to(torch.uint8).reshape(-1)
        shape = _hip.RnvpShape.make(1, self.var_size, self.cond_size, self.hidden, self.activation,
                                    alt_masks=_hip.RnvpShape.classify_masks(host_mask.numpy()))
        mask = host_mask.to(dev).contiguous()
        P = _hip.param_count(shape)
        return dev, X, C, shape, mask, self._layer_params(dev)[:P]

    def f(self, X, C=None):
        if self._wants_graph(X, C):
            Xg = X if torch.is_tensor(X) else torch.as_tensor(X, dtype=torch.float32)
            dev, _, C, shape, mask, params = self._prep(Xg.detach(), C)           # (C keeps its graph: .to / .contiguous)
            view = _LayerView(self, shape, mask, params, dev)
            return FlowFunction.apply(view, Xg.to(dev, torch.float32).contiguous(), C, *view.param_list)
        dev, X, C, shape, mask, params = self._prep(X, C)
        n = X.shape[0]
        X_new = torch.empty_like(X)
        log_det = torch.empty(n, dtype=torch.float32, device=dev)
        _hip.forward_logprob(shape, params, mask, X, C, None, n, X_new, log_det, None, None, self._ws(shape, dev, 0, n))
        return X_new, log_det

    def _wants_graph(self, X, C):
        return torch.is_grad_enabled() and ((torch.is_tensor(X) and X.requires_grad) or (torch.is_tensor(C) and C.requires_grad)
                                            or any(p.requires_grad for p in self.parameters()))

    def g(self, X, C=None):
        if self._wants_graph(X, C):       # realnvp.py:120-129 is an autograd graph: one node, backward = rnvp_inverse_backward (L = 1)
            Xg = X if torch.is_tensor(X) else torch.as_tensor(X, dtype=torch.float32)
            dev, _, C, shape, mask, params = self._prep(Xg.detach(), C)
            view = _LayerView(self, shape, mask, params, dev)
            return InverseFunction.apply(view, Xg.to(dev, torch.float32).contiguous(), C, *view.param_list)
        dev, X, C, shape, mask, params = self._prep(X, C)
        X_new = torch.empty_like(X)
        _hip.inverse(shape, params, mask, X, C, X.shape[0], X_new, self._ws(shape, dev, 1, X.shape[0]))
        return X_new

    @staticmethod
    def _ws(shape, dev, op, n):
        nb = _hip.workspace_bytes(shape, op, n)
        return torch.empty(nb, dtype=torch.uint8, device=dev) if nb > 0 else None


class RealNVP(GenModel):
    """RealNVP normalizing flow with the reference's sklearn-style interface (realnvp.py:133-282).

    Parameters (identical names and defaults to the reference):
        n_layers=8, hidden=(10,), activation='tanh', batch_size=32, n_epochs=10, lr=1e-4,
        weight_decay=0, verbose=0
    ``fit(X, C=None)`` takes numpy arrays [n, d] / [n, c]; ``sample(C)`` takes [n, c] or a python
    int and returns a float32 numpy array [n, d].  A second ``fit`` continues training (same
    optimizer state) and keeps appending to ``loss_history`` (one 0-dim CPU tensor per batch).
    Under ``torch.distributed`` (one process per GPU) ``fit`` is data parallel.
    """

    def __init__(self, n_layers=8, hidden=(10,), activation='tanh',
                 batch_size=32, n_epochs=10, lr=0.0001, weight_decay=0, verbose=0, *, prior_rng='host',
                 precision=None, small_calls=None):
        super().__init__()
        # build-only, keyword-only: None / 'auto', 'f32' or 'bx3' -- see NormalizingFlow
        self.precision = precision
        # build-only, keyword-only: None / 'invariant' or 'latency' -- see NormalizingFlow
        self.small_calls = small_calls
        # build-only, keyword-only: 'host' draws the prior on the global CPU generator like the
        # reference (bit-identical stream, ~25 ms per million 16-d rows); 'device' draws on the GPU
        self.prior_rng = prior_rng
        self.n_layers = n_layers
        self.hidden = hidden
        self.activation = activation
        self.batch_size = batch_size
        self.n_epochs = n_epochs
        self.lr = lr
        self.weight_decay = weight_decay
        self.verbose = verbose

        self.prior = None
        self.nf = None
        self.opt = None

        self.loss_history = []

    def _model_init(self, X, C):
        """Lazy construction on first fit (realnvp.py:180-207): prior, layers with masks
        (arange(d) + i) % 2, flow, optimizer -- each only if still None (warm start)."""
        require_hip(DEVICE)
        var_size = X.shape[1]
        cond_size = C.shape[1] if C is not None else 0

        if self.prior is None:
            self.prior = StandardNormalPrior(var_size, DEVICE, host_rng=(self.prior_rng != 'device'))

        if self.nf is None:
            layers = [RealNVPLayer(var_size=var_size, cond_size=cond_size,
                                   mask=((torch.arange(var_size) + i) % 2),
                                   hidden=self.hidden, activation=self.activation)
                      for i in range(self.n_layers)]
            self.nf = NormalizingFlow(layers=layers, prior=self.prior, precision=self.precision,
                                      small_calls=self.small_calls)
            eng = self.nf.engine()                 # moves the parameters into one flat HIP buffer
            _, world = dist_info()
            if world > 1:
                broadcast_(eng.flat, src=0)        # identical replicas whatever the local seeds were
            self.opt = FlatAdam(eng.flat.numel(), eng.device, lr=self.lr, weight_decay=self.weight_decay)

    def fit(self, X, C=None):
        self._model_init(X, C)
        eng = self.nf.engine()
        # The epoch shuffles: the loader seeds are drawn here -- the point of the generator stream where the reference's
        # first DataLoader draws them (nothing between here and the batch loop consumes the generator) -- and, on a single
        # GPU, the first permutations start computing on worker threads while the data uploads.  (Data parallel: the
        # seeds are rank 0's, broadcast inside fit_epochs, so nothing is started before that.)
        perms = PermutationPrefetcher(len(X), self.n_epochs, device=eng.device)
        if dist_info()[1] == 1:
            perms.start()
        # numpy (any float dtype) -> float32 on the device, once (realnvp.py:226-228); the cast runs on
        # the device (same round-to-nearest as the host cast, without a host-side pass over the data)
        Xd = _to_device_f32(X, eng.device)
        Cd = None if C is None else _to_device_f32(C, eng.device)
        # a prior the user assigned before fit() is kept and trained against, as in the reference (realnvp.py:189-191):
        # its log_prob / gradient come from torch, the flow's backward from the HIP kernel (rnvp_loss_grad_zseed)
        prior = None if self.nf._fused_prior() else self.nf.prior
        eng._cond(Cd, Xd.shape[0])
        bar = None
        if self.verbose >= 1:
            from tqdm.auto import tqdm
            bar = tqdm(total=self.n_epochs, unit='epoch')

        def hook(epoch, losses):
            """progress text of realnvp.py:256-262, from the epoch's loss vector (read back once per epoch, one epoch
            behind the GPU): verbose == 1 shows the epoch's last batch; verbose >= 2 walks the batches i with
            i % display_delta == 0 exactly like the reference's inner loop, so the bar ends the epoch on the same text"""
            if bar is None:
                return
            bar.update(1)
            if self.verbose >= 2:
                display_delta = max(1, (len(X) // self.batch_size) // self.verbose)
                for i in range(0, losses.numel(), display_delta):
                    bar.set_description("loss: %.4f" % float(losses[i]))
            else:
                bar.set_description("loss: %.4f" % float(losses[-1]))

        fit_epochs(eng, self.opt, Xd, Cd, self.batch_size, self.n_epochs, self.loss_history, hook if bar is not None else None,
                   prior=prior, perms=perms)
        if bar is not None:
            bar.close()

    def sample(self, C=100, *, distributed=None):
        """``distributed`` (build-only, keyword-only) matters under torch.distributed only:
        None -- every rank draws all n rows (the reference's behaviour in each process);
        'shard' -- this rank draws only its contiguous share of the rows and returns that block;
        'gather' -- shares are drawn per rank, then all-gathered: every rank returns all n rows.
        With the host prior the n x d normal draw is made in full on every rank (same generator state ->
        same stream; torch's CPU generator cannot skip ahead) and sliced, so the union of the shares is the
        single-process sample but the draw itself does not speed up with more ranks.  With
        prior_rng='device' rank 0's seed is broadcast and each rank draws only its own rows of the
        counter-based stream inside the inverse kernel: the shares concatenate to the single-process device
        draw bit for bit, whatever the ranks' generator states -- use it for large sharded draws (C4)."""
        n = C if type(C) == type(1) else len(C)
        rank, world = dist_info()
        if distributed is not None and world > 1:
            return self._sample_sharded(C, n, rank, world, gather=(distributed == 'gather'))
        if self.nf.pipelined_rows(n):
            # large draws: prior / H2D / inverse kernel / D2H overlapped over row chunks, same values
            return self.nf.sample_to_host(C)
        if type(C) != type(1):
            C = _to_device_f32(C, self.nf.engine().device)
        with torch.no_grad():       # realnvp.py:280 detaches the sample at once: no graph is recorded in the first place
            X = self.nf.sample(C).cpu().detach().numpy()
        return X

    def _sample_sharded(self, C, n, rank, world, gather):
        import torch.distributed as dist
        eng = self.nf.engine()
        lo, hi = shard_bounds(0, n, rank, world)
        Cl = None if type(C) == type(1) else _to_device_f32(C[lo:hi], eng.device)
        if isinstance(self.prior, StandardNormalPrior) and not self.prior.host_rng:
            # counter-based prior: every rank uses rank 0's seed and its own global rows
            t = torch.tensor([self.prior.next_seed()], dtype=torch.int64, device=eng.device)
            broadcast_(t, src=0)
            x = eng.sample(hi - lo, Cl, int(t.item()), row_offset=lo)
        else:
            if isinstance(self.prior, StandardNormalPrior):
                # the full reference stream, sliced: every rank consumes its generator like a single process would; drawn on
                # the device where that is validated (nflow.HostStreamOnDevice: 1 ms instead of 34 per 16M numbers)
                z = self.prior.sample((n,))[lo:hi].contiguous()
            else:
                z = torch.as_tensor(self.prior.sample((n,)), dtype=torch.float32)[lo:hi].to(eng.device).contiguous()
            x = eng.inverse(z, Cl, out=z) if hi > lo else z
        if not gather:
            return x.cpu().detach().numpy()
        # equal-size all_gather: pad every share to the largest one
        width = x.shape[1]
        cap = -(-n // world)
        buf = torch.zeros((cap, width), dtype=torch.float32, device=eng.device)
        buf[:hi - lo] = x
        parts = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(parts, buf)
        rows = [shard_bounds(0, n, r, world) for r in range(world)]
        return torch.cat([p[:b - a] for p, (a, b) in zip(parts, rows)]).cpu().numpy()
