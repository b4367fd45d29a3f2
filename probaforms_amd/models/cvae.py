"""Conditional VAE, MI355X build (BASELINE.json configs[4]).

Mirrors `probaforms.models.cvae` (/root/reference/probaforms/models/cvae.py): `Encoder`, `Decoder`
and `CVAE(GenModel)` with the reference's constructor defaults, `state_dict` keys
(`model.{2k}.weight`, `mu.weight`, `log_sigma.weight`, ...), RNG consumption (networks are RE-BUILT
on every fit, cvae.py:164-184; eps for sample_z and the latent draws of `sample` come from the
global CPU generator, cvae.py:187,285) and return values (`fit` returns self; `loss_history` holds
one full-data loss per epoch).  The encoder, the reparameterisation, the decoder, the KL + MSE loss,
their backward and Adam run in librnvp_hip.so (include/cvae_hip.h, include/rnvp_hip.h).
"""
import numpy as np
import torch
import torch.nn as nn

from .. import _hip
from .._engine import (FlatAdam, _perm_pool, all_reduce_sum, batch_bounds, broadcast_, default_device, dist_info,
                       flatten_parameters, is_flat, permutation_from_seed, require_hip, shard_bounds)
from .interfaces import GenModel

DEVICE = default_device()


def _trunk(n_inputs, hidden, activation):
    widths = [n_inputs] + list(hidden)
    net = nn.Sequential()
    for w_in, w_out in zip(widths[:-1], widths[1:]):
        net.append(nn.Linear(w_in, w_out))
        net.append(nn.Tanh() if activation == 'tanh' else nn.ReLU())
    return net


class _CvaeCore:
    """flat storage shared by an Encoder/Decoder pair, in the order include/cvae_hip.h expects"""

    def __init__(self, encoder, decoder, d, c, lat, hidden, activation, device):
        require_hip(device)
        self.device = torch.device(device)
        self.shape = _hip.CvaeShape.make(d, c, lat, hidden, activation)
        e, dd = encoder, decoder
        trunk = [p for m in e.model if isinstance(m, nn.Linear) for p in (m.weight, m.bias)]
        heads = [e.mu.weight, e.log_sigma.weight, e.mu.bias, e.log_sigma.bias]        # ONE Linear of 2*lat outputs
        dec = [p for m in dd.model if isinstance(m, nn.Linear) for p in (m.weight, m.bias)]
        self.plist = trunk + heads + dec
        self.P = sum(p.numel() for p in self.plist)
        assert self.P == _hip.cvae_param_count(self.shape)
        self.flat = None
        self.ws = None
        self.gbuf = None
        self.sync()

    def sync(self):
        if not is_flat(self.plist, self.flat):
            self.flat = flatten_parameters(self.plist, self.device)
        return self.flat[:self.P]

    def workspace(self, rows):
        nb = _hip.cvae_workspace_bytes(self.shape, rows)
        if self.ws is None or self.ws.numel() < nb:
            self.ws = torch.empty(nb, dtype=torch.uint8, device=self.device)
        return self.ws

    def grads(self):
        if self.gbuf is None:
            self.gbuf = torch.zeros(self.P + 1 + (-(self.P + 1)) % 4, dtype=torch.float32, device=self.device)
        return self.gbuf


def _dev(t, device):
    return None if t is None else torch.as_tensor(t, dtype=torch.float32).to(device).contiguous()


class Encoder(nn.Module):
    """[X || C] -> hidden MLP -> (mu, log_sigma)   (cvae.py:14-64)"""

    def __init__(self, n_inputs, lat_size, hidden=(10,), activation='tanh'):
        super().__init__()
        self.model = _trunk(n_inputs, hidden, activation)
        self.mu = nn.Linear(hidden[-1], lat_size)              # creation order = RNG order: trunk, mu, log_sigma
        self.log_sigma = nn.Linear(hidden[-1], lat_size)
        self._core = None

    def forward(self, X, C=None):
        core = self._core
        if core is None:
            raise RuntimeError("Encoder must belong to a CVAE (its weights live in the CVAE's flat HIP buffer)")
        X, C = _dev(X, core.device), _dev(C, core.device)
        n = X.shape[0]
        mu = torch.empty(n, core.shape.lat, device=core.device); ls = torch.empty_like(mu)
        _hip.cvae_encode(core.shape, core.sync(), X, C, n, mu, ls, core.workspace(n))
        return mu, ls


class Decoder(nn.Module):
    """[Z || C] -> hidden MLP -> X_rec   (cvae.py:68-113)"""

    def __init__(self, n_inputs, n_outputs, hidden=(10,), activation='tanh'):
        super().__init__()
        self.model = _trunk(n_inputs, hidden, activation)
        self.model.append(nn.Linear(hidden[-1], n_outputs))
        self._core = None

    def forward(self, X, C=None):
        core = self._core
        if core is None:
            raise RuntimeError("Decoder must belong to a CVAE (its weights live in the CVAE's flat HIP buffer)")
        Z, C = _dev(X, core.device), _dev(C, core.device)
        out = torch.empty(Z.shape[0], core.shape.d, device=core.device)
        _hip.cvae_decode(core.shape, core.sync(), Z, C, Z.shape[0], out, core.workspace(Z.shape[0]))
        return out


def _perm_into(buf, seed):
    g = torch.Generator()
    g.manual_seed(seed)
    return torch.randperm(buf.numel(), generator=g, out=buf)


class _FitDraws:
    """Every draw CVAE.fit makes on the global CPU generator, made AHEAD of the GPU on a worker thread.

    The reference interleaves, per epoch: the two int64 seed draws of a fresh DataLoader iterator (cvae.py:235), one
    `randn(B, latent)` per batch (sample_z, cvae.py:187) and one `randn(n, latent)` for the full-data loss
    (cvae.py:254-259).  The sequence depends on nothing the GPU computes, so a private generator that starts from the
    global generator's state replays it exactly, epochs ahead; `finish()` leaves the global generator where the
    reference's fit would have left it.  The shuffles themselves (randperm from the drawn seed, private generator) run on
    the shared permutation pool.  Everything is written into a small ring of reusable (pinned) host buffers: fresh
    8 MB tensors per epoch cost more in page faults on the MI355X hosts than the draws themselves, and pinned memory
    uploads asynchronously.  At C5 (n = 1M, latent 2) an epoch needs ~4M normals: the stream, not the GPU (~2.5 ms per
    epoch), bounds the fit -- `noise_rng='device'` lifts that."""

    def __init__(self, n, bounds, lat, n_epochs, device=None, slots=3):
        import queue, threading
        self.n, self.bounds, self.lat, self.n_epochs = n, bounds, lat, n_epochs
        self.gen = torch.Generator()
        self.gen.set_state(torch.get_rng_state())
        pin = device is not None and torch.device(device).type == "cuda" and n * lat >= (1 << 16)
        # Round 4: the SAME stream drawn on the device (nflow.HostStreamOnDevice: torch.randn's bits, validated against torch at
        # run time) whenever every draw of the fit has at least 16 elements (torch draws smaller tensors another way): the eps
        # buffers then live in device memory, the host keeps only the two seed draws per epoch and the shuffle
        from .nflow import HostStreamOnDevice
        self.on_device = (pin and n * lat >= 16 and all((e - s) * lat >= 16 for (s, e) in bounds) and HostStreamOnDevice.usable(device))
        self.device = torch.device(device) if device is not None else None
        self.free, self.ready = queue.Queue(), queue.Queue()
        edev = self.device if self.on_device else None
        if self.on_device:
            # the device draws of an epoch take ~1 ms; what takes time is the epoch's host randperm (12 ms at n = 1M, on the shared
            # pool): a deeper ring lets eight of them run at once (the eps buffers are device memory: 2 x n x latent floats a slot)
            slots = max(slots, 8)
        for _ in range(max(1, min(slots, n_epochs))):
            self.free.put(((torch.empty(n, dtype=torch.int64, pin_memory=pin),
                            torch.empty(n, lat, dtype=torch.float32, device=edev) if edev is not None else torch.empty(n, lat, dtype=torch.float32, pin_memory=pin),
                            torch.empty(n, lat, dtype=torch.float32, device=edev) if edev is not None else torch.empty(n, lat, dtype=torch.float32, pin_memory=pin)), None))
        self.stop = False
        self.thread = threading.Thread(target=self._run, name="cvae-draws", daemon=True)
        self.thread.start()

    def _run(self):
        import queue
        try:
            g = self.gen
            pool = _perm_pool(8 if self.on_device else 2) if self.n >= 65536 else None
            side = torch.cuda.Stream(self.device) if self.on_device else None
            dev_perms = 0
            if self.on_device:
                from .nflow import HostStreamOnDevice
                from .._engine import DeviceShuffle, effective_cpus
                # the first epochs' shuffles on the device too (rnvp_randperm_torch_cpu: torch.randperm's bits): nothing hides the
                # host's 12 ms in front of the first epoch; later ones come from the pool, ahead of the GPU
                if pool is not None and self.n <= DeviceShuffle.MAX_N and DeviceShuffle.usable(self.device):
                    dev_perms = self.n_epochs if effective_cpus() < 4 else 2
            for epoch in range(self.n_epochs):
                slot = None
                while slot is None and not self.stop:
                    try:
                        slot, last_reader = self.free.get(timeout=0.1)
                    except queue.Empty:
                        pass
                if self.stop:
                    return
                if last_reader is not None:
                    # the slot's previous epoch: its H2D copies (pinned host buffers) / its kernels (device eps buffers) must be
                    # done before anything overwrites them.  Waiting HERE stalls neither the consumer nor the GPU.
                    last_reader.synchronize()
                perm, eps, eps_full = slot
                torch.empty((), dtype=torch.int64).random_(generator=g)                    # loader base seed
                seed = int(torch.empty((), dtype=torch.int64).random_(generator=g).item())  # RandomSampler seed
                perm_dev = None
                if epoch < dev_perms:
                    fut = None
                    with torch.cuda.stream(side), torch.cuda.device(self.device):
                        perm_dev = DeviceShuffle.draw(self.n, seed, self.device)                  # complete before end() returns below
                else:
                    fut = pool.submit(_perm_into, perm, seed) if pool else _perm_into(perm, seed)
                if self.on_device:
                    # the generator's state goes down, the epoch's draws are chained on the device (this thread's own stream),
                    # the advanced state comes back (end() waits for the draws): g is where the host draws would have left it
                    with torch.cuda.stream(side):
                        hs = HostStreamOnDevice(self.device, g).begin()
                        for (s, e) in self.bounds:
                            hs.draw(eps[s:e])
                        hs.draw(eps_full)
                        hs.end()
                else:
                    for (s, e) in self.bounds:
                        torch.randn(e - s, self.lat, generator=g, out=eps[s:e])
                    torch.randn(self.n, self.lat, generator=g, out=eps_full)
                self.ready.put((slot, fut, perm_dev))
        except BaseException as ex:               # surfaces in the consumer
            self.ready.put(ex)

    def next_epoch(self):
        """-> (slot, perm [n] int64, eps of the batches [n, latent], eps of the loss pass [n, latent]); the tensors are
        the slot's buffers: give the slot back with release() once every reader of them is ENQUEUED, handing over an event
        recorded behind the last one (device eps buffers are read by the epoch's kernels, not only by its uploads)"""
        item = self.ready.get()
        if isinstance(item, BaseException):
            raise item
        slot, fut, perm_dev = item
        if hasattr(fut, "result"):
            fut.result()
        return slot, (slot[0] if perm_dev is None else perm_dev), slot[1], slot[2]

    def release(self, slot, last_reader=None):
        """last_reader: a recorded torch.cuda.Event (or None when nothing asynchronous reads the slot); the worker waits for it
        before it overwrites the slot"""
        self.free.put((slot, last_reader))

    def finish(self):
        self.thread.join()
        torch.set_rng_state(self.gen.get_state())

    def abort(self):
        self.stop = True


def _rank0(t):
    """under torch.distributed: rank 0's tensor on every rank (in place); otherwise t"""
    if dist_info()[1] > 1:
        broadcast_(t, 0)
    return t


class CVAE(GenModel):
    """Conditional VAE with the reference's interface (cvae.py:116-291).

    CVAE(latent_dim=2, hidden=(10,), activation='tanh', batch_size=32, n_epochs=10, lr=1e-4,
         weight_decay=0, KL_weight=0.001, verbose=0); fit(X, C=None) -> self; sample(C=10)."""

    def __init__(self, latent_dim=2, hidden=(10,), activation='tanh', batch_size=32, n_epochs=10, lr=0.0001,
                 weight_decay=0, KL_weight=0.001, verbose=0, *, noise_rng='host'):
        """noise_rng (build-only, keyword-only): 'host' -- eps of sample_z from the global CPU generator, the reference's
        stream (cvae.py:187), produced ahead of the GPU by _FitDraws; 'device' -- a counter-based draw on the GPU
        (rnvp_prior_normal keyed by one host-drawn seed per batch): not the reference's numbers, no host work per row."""
        super().__init__()
        if noise_rng not in ('host', 'device'):
            raise ValueError("noise_rng must be 'host' or 'device'")
        self.noise_rng = noise_rng
        self.lat_size = latent_dim
        self.hidden = hidden
        self.activation = activation
        self.batch_size = batch_size
        self.n_epochs = n_epochs
        self.latent_dim = latent_dim
        self.lr = lr
        self.weight_decay = weight_decay
        self.KL_weight = KL_weight
        self.verbose = verbose
        self.opt = None
        self._core = None

    def _model_init(self, X, C=None):
        """fresh networks and optimizer on EVERY fit, as the reference (cvae.py:157-184)"""
        require_hip(DEVICE)
        c_len = 0 if C is None else C.shape[1]
        self.encoder = Encoder(X.shape[1] + c_len, self.lat_size, self.hidden, self.activation)
        self.decoder = Decoder(self.lat_size + c_len, X.shape[1], self.hidden, self.activation)
        core = _CvaeCore(self.encoder, self.decoder, X.shape[1], c_len, self.lat_size, self.hidden,
                         self.activation, DEVICE)
        self.encoder._core = self.decoder._core = self._core = core
        _, world = dist_info()
        if world > 1:
            broadcast_(core.flat, 0)
        self.opt = FlatAdam(core.flat.numel(), core.device, lr=self.lr, weight_decay=self.weight_decay)

    def _device_eps(self, rows):
        """counter-based N(0, 1) draw on the GPU, keyed by one int64 from the global CPU generator"""
        seed = int(torch.empty((), dtype=torch.int64).random_().item())
        eps = torch.empty(rows, self.lat_size, dtype=torch.float32, device=self._core.device)
        _hip.prior_normal(seed, 0, rows, self.lat_size, eps)
        return eps

    def compute_loss(self, x_batch, cond_batch, eps=None):
        """KL_weight * KL + MSE on one batch with freshly drawn eps (cvae.py:195-203); no autograd graph.
        eps: the [n, latent] normal draw if the caller already made it (fit's look-ahead stream)"""
        core = self._core
        x, c = _dev(x_batch, core.device), _dev(cond_batch, core.device)
        n = x.shape[0]
        if eps is None:
            eps = self._device_eps(n) if self.noise_rng == 'device' else torch.randn(n, self.lat_size)   # cvae.py:187
        eps = _rank0(eps.to(core.device))
        loss = torch.zeros(1, device=core.device)
        _hip.cvae_loss_grad(core.shape, core.sync(), x, c, None, eps, n, 1.0 / n, self.KL_weight, None, loss,
                            core.workspace(n))
        return loss.reshape(())

    def fit(self, X, C=None):
        self._model_init(X, C)
        core = self._core
        dev = core.device
        Xd = torch.tensor(np.asarray(X), dtype=torch.float, device=dev).contiguous()
        Cd = None if C is None else torch.tensor(np.asarray(C), dtype=torch.float, device=dev).contiguous()
        n = Xd.shape[0]
        rank, world = dist_info()
        self.loss_history = []
        bounds = batch_bounds(n, self.batch_size)
        lr, b1, b2, eps_adam, wd = self.opt.hyper
        bar = None
        if self.verbose >= 1:
            from tqdm.auto import tqdm
            bar = tqdm(total=self.n_epochs, unit='epoch')
        host_noise = self.noise_rng != 'device'
        draws = _FitDraws(n, bounds, self.lat_size, self.n_epochs, dev) if host_noise else None
        perms = None
        if not host_noise:
            from .._engine import PermutationPrefetcher
            perms = PermutationPrefetcher(n, self.n_epochs, device=dev).start()
        epoch_losses = None              # per-batch losses of cvae_fit_epoch (the reference keeps only the per-epoch loss)
        pending = []                     # device scalars of the per-epoch losses, read back once (or one epoch behind)
        held = None                      # the epoch's slot of draw buffers, released behind the epoch's last reader

        def flush(keep):
            while len(pending) > keep:
                v = pending.pop(0).detach().cpu()
                self.loss_history.append(v)
                if bar is not None:
                    bar.update(1); bar.set_description("loss: %.4f" % float(v))

        def upload(t):
            return t.to(dev, non_blocking=True) if dev.type == "cuda" else t.clone()

        def release_slot():
            # Give the epoch's draw buffers back with an event behind everything enqueued so far.  Device-resident eps (the
            # round-4 stream, _FitDraws.on_device) is read in place by cvae_fit_epoch / cvae_train_step / compute_loss, so
            # its slot goes back only after the epoch's LAST kernel is enqueued (an event recorded after the
            # uploads let the worker overwrite noise the GPU had not read yet once all eight slots were in use).
            nonlocal held
            if held is None:
                return
            ev = None
            if dev.type == "cuda":
                ev = torch.cuda.Event(); ev.record()
            draws.release(held, ev)
            held = None

        try:
            for epoch in range(self.n_epochs):
                # data parallel: every rank consumes its own generator like a single process would, but walks rank 0's
                # shuffle and noise (broadcast), so the shards tile ONE permutation and the run equals rank 0's run
                if host_noise:
                    slot, perm_h, eps_h, eps_full_h = draws.next_epoch()           # DataLoader(shuffle=True), cvae.py:235
                    perm = _rank0(upload(perm_h))
                    if perm_h.is_cuda:                  # drawn on the worker's stream (the first epochs): tell the allocator who reads it
                        perm.record_stream(torch.cuda.current_stream(dev))
                    eps_all = _rank0(upload(eps_h))                                # sample_z of every batch, cvae.py:187
                    eps_full = upload(eps_full_h)
                    held = slot
                    if not draws.on_device:
                        release_slot()           # pinned host buffers: the uploads just enqueued are their only readers
                else:
                    perm = _rank0(perms.get(epoch).to(dev))
                    eps_full = None
                if world == 1 and host_noise:
                    # one GPU, the reference's noise stream (already on the device for the whole epoch): all batches in ONE
                    # library call -- a single persistent launch per epoch for the reference's default sizes (cvae_fit_epoch)
                    g = core.grads()
                    if epoch_losses is None or epoch_losses.numel() != len(bounds):
                        epoch_losses = torch.zeros(len(bounds), dtype=torch.float32, device=dev)
                    _hip.cvae_fit_epoch(core.shape, core.sync(), Xd, Cd, perm, eps_all, n, self.batch_size, self.KL_weight,
                                        g[:core.P], epoch_losses, self.opt.exp_avg[:core.P], self.opt.exp_avg_sq[:core.P],
                                        lr, b1, b2, eps_adam, wd, self.opt.step_count + 1, core.workspace(min(n, self.batch_size)))
                    self.opt.step_count += len(bounds)
                    pending.append(self.compute_loss(Xd, Cd, eps_full))      # cvae.py:254-259
                    release_slot()
                    flush(1 if bar is not None else self.n_epochs)
                    continue
                for (s, e) in bounds:
                    B = e - s
                    eps = eps_all[s:e] if host_noise else _rank0(self._device_eps(B))
                    g = core.grads()
                    lo, hi = (s, e) if world == 1 else shard_bounds(s, e, rank, world)
                    if world == 1:          # loss + gradient + Adam in one call (the optimizer fused into the last kernel)
                        self.opt.step_count += 1
                        _hip.cvae_train_step(core.shape, core.sync(), Xd, Cd, perm[lo:hi], eps[lo - s:hi - s], hi - lo, 1.0 / B,
                                             self.KL_weight, g[:core.P], g[core.P:core.P + 1], self.opt.exp_avg[:core.P],
                                             self.opt.exp_avg_sq[:core.P], lr, b1, b2, eps_adam, wd, self.opt.step_count,
                                             core.workspace(B))
                        continue
                    _hip.cvae_loss_grad(core.shape, core.sync(), Xd, Cd, perm[lo:hi], eps[lo - s:hi - s], hi - lo,
                                        1.0 / B, self.KL_weight, g[:core.P], g[core.P:core.P + 1], core.workspace(B))
                    all_reduce_sum(g[:core.P + 1])
                    self.opt.step_count += 1
                    _hip.adam_step(core.sync(), g[:core.P], self.opt.exp_avg[:core.P], self.opt.exp_avg_sq[:core.P],
                                   core.P, lr, b1, b2, eps_adam, wd, self.opt.step_count)
                pending.append(self.compute_loss(Xd, Cd, eps_full))      # cvae.py:254-259
                release_slot()
                flush(1 if bar is not None else self.n_epochs)     # never make the GPU wait for the host between epochs
            flush(0)
            if draws is not None:
                draws.finish()
        finally:
            if draws is not None:
                draws.abort()
            if perms is not None:
                perms.close()
        if bar is not None:
            bar.close()
        return self

    def sample(self, C=10):
        core = self._core
        if type(C) != type(1):
            Z = torch.normal(0, 1, (len(C), self.lat_size))                        # cvae.py:285 (CPU generator)
            X = self.decoder(Z, torch.tensor(np.asarray(C), dtype=torch.float, device=core.device))
        else:
            Z = torch.normal(0, 1, (C, self.lat_size))
            X = self.decoder(Z, None)
        return X.cpu().detach().numpy()
