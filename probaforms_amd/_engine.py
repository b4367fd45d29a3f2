"""Host-side engine of the RealNVP hot path: flat parameter storage, workspaces, the
DataLoader-equivalent shuffle, Adam bookkeeping and data-parallel sharding.

Everything numeric happens in librnvp_hip.so (probaforms_amd/_hip.py); this file only
decides WHICH rows each call sees and keeps the buffers the kernels read and write.
"""
import math
import os
import threading
import weakref

import numpy as np
import torch

from . import _hip


# ----------------------------------------------------------------------------------------
# device selection (reference: module-level DEVICE from env var `device`,
# /root/reference/probaforms/models/realnvp.py:12-15; the build's default is the GPU)
# ----------------------------------------------------------------------------------------
def default_device():
    env = os.environ.get("device")
    if env:
        return torch.device(env)
    return torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))


def require_hip(device):
    if torch.device(device).type != "cuda":
        raise RuntimeError(
            "probaforms_amd runs the RealNVP path on a HIP device only (got device=%r). "
            "Unset the `device` environment variable or set it to cuda[:i]; there is no CPU fallback."
            % (device,))


# ----------------------------------------------------------------------------------------
# shuffle: the batch composition of `DataLoader(dataset, batch_size, shuffle=True)`
# (realnvp.py:237), reproduced by making the same generator calls in the same order.
# ----------------------------------------------------------------------------------------
def loader_permutation(n):
    """Row order of one epoch.  A fresh DataLoader iterator first draws its base seed from the
    global CPU generator, then RandomSampler draws the seed of a private generator and takes
    randperm(n) from it (pinned by tests/golden/loader_indices.npz)."""
    torch.empty((), dtype=torch.int64).random_()                 # _BaseDataLoaderIter base seed
    seed = int(torch.empty((), dtype=torch.int64).random_().item())   # RandomSampler seed
    g = torch.Generator()
    g.manual_seed(seed)
    return torch.randperm(n, generator=g)


def draw_loader_seed():
    """the two int64 draws a fresh DataLoader iterator makes on the global CPU generator; returns the
    RandomSampler seed (the second one)"""
    torch.empty((), dtype=torch.int64).random_()
    return int(torch.empty((), dtype=torch.int64).random_().item())


def permutation_from_seed(n, seed):
    g = torch.Generator()
    g.manual_seed(seed)
    return torch.randperm(n, generator=g)


def effective_cpus():
    """CPUs this process may actually use: the smaller of the affinity mask and the cgroup CPU quota (containers often
    show every host CPU in os.cpu_count() while the quota is a fraction of them -- the MI355X boxes: 256 shown, 16 granted;
    running more threads than that gets the whole process throttled for the rest of each 100 ms period)."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                      # cgroup v2: "<quota|max> <period>"
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:      # cgroup v1
                quota = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                period = int(f.read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return max(1, n)


_PERM_POOL = None
_PERM_POOL_LOCK = threading.Lock()


def _reset_perm_pool_after_fork():
    """a forked child inherits the executor object but none of its worker threads: submit() would wait forever"""
    global _PERM_POOL, _PERM_POOL_LOCK
    _PERM_POOL = None
    _PERM_POOL_LOCK = threading.Lock()


if hasattr(os, "register_at_fork"):
    os.register_at_fork(after_in_child=_reset_perm_pool_after_fork)


_PERM_POOL_WORKERS = 12          # the most any caller asks for (PermutationPrefetcher: min(12, cpus // 2); CVAE's draw worker: 2)


def _perm_pool(workers):
    """One process-wide pool for the permutation workers: starting a thread costs ~3 ms on the MI355X hosts (measured:
    12 threads = 39 ms, a quarter of a 32-epoch fit at n = 1M), so the threads are started once and kept.  Called from
    the fitting thread and from CVAE's draw worker, which keep the returned executor for a whole fit: the pool is
    therefore created ONCE at the largest size any caller needs and never replaced or shut down (ThreadPoolExecutor starts
    its threads lazily, so a caller that needs two pays for two)."""
    global _PERM_POOL
    with _PERM_POOL_LOCK:
        if _PERM_POOL is None:
            from concurrent.futures import ThreadPoolExecutor
            size = max(_PERM_POOL_WORKERS, int(workers))
            _PERM_POOL = (ThreadPoolExecutor(max_workers=size, thread_name_prefix="rnvp-perm"), size)
        return _PERM_POOL[0]


class DeviceShuffle:
    """`torch.randperm(n, generator=g)` of a CPU generator, drawn ON THE DEVICE with the same bits (rnvp_randperm_torch_cpu): the
    DataLoader's epoch shuffle (realnvp.py:235) without the host's serial 9 ns per row.  The library restates what torch's CPU
    kernel does (mt19937 draws, Fisher-Yates swaps); `usable(device)` compares both ways once per process and device and the
    callers keep the host shuffle on any difference (or with RNVP_HOST_SHUFFLE_ON_DEVICE=0)."""

    _ok = {}
    _lock = threading.Lock()
    MAX_N = 0xffffffff // 20 - 1          # torch shuffles larger tensors another way

    @staticmethod
    def draw(n, seed, device, ws=None):
        from .models.nflow import HostStreamOnDevice
        g = torch.Generator()
        g.manual_seed(seed)
        _, mt = HostStreamOnDevice._unpack(g)
        mtd = torch.from_numpy(mt.copy()).to(device)
        out = torch.empty(n, dtype=torch.int64, device=device)
        if ws is None:
            ws = torch.empty(_hip.randperm_workspace_bytes(n), dtype=torch.uint8, device=device)
        _hip.randperm_torch_cpu(mtd, n, out, ws)
        return out

    @classmethod
    def usable(cls, device):
        device = torch.device(device)
        if device.type != "cuda" or os.environ.get("RNVP_HOST_SHUFFLE_ON_DEVICE", "1") == "0":
            return False
        key = (os.getpid(), device.index)
        with cls._lock:
            if key not in cls._ok:
                ok = True
                try:
                    for n, seed in ((70001, 0x1234567887654321), (1000, 5)):
                        with torch.cuda.device(device):
                            got = cls.draw(n, seed, device).cpu()
                        ok = ok and bool(torch.equal(got, permutation_from_seed(n, seed)))
                except Exception:
                    ok = False
                cls._ok[key] = ok
            return cls._ok[key]


class PermutationPrefetcher:
    """Epoch permutations of `DataLoader(shuffle=True)` computed ahead of the GPU.

    Nothing else consumes the global generator during RealNVP.fit, so the per-epoch seed draws can all
    be made up front (same values, same final generator state as the reference's epoch-by-epoch
    draws); the expensive `randperm(n)` calls then use PRIVATE generators and are independent, so a
    few worker threads run them while the GPU trains (randperm releases the GIL).  At n = 1M one
    permutation costs ~9 ms of host time against ~5 ms of GPU time per epoch.

    `device` (optional): the FIRST epochs' permutations are drawn on that GPU instead (DeviceShuffle: 1.8 ms per million rows, the
    same bits) -- nothing hides the host's 9 ms in front of the first epoch, and the second epoch's would be late once the first
    starts early; from the third epoch on the worker threads are ahead of the GPU and cost it nothing.  A host with fewer than four
    usable CPUs draws every epoch on the device."""

    def __init__(self, n, n_epochs, workers=None, lookahead=None, device=None):
        self.n, self.n_epochs = n, n_epochs
        self.seeds = [draw_loader_seed() for _ in range(n_epochs)]
        if workers is None:          # one serial randperm costs ~11 ns per row: enough of them in flight to keep ahead of the GPU
            workers = max(1, min(12, effective_cpus() // 2))
        self.lookahead = max(1, lookahead if lookahead is not None else workers)
        self.pool = _perm_pool(max(1, workers)) if n_epochs >= 1 and n >= 65536 else None
        self.futs = {}
        self.next_submit = 0
        self.device, self.dev_epochs, self._ws, self._early = None, 0, None, {}
        if device is not None and self.pool is not None and n <= DeviceShuffle.MAX_N and DeviceShuffle.usable(device):
            self.device = torch.device(device)
            self.dev_epochs = n_epochs if effective_cpus() < 4 else min(2, n_epochs)
            self.next_submit = self.dev_epochs

    def host_only(self):
        """every epoch from the worker threads after all (a caller that wants the permutations on the host); before the first
        get().  Futures start() already submitted (epochs >= dev_epochs) are kept; only the device-drawn epochs are added."""
        if self.dev_epochs:
            for e in range(self.dev_epochs):
                if e not in self.futs and self.pool is not None:
                    self.futs[e] = self.pool.submit(permutation_from_seed, self.n, self.seeds[e])
            self.next_submit = max(self.next_submit, self.dev_epochs)
            self.dev_epochs = 0
            self._early.clear()
        return self

    def _submit_upto(self, epoch):
        while self.next_submit < self.n_epochs and self.next_submit <= epoch + self.lookahead:
            e = self.next_submit
            self.futs[e] = self.pool.submit(permutation_from_seed, self.n, self.seeds[e])
            self.next_submit += 1

    def start(self):
        """begin computing the first permutations now (they overlap whatever the caller does next, e.g. the upload)"""
        if self.pool is not None:
            self._submit_upto(self.dev_epochs)
            if self.dev_epochs and not self._early:
                # the device-drawn ones too: on their own stream, beside the caller's upload of X and C (copy engines)
                with torch.cuda.device(self.device):
                    st = torch.cuda.Stream(device=self.device)
                    st.wait_stream(torch.cuda.current_stream(self.device))
                    with torch.cuda.stream(st):
                        ws = torch.empty(_hip.randperm_workspace_bytes(self.n), dtype=torch.uint8, device=self.device)
                        # at most TWO ahead (8 bytes a row each): a host with few CPUs draws every epoch on the device, and
                        # n_epochs x n x 8 bytes up front (12.8 GB at 16M rows x 100 epochs) would sit in front of the first
                        # batch; get() draws the later ones on the caller's stream as the fit reaches them
                        for e in range(min(2, self.dev_epochs)):
                            t = DeviceShuffle.draw(self.n, self.seeds[e], self.device, ws)
                            ev = torch.cuda.Event()
                            ev.record(st)
                            self._early[e] = (t, ev)
        return self

    def get(self, epoch):
        """epoch's permutation: a CPU tensor, or (the first epochs, see above) a tensor on `device` drawn on the CURRENT stream"""
        if self.pool is None:
            return permutation_from_seed(self.n, self.seeds[epoch])
        self._submit_upto(epoch)
        if epoch < self.dev_epochs:
            if epoch in self._early:                                  # drawn by start(): hand it to the caller's stream
                t, ev = self._early.pop(epoch)
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(ev)
                t.record_stream(cur)
                return t
            with torch.cuda.device(self.device):
                stream = torch.cuda.current_stream(self.device)
                if self._ws is None or self._ws[1] != stream:        # (a fit draws all of them on one stream: one allocation)
                    self._ws = (torch.empty(_hip.randperm_workspace_bytes(self.n), dtype=torch.uint8, device=self.device), stream)
                return DeviceShuffle.draw(self.n, self.seeds[epoch], self.device, self._ws[0])
        return self.futs.pop(epoch).result()

    def close(self):
        for f in self.futs.values():          # the pool is shared and stays; drop what this fit no longer needs
            f.cancel()
        self.futs.clear()
        self._early.clear()                   # device permutations start() drew and nobody fetched
        self._ws = None


def batch_bounds(n, batch_size):
    """[(start, stop)] of consecutive slices of the permutation; last one ragged (drop_last=False)."""
    return [(s, min(s + batch_size, n)) for s in range(0, n, batch_size)]


def shard_bounds(start, stop, rank, world):
    """Contiguous sub-slice of a global batch owned by `rank`; remainder rows go to the low ranks,
    so a ragged batch may leave high ranks with zero rows (they contribute zero gradients)."""
    rows = stop - start
    base, rem = divmod(rows, world)
    lo = start + rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


# ----------------------------------------------------------------------------------------
# flat parameter storage
# ----------------------------------------------------------------------------------------
def flatten_parameters(params, device):
    """Move `params` (list of nn.Parameter, reference order) into ONE contiguous float32 device
    buffer and re-point every parameter at its slice, so state_dict()/load_state_dict() and the
    kernels see the same memory.  Returns the flat tensor."""
    total = sum(p.numel() for p in params)
    pad = (-total) % 4                                   # float4 kernels: keep 16-byte multiples
    with torch.no_grad():
        if params and all(p.device.type == "cpu" for p in params):
            # freshly built modules: gather on the host, ONE upload (a default flow has 32 small tensors)
            parts = [p.detach().reshape(-1).to(torch.float32) for p in params]
            if pad:
                parts.append(torch.zeros(pad, dtype=torch.float32))
            flat = torch.cat(parts).to(device)
            off = 0
            for p in params:
                n = p.numel()
                p.data = flat[off:off + n].view(p.shape)
                off += n
            return flat
        flat = torch.zeros(total + pad, dtype=torch.float32, device=device)
        off = 0
        for p in params:
            n = p.numel()
            view = flat[off:off + n].view(p.shape)
            view.copy_(p.detach().to(device=device, dtype=torch.float32))
            p.data = view
            off += n
    return flat


def is_flat(params, flat):
    """True while every parameter is still the expected slice of `flat` (a later .to()/.float()
    call re-allocates them)."""
    if flat is None:
        return False
    off = flat.data_ptr()
    for p in params:
        if p.data_ptr() != off or p.dtype != torch.float32 or not p.is_contiguous():
            return False
        off += p.numel() * 4
    return True


# Writes the LIBRARY makes to parameter storage go through raw pointers and leave torch's version counters untouched; they are
# counted here per storage (keyed by its base address), so that an autograd node recorded on a flow OR on one of its layers
# alone (whose parameters are views of the flow's buffer) can tell that its parameters changed since the forward.
_STORAGE_WRITES = {}
_STORAGE_WATCHED = set()


def _forget_storage(k):
    _STORAGE_WRITES.pop(k, None)
    _STORAGE_WATCHED.discard(k)


def note_param_write(t):
    st = t.untyped_storage()
    k = st.data_ptr()
    _STORAGE_WRITES[k] = _STORAGE_WRITES.get(k, 0) + 1
    if k not in _STORAGE_WATCHED:
        # the entry dies with the storage: an allocation that later reuses the address must not inherit the count (a false
        # "parameters changed since the forward"), and a long-lived process that builds many models must not grow the table
        _STORAGE_WATCHED.add(k)
        weakref.finalize(st, _forget_storage, k)


def param_state(param_list):
    """what FlowFunction compares between forward and backward: torch's version counters see in-place edits made through
    tensors (a torch.optim step, `p.add_()`, load_state_dict -- every nn.Parameter keeps its own counter), the per-storage
    write count the library's own updates (Adam, train_step, fit_epoch*), the data pointers a re-allocation"""
    return tuple((p.data_ptr(), p._version, _STORAGE_WRITES.get(p.untyped_storage().data_ptr(), 0)) for p in param_list)


class FlatAdam:
    """State of `torch.optim.Adam(nf.parameters(), lr, weight_decay)` (realnvp.py:205-207) kept as
    two flat buffers next to the flat parameters; the update itself is rnvp_adam_step."""

    def __init__(self, n_params, device, lr, weight_decay, betas=(0.9, 0.999), eps=1e-8):
        self.param_groups = [dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay,
                                  amsgrad=False)]
        self.exp_avg = torch.zeros(n_params, dtype=torch.float32, device=device)
        self.exp_avg_sq = torch.zeros(n_params, dtype=torch.float32, device=device)
        self.step_count = 0

    @property
    def hyper(self):
        g = self.param_groups[0]
        return g["lr"], g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"]

    def zero_grad(self, set_to_none=True):   # gradients live in the engine's scratch buffer
        return None

    def state_dict(self):
        return dict(step=self.step_count, exp_avg=self.exp_avg.clone(), exp_avg_sq=self.exp_avg_sq.clone(),
                    param_groups=[dict(g) for g in self.param_groups])

    def load_state_dict(self, sd):
        self.step_count = int(sd["step"])
        self.exp_avg.copy_(sd["exp_avg"]); self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.param_groups = [dict(g) for g in sd["param_groups"]]


class _Workspace:
    def __init__(self, device):
        self.device = device
        self.buf = None

    def get(self, nbytes):
        if nbytes <= 0:
            return None
        if self.buf is None or self.buf.numel() < nbytes:
            self.buf = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        return self.buf


class FlowEngine:
    """Owns the device-side state of one RealNVP flow and issues the HIP calls.

    layers: list of RealNVPLayer with identical (var_size, cond_size, hidden, activation).
    """

    def __init__(self, layers, device, precision=None, small_calls=None):
        require_hip(device)
        self.device = torch.device(device)
        l0 = layers[0]
        self.L, self.d, self.c = len(layers), l0.var_size, l0.cond_size
        self.hidden, self.activation = tuple(l0.hidden), l0.activation
        for l in layers:
            if (l.var_size, l.cond_size, tuple(l.hidden), l.activation) != (self.d, self.c, self.hidden, self.activation):
                raise ValueError("all coupling layers of a flow must share var_size, cond_size, hidden and activation")
        self.shape = _hip.RnvpShape.make(self.L, self.d, self.c, self.hidden, self.activation, precision=precision,
                                         small_calls=_hip.SMALL_CALLS[small_calls or 'invariant'])
        self.param_list = [p for l in layers for p in l.parameters()]
        self.P = sum(p.numel() for p in self.param_list)
        if self.P != _hip.param_count(self.shape):
            raise RuntimeError("parameter count mismatch between the modules (%d) and librnvp_hip (%d)"
                               % (self.P, _hip.param_count(self.shape)))
        self.flat = None
        self.masks_host = None
        self._mask_key = None
        self.layers = layers
        self._ws = {op: _Workspace(self.device) for op in (_hip.OP_FORWARD, _hip.OP_INVERSE, _hip.OP_TRAIN)}
        self._ws_rows = {}
        self.gbuf = None            # [P + 1]: gradient, then the batch loss (one all-reduce message)
        self.sync_params()

    # -- storage -------------------------------------------------------------------------
    def sync_params(self):
        """(Re)build the flat buffer if the module parameters were moved or replaced."""
        if not is_flat(self.param_list, self.flat):
            self.flat = flatten_parameters(self.param_list, self.device)
        # masks are plain attributes of the layers (realnvp.py:68); re-read them only when one was
        # replaced or modified in place (identity + version counter), not on every call
        key = tuple((id(l.mask), l.mask._version) for l in self.layers)
        if key != self._mask_key:
            self._mask_key = key
            masks = torch.stack([l.mask.detach().to("cpu").to(torch.uint8).reshape(-1) for l in self.layers])
            self.masks_host = np.ascontiguousarray(masks.numpy())
            self.masks = masks.to(self.device).contiguous()
            # declare the reference's alternating pattern (realnvp.py:199) when that is what we hold
            self.shape.alt_masks = _hip.RnvpShape.classify_masks(self.masks_host)
        return self.flat

    @property
    def params(self):
        return self.flat[:self.P]

    def workspace(self, op, rows):
        rows = max(int(rows), 1)
        key = op
        if self._ws_rows.get(key, 0) < rows:
            self._ws_rows[key] = rows
        return self._ws[op].get(_hip.workspace_bytes(self.shape, op, self._ws_rows[key]))

    def _cond(self, c, n):
        if self.c == 0:
            if c is not None:
                raise RuntimeError("this flow was built without conditions (cond_size=0) but C was given")
            return None
        if c is None:
            # same failure mode as the reference: the first Linear sees d instead of d+c columns
            raise RuntimeError("mat1 and mat2 shapes cannot be multiplied (%dx%d and %dx%d): the flow was fit "
                               "with cond_size=%d, conditions are required" % (n, self.d, self.d + self.c,
                                                                             self.hidden[0], self.c))
        if c.shape != (n, self.c):
            raise RuntimeError("C must have shape (%d, %d), got %s" % (n, self.c, tuple(c.shape)))
        return c

    def _rows(self, x, rows):
        if x.dim() != 2 or x.shape[1] != self.d:
            raise RuntimeError("X must have shape (n, %d), got %s" % (self.d, tuple(x.shape)))
        return x.shape[0] if rows is None else rows.numel()

    # -- hot path --------------------------------------------------------------------------
    def forward(self, x, c, rows=None, want_z=True, want_logdet=False, want_logp=True, want_sum=False):
        """-> (z, logdet, logp, logp_sum); unrequested entries are None."""
        self.sync_params()
        n = self._rows(x, rows)
        c = self._cond(c, x.shape[0])
        dev = self.device
        z = torch.empty(n, self.d, dtype=torch.float32, device=dev) if want_z else None
        ld = torch.empty(n, dtype=torch.float32, device=dev) if want_logdet else None
        lp = torch.empty(n, dtype=torch.float32, device=dev) if want_logp else None
        tot = torch.zeros(1, dtype=torch.float32, device=dev) if want_sum else None
        ws = self.workspace(_hip.OP_FORWARD, n)
        _hip.forward_logprob(self.shape, self.params, self.masks, x, c, rows, n, z, ld, lp, tot, ws)
        return z, ld, lp, tot

    def forward_autograd(self, x, c):
        """(z, logdet) = f(x, c) WITH an autograd graph: `FlowFunction` records the inputs, and its backward is the
        hand-derived HIP backward (rnvp_backward) -- what torch autograd builds op by op in the reference when a user
        differentiates nf.log_prob / layer.f (nflow.py:107-117, realnvp.py:246-250)."""
        self.sync_params()
        c = self._cond(c, x.shape[0])
        return FlowFunction.apply(self, x, c, *self.param_list)

    def inverse_autograd(self, z, c):
        """x = g(z, c) WITH an autograd graph (InverseFunction: backward = rnvp_inverse_backward)"""
        self.sync_params()
        c = self._cond(c, z.shape[0])
        return InverseFunction.apply(self, z, c, *self.param_list)

    def wants_graph(self, *tensors):
        """grad mode is on and something that feeds the call requires a gradient"""
        return torch.is_grad_enabled() and (any(torch.is_tensor(t) and t.requires_grad for t in tensors) or
                                            any(p.requires_grad for p in self.param_list))

    def inverse(self, z, c, out=None):
        self.sync_params()
        n = self._rows(z, None)
        c = self._cond(c, n)
        x = torch.empty_like(z) if out is None else out
        _hip.inverse(self.shape, self.params, self.masks, z, c, n, x, self.workspace(_hip.OP_INVERSE, n))
        return x

    def sample(self, n, c, seed, row_offset=0, out=None):
        """x[r] = g(z(seed, row_offset + r), c[r]) for r < n: the counter-based prior drawn inside the inverse
        kernel (rnvp_sample); `out` [n, d] optional."""
        self.sync_params()
        n = int(n)
        c = self._cond(c, n)
        x = torch.empty(n, self.d, dtype=torch.float32, device=self.device) if out is None else out
        if tuple(x.shape) != (n, self.d):
            raise RuntimeError("out must have shape (%d, %d), got %s" % (n, self.d, tuple(x.shape)))
        _hip.sample(self.shape, self.params, self.masks, c, n, seed, row_offset, x, self.workspace(_hip.OP_INVERSE, n))
        return x

    def ensure_gbuf(self):
        if self.gbuf is None or self.gbuf.device != self.device:
            pad = (-(self.P + 1)) % 4
            self.gbuf = torch.zeros(self.P + 1 + pad, dtype=torch.float32, device=self.device)
        return self.gbuf

    def loss_grad(self, x, c, rows, n_rows, inv_B):
        """gbuf[:P] <- d loss / d params for this shard, gbuf[P] <- its share of the batch loss."""
        g = self.ensure_gbuf()
        ws = self.workspace(_hip.OP_TRAIN, max(n_rows, 1))
        _hip.loss_grad(self.shape, self.params, self.masks, x, c, rows, n_rows, inv_B,
                       g[:self.P], g[self.P:self.P + 1], ws)
        return g

    def loss_grad_prior(self, prior, x, c, rows, n_rows, inv_B):
        """loss_grad for a prior other than the fused N(0, I) (realnvp.py:189 keeps a user-assigned prior; nflow.py:115
        calls its log_prob): z from the forward kernel, the prior's log-density and its gradient w.r.t. z from torch
        autograd, then the hand-derived backward seeded with d loss / d z (rnvp_loss_grad_zseed)."""
        g = self.ensure_gbuf()
        if n_rows == 0:
            g[:self.P + 1].zero_()
            return g
        z, _, _, _ = self.forward(x, c, rows=rows, want_z=True, want_logp=False)
        z = z.detach().requires_grad_(True)
        with torch.enable_grad():
            lp_sum = prior.log_prob(z).sum()
            (gz,) = torch.autograd.grad(-inv_B * lp_sum, z)
        ws = self.workspace(_hip.OP_TRAIN, max(n_rows, 1))
        _hip.loss_grad_zseed(self.shape, self.params, self.masks, x, c, rows, n_rows, inv_B, gz.contiguous().float(),
                             g[:self.P], g[self.P:self.P + 1], ws)
        g[self.P:self.P + 1] -= inv_B * lp_sum.detach().float()
        return g

    def adam(self, opt):
        note_param_write(self.flat)
        lr, b1, b2, eps, wd = opt.hyper
        opt.step_count += 1
        g = self.ensure_gbuf()
        _hip.adam_step(self.params, g[:self.P], opt.exp_avg[:self.P], opt.exp_avg_sq[:self.P], self.P,
                       lr, b1, b2, eps, wd, opt.step_count)

    def finish_dp_step(self, opt, loss_out):
        """data parallel, after the all-reduce of gbuf[:P+1]: batch loss -> loss_out[0:1] and Adam, one launch"""
        note_param_write(self.flat)
        lr, b1, b2, eps, wd = opt.hyper
        opt.step_count += 1
        g = self.ensure_gbuf()
        _hip.dp_finish_step(self.params, g, opt.exp_avg[:self.P], opt.exp_avg_sq[:self.P], self.P,
                            lr, b1, b2, eps, wd, opt.step_count, loss_out)

    def fit_epoch(self, opt, x, c, perm, batch_size, losses):
        """single-GPU: all batches of one epoch in ONE library call (rnvp_fit_epoch)"""
        note_param_write(self.flat)
        lr, b1, b2, eps, wd = opt.hyper
        n = perm.numel()
        g = self.ensure_gbuf()
        ws = self.workspace(_hip.OP_TRAIN, min(n, batch_size))
        _hip.fit_epoch(self.shape, self.params, self.masks, x, c, perm, n, batch_size, g[:self.P], losses,
                       opt.exp_avg[:self.P], opt.exp_avg_sq[:self.P], lr, b1, b2, eps, wd, opt.step_count + 1, ws)
        opt.step_count += len(batch_bounds(n, batch_size))

    def fit_epochs(self, opt, x, c, perms, batch_size, losses):
        """single-GPU: every batch of SEVERAL epochs in one library call (rnvp_fit_epochs): perms [n_epochs, n], losses
        [n_epochs, batches per epoch].  For a model that fits one CU's LDS that is ONE launch for the whole fit."""
        note_param_write(self.flat)
        lr, b1, b2, eps, wd = opt.hyper
        n_epochs, n = perms.shape
        g = self.ensure_gbuf()
        ws = self.workspace(_hip.OP_TRAIN, min(n, batch_size))
        _hip.fit_epochs(self.shape, self.params, self.masks, x, c, perms, n, batch_size, n_epochs, g[:self.P], losses,
                        opt.exp_avg[:self.P], opt.exp_avg_sq[:self.P], lr, b1, b2, eps, wd, opt.step_count + 1, ws)
        opt.step_count += n_epochs * len(batch_bounds(n, batch_size))

    def fit_epoch_dp(self, opt, comm, x, c, perm, batch_size, losses, exchange=None, rank=0, world=1, chunks=1):
        """data parallel: all batches of one epoch in ONE library call on ONE stream: per batch this rank's loss + gradient, the
        all-reduce of [gradient | loss], loss read-out + Adam.  comm: the library's RCCL communicator (rnvp_fit_epoch_dp);
        exchange: a callable(tensor, count) summing in place over the ranks -- process groups that are not RCCL, e.g. gloo
        (rnvp_fit_epoch_dp_cb; rank / world place this process); neither: the same loop for a single rank."""
        note_param_write(self.flat)
        lr, b1, b2, eps, wd = opt.hyper
        n = perm.numel()
        g = self.ensure_gbuf()
        ws = self.workspace(_hip.OP_TRAIN, min(n, batch_size))
        if exchange is not None:
            _hip.fit_epoch_dp_cb(exchange, rank, world, self.shape, self.params, self.masks, x, c, perm, n, batch_size,
                                 g[:self.P + 1], losses, opt.exp_avg[:self.P], opt.exp_avg_sq[:self.P], lr, b1, b2, eps, wd,
                                 opt.step_count + 1, ws, chunks=chunks)
        else:
            _hip.fit_epoch_dp(comm, self.shape, self.params, self.masks, x, c, perm, n, batch_size, g[:self.P + 1], losses,
                              opt.exp_avg[:self.P], opt.exp_avg_sq[:self.P], lr, b1, b2, eps, wd, opt.step_count + 1, ws)
        opt.step_count += len(batch_bounds(n, batch_size))

    def train_step(self, opt, x, c, rows, n_rows, inv_B, loss_out):
        """single-GPU fused step: loss+grad, Adam; the batch loss lands in loss_out[0:1]."""
        note_param_write(self.flat)
        lr, b1, b2, eps, wd = opt.hyper
        opt.step_count += 1
        g = self.ensure_gbuf()
        ws = self.workspace(_hip.OP_TRAIN, max(n_rows, 1))
        _hip.train_step(self.shape, self.params, self.masks, x, c, rows, n_rows, inv_B, g[:self.P], loss_out,
                        opt.exp_avg[:self.P], opt.exp_avg_sq[:self.P], lr, b1, b2, eps, wd, opt.step_count, ws)


def _cond_workspace(engine, rows):
    """workspace of rnvp_backward_cond / rnvp_inverse_backward for `rows` rows (the any-shape 16-row MFMA kernel, whatever
    family serves the flow otherwise; the one-thread-per-row VALU kernel for the shapes whose tile image that kernel cannot
    hold); raises where no kernel can take the shape"""
    nb = _hip.backward_cond_workspace_bytes(engine.shape, max(int(rows), 1))
    if nb == 0:
        raise RuntimeError("the gradient with respect to the conditions and the backward through g / sample: this flow's "
                           "hidden sizes exceed a CU's LDS even for an 8-row tile of the one-thread-per-row kernel")
    # kept on the engine like the other operations' workspaces (grown, never shrunk): a user's training loop calls this every step
    ws = getattr(engine, "_ws_cond", None)
    if ws is None:
        ws = engine._ws_cond = _Workspace(engine.device)
    return ws.get(nb)


def _split_param_grads(eng, gflat, needs, first):
    grads, off = [], 0
    for i, p in enumerate(eng.param_list):
        k = p.numel()
        grads.append(gflat[off:off + k].view(p.shape) if needs[first + i] else None)
        off += k
    return tuple(grads)


class FlowFunction(torch.autograd.Function):
    """z, logdet = flow.f(x, c) as one autograd node.

    forward : rnvp_forward_logprob (the fused stack; L = 1 for a single RealNVPLayer)
    backward: rnvp_backward -- (d loss / d z, d loss / d logdet) -> d loss / d x and d loss / d every parameter, the
              hand-derived backward of SURVEY.md 3.3 (the same kernels RealNVP.fit uses, seeded by the caller).  When the
              CONDITIONS require a gradient (they enter through torch.cat((X * mask, C)) in the reference, realnvp.py:92: a
              learned condition encoder) the call is rnvp_backward_cond, which also returns d loss / d c.
    The parameters are passed as inputs only so that autograd routes their gradients; the kernels read the engine's flat
    buffer, of which every parameter is a view."""

    @staticmethod
    def forward(ctx, engine, x, c, *params):
        n = x.shape[0]
        x = x.detach().contiguous()
        c = None if c is None else c.detach().contiguous()
        z = torch.empty_like(x)
        ld = torch.empty(n, dtype=torch.float32, device=x.device)
        if n > 0:
            _hip.forward_logprob(engine.shape, engine.params, engine.masks, x, c, None, n, z, ld, None, None,
                                 engine.workspace(_hip.OP_FORWARD, n))
        ctx.engine, ctx.n = engine, n
        ctx.save_for_backward(x, c if c is not None else x.new_empty(0))
        ctx.has_c = c is not None
        ctx.version = param_state(engine.param_list)
        return z, ld

    @staticmethod
    def backward(ctx, gz, gld):
        eng, n = ctx.engine, ctx.n
        x, c = ctx.saved_tensors
        c = c if ctx.has_c else None
        if param_state(eng.param_list) != ctx.version:
            raise RuntimeError("the flow's parameters were modified in place between forward and backward "
                               "(one of the variables needed for gradient computation has been modified)")
        dev = x.device
        gz = torch.zeros_like(x) if gz is None else gz.to(torch.float32).contiguous()
        gld = torch.zeros(n, dtype=torch.float32, device=dev) if gld is None else gld.to(torch.float32).contiguous()
        gflat = torch.zeros(eng.P + (-eng.P) % 4, dtype=torch.float32, device=dev)
        gx = torch.empty_like(x) if ctx.needs_input_grad[1] else None
        want_gc = c is not None and ctx.needs_input_grad[2]
        gc = torch.empty_like(c) if want_gc else None
        if n > 0 and want_gc:
            _hip.backward_cond(eng.shape, eng.params, eng.masks, x, c, None, n, gz, gld, gflat[:eng.P], gx, gc,
                               _cond_workspace(eng, n))
        elif n > 0:
            _hip.backward(eng.shape, eng.params, eng.masks, x, c, None, n, gz, gld, gflat[:eng.P], gx,
                          eng.workspace(_hip.OP_TRAIN, n))
        else:
            if gx is not None:
                gx.zero_()
        return (None, gx, gc) + _split_param_grads(eng, gflat, ctx.needs_input_grad, 3)


class InverseFunction(torch.autograd.Function):
    """x = flow.g(z, c) as one autograd node (realnvp.py:120-129; nflow.py:141-145 returns the sample WITH its graph).

    forward : rnvp_inverse;   backward: rnvp_inverse_backward -- d loss / d x -> d loss / d z, d loss / d c and d loss / d every
    parameter (reverse-KL and other sample-based losses written against the reference's classes)."""

    @staticmethod
    def forward(ctx, engine, z, c, *params):
        n = z.shape[0]
        z = z.detach().contiguous()
        c = None if c is None else c.detach().contiguous()
        x = torch.empty_like(z)
        if n > 0:
            _hip.inverse(engine.shape, engine.params, engine.masks, z, c, n, x, engine.workspace(_hip.OP_INVERSE, n))
        ctx.engine, ctx.n = engine, n
        ctx.save_for_backward(z, c if c is not None else z.new_empty(0))
        ctx.has_c = c is not None
        ctx.version = param_state(engine.param_list)
        return x

    @staticmethod
    def backward(ctx, gx):
        eng, n = ctx.engine, ctx.n
        z, c = ctx.saved_tensors
        c = c if ctx.has_c else None
        if param_state(eng.param_list) != ctx.version:
            raise RuntimeError("the flow's parameters were modified in place between forward and backward "
                               "(one of the variables needed for gradient computation has been modified)")
        gx = torch.zeros_like(z) if gx is None else gx.to(torch.float32).contiguous()
        gflat = torch.zeros(eng.P + (-eng.P) % 4, dtype=torch.float32, device=z.device)
        gz = torch.empty_like(z) if ctx.needs_input_grad[1] else None
        gc = torch.empty_like(c) if (c is not None and ctx.needs_input_grad[2]) else None
        if n > 0:
            _hip.inverse_backward(eng.shape, eng.params, eng.masks, z, c, n, gx, gflat[:eng.P], gz, gc, _cond_workspace(eng, n))
        else:
            if gz is not None:
                gz.zero_()
        return (None, gz, gc) + _split_param_grads(eng, gflat, ctx.needs_input_grad, 3)


# ----------------------------------------------------------------------------------------
# data-parallel helpers (one process per GPU, torch.distributed; backend "nccl" is RCCL)
# ----------------------------------------------------------------------------------------
def dist_info():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def all_reduce_sum(t):
    import torch.distributed as dist
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def broadcast_(t, src=0):
    import torch.distributed as dist
    dist.broadcast(t, src=src)
    return t


_DP_COMM = {}          # (world, rank, device index) -> the library's RCCL communicator of this process


def dp_communicator(device):
    """The library's own RCCL communicator for the ranks of torch.distributed's default group (one per process, made on
    first use: rank 0 draws the id, torch.distributed carries it to the others).  None when the job does not run one
    rank per GPU over RCCL (world 1; gloo test jobs that put several ranks on one GPU) -- callers then take the per-batch
    Python loop over torch.distributed instead."""
    import torch.distributed as dist
    rank, world = dist_info()
    if world == 1 or dist.get_backend() != "nccl":
        return None
    key = (world, rank, torch.device(device).index)
    pg = dist.distributed_c10d._get_default_group()
    cached = _DP_COMM.get(key)
    if cached is not None and cached[0] is not pg:
        # the default process group was destroyed and re-created since: the old communicator's peers are gone
        try:
            _hip.dp_destroy(cached[1])
        except Exception:
            pass
        cached = None
    if cached is None:
        # Whether the communicator exists must be the SAME answer on every rank (it selects which collectives the fit
        # issues): each step is followed by an agreement over torch.distributed, and one rank's failure sends all of them
        # to the per-batch loop (with a warning -- slower, not wrong).
        dev = torch.device(device)

        def agree(ok):
            t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(t.item())
        uid, comm, err = None, None, None
        try:
            uid = _hip.dp_unique_id() if rank == 0 else None
        except Exception as e:          # librccl missing / not loadable
            err = e
        # the 128 id bytes + a validity flag travel as ONE uint8 tensor on the engine's device (a pickled object list would be
        # staged on torch.cuda.current_device(), which need not be this rank's engine device)
        box = torch.zeros(129, dtype=torch.uint8, device=dev)
        if rank == 0 and uid is not None:
            box[:128] = torch.frombuffer(bytearray(uid), dtype=torch.uint8).to(dev)
            box[128] = 1
        dist.broadcast(box, src=0)
        host = box.cpu()
        have_id = bool(host[128].item())
        if agree(have_id):
            try:
                with torch.cuda.device(dev):
                    comm = _hip.dp_init(bytes(host[:128].numpy().tobytes()), rank, world)
            except Exception as e:
                err = e
            if not agree(comm is not None):
                _hip.dp_destroy(comm)
                comm = None
        if comm is None:
            import warnings
            warnings.warn("probaforms_amd: no RCCL communicator for the in-library data-parallel loop (%r); "
                          "falling back to the per-batch loop over torch.distributed" % (err,))
        cached = _DP_COMM[key] = (pg, comm)
    return cached[1]


def run_epoch(engine, opt, comm, X, C, perm, bounds, batch_size, rank, world, losses, prior=None):
    """every batch of one epoch (realnvp.py:237-254), enqueued without waiting for the GPU:
    one GPU            -> rnvp_fit_epoch (a fused loss + gradient + Adam step per batch, looped in the library);
    N ranks over RCCL  -> rnvp_fit_epoch_dp (per batch: this rank's share, all-reduce on the same stream, Adam);
    N ranks, another process group (gloo) -> rnvp_fit_epoch_dp_cb: the same in-library loop, torch.distributed's all_reduce as the exchange;
    a user-assigned prior -> batch by batch from here with torch.distributed.
    bench.py times exactly this function."""
    if world == 1 and prior is None and comm is None:
        engine.fit_epoch(opt, X, C, perm, batch_size, losses)
        return
    if prior is None and (comm is not None or world == 1):
        engine.fit_epoch_dp(opt, comm, X, C, perm, batch_size, losses)
        return
    if prior is None:
        # a process group that is not RCCL (gloo: CPU test jobs, several ranks on one GPU): the same in-library loop, the exchange
        # handed in as a callback over torch.distributed -- one code path for the shard arithmetic whatever carries the sum
        engine.fit_epoch_dp(opt, None, X, C, perm, batch_size, losses, exchange=lambda t, count: all_reduce_sum(t), rank=rank, world=world)
        return
    for k, (s, e) in enumerate(bounds):
        lo, hi = shard_bounds(s, e, rank, world)
        if prior is None:
            g = engine.loss_grad(X, C, perm[lo:hi], hi - lo, 1.0 / (e - s))
        else:
            g = engine.loss_grad_prior(prior, X, C, perm[lo:hi], hi - lo, 1.0 / (e - s))
        if world == 1:
            losses[k:k + 1].copy_(g[engine.P:engine.P + 1])
            engine.adam(opt)
            continue
        all_reduce_sum(g[:engine.P + 1])
        engine.finish_dp_step(opt, losses[k:k + 1])


PROTOCOL_NAN_BITS = 0x7fc0dead       # csrc/rnvp_mfma_layer.h kProtocolNaN


def check_losses(host):
    """Raise if a batch loss is the library's protocol-error NaN: a wave of the training kernel gave up a bounded wait
    (csrc/rnvp_mfma_layer.h spin_nap).  The step's finish kernel then applied no Adam update -- the parameters hold their
    last good values -- and wrote this payload instead of a loss.  A NaN that training itself produced (a diverged fit)
    passes through, as it does in the reference."""
    if host.numel() and bool(torch.isnan(host).any()):
        bits = host.detach().contiguous().view(torch.int32)
        if bool((bits == PROTOCOL_NAN_BITS).any()):
            raise RuntimeError("probaforms_amd: a training launch ended on a synchronisation time-out inside the kernel "
                               "(protocol error); the optimizer step was skipped and the parameters are unchanged since the "
                               "last good batch")
    return host


def fit_epochs(engine, opt, X, C, batch_size, n_epochs, loss_history, epoch_hook=None, prior=None, perms=None):
    """The batch loop of RealNVP.fit (realnvp.py:235-262) on device-resident X [n,d], C [n,c].

    Single GPU: one rnvp_fit_epoch call per epoch (a fused rnvp_train_step per batch, looped inside the
    library).  With torch.distributed initialised every
    rank walks the SAME permutation (rank 0's loader seeds are broadcast), takes its contiguous share of each global batch, and the
    flat [gradient | loss] buffer is all-reduced (SUM) before an identical Adam step on every
    rank -- gradients are scaled by 1/B_global inside the kernel, so the sum is the batch mean.  Over RCCL (one rank per
    GPU) that loop, too, is one library call per epoch on one stream (rnvp_fit_epoch_dp on the library's own communicator);
    other process groups (gloo) run the same library loop with torch.distributed's all_reduce handed in as the exchange
    (rnvp_fit_epoch_dp_cb); only user-assigned priors run batch by batch from here.
    Losses stay on the device; one copy per epoch feeds loss_history (one entry per batch, as
    realnvp.py:254)."""
    rank, world = dist_info()
    n = X.shape[0]
    bounds = batch_bounds(n, batch_size)
    dev = engine.device
    # the module parameters may have been re-allocated since the last call (nf.cpu(), .float(),
    # load_state_dict(assign=True)): train what the modules hold now, not a stale flat buffer
    # Data parallel: rank 0's flat buffer is broadcast at the start of EVERY fit -- P floats once per call.  Whether a
    # rank re-allocated its parameters is a rank-local fact and must not decide whether a collective is entered.
    engine.sync_params()
    if world > 1:
        broadcast_(engine.flat, src=0)
        note_param_write(engine.flat)
    if perms is None:
        perms = PermutationPrefetcher(n, n_epochs, device=dev)
    try:
        if world > 1:
            # every rank must walk rank 0's shuffle, whatever state its own generator is in (each rank still
            # consumes its generator exactly like a single process would)
            t = torch.tensor(perms.seeds, dtype=torch.int64, device=dev)
            broadcast_(t, src=0)
            perms.seeds = [int(v) for v in t.cpu()]
        # One GPU, nobody watching the epochs go by (verbose == 0), a model small enough to live in one CU's LDS: the whole fit
        # is one library call (one persistent launch: the parameters never leave LDS between epochs).  The permutations of all
        # epochs are a few MB at the sizes such models are fitted on.
        if (world == 1 and prior is None and epoch_hook is None and n_epochs > 0 and dev.type == "cuda"
                and n_epochs * n <= (1 << 24) and _hip.fit_epoch_resident(engine.shape, batch_size)):
            host = torch.empty((n_epochs, n), dtype=torch.int64, pin_memory=True)
            perms.host_only()
            for e in range(n_epochs):
                host[e].copy_(perms.get(e))
            losses = torch.zeros((n_epochs, len(bounds)), dtype=torch.float32, device=dev)
            engine.fit_epochs(opt, X, C, host.to(dev, non_blocking=True), batch_size, losses)
            flat = losses.reshape(-1).cpu()
            loss_history.extend(flat.unbind(0))          # 0-d tensors like the reference's (realnvp.py:254), views of one buffer
            return loss_history
        _fit_epochs_loop(engine, opt, X, C, batch_size, n_epochs, loss_history, epoch_hook, perms, bounds, rank, world, prior)
    finally:
        perms.close()
    return loss_history


def _fit_epochs_loop(engine, opt, X, C, batch_size, n_epochs, loss_history, epoch_hook, perms, bounds, rank, world,
                     prior=None):
    """prior: None for the fused N(0, I); otherwise the user's prior object (log_prob differentiable by torch)"""
    dev = engine.device
    comm = dp_communicator(dev) if world > 1 else None
    # The next epoch's permutation (8 bytes per row) is uploaded on a side stream while this epoch's kernels run:
    # the staged copy blocks only the host, which has nothing else to do until the epoch's losses come back.
    side = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None

    def upload(epoch):
        if side is None:
            return perms.get(epoch).to(dev), None
        with torch.cuda.stream(side):
            d = perms.get(epoch).to(dev, non_blocking=True)     # from the worker threads, or drawn on this stream (the first epochs)
            ev = torch.cuda.Event()
            ev.record(side)
        return d, ev

    def read_back(epoch, losses):
        host = check_losses(losses.cpu())
        loss_history.extend(host.unbind(0))
        if epoch_hook is not None:
            epoch_hook(epoch, host)                      # the epoch's per-batch losses, in batch order

    # Epoch e's losses are read back only after epoch e+1 has been enqueued (and its successor's permutation uploaded),
    # so the GPU never waits for the host between epochs; loss_history / the progress hook trail by one epoch.
    pending = None
    nxt = upload(0) if n_epochs > 0 else None
    ok = False
    try:
        for epoch in range(n_epochs):
            perm, ev = nxt
            if ev is not None:
                torch.cuda.current_stream(dev).wait_event(ev)
                perm.record_stream(torch.cuda.current_stream(dev))
            losses = torch.zeros(len(bounds), dtype=torch.float32, device=dev)
            run_epoch(engine, opt, comm, X, C, perm, bounds, batch_size, rank, world, losses, prior)
            if epoch + 1 < n_epochs:
                nxt = upload(epoch + 1)
            if pending is not None:
                done, pending = pending, None
                read_back(*done)
            pending = (epoch, losses)
        ok = True
    finally:
        # the last epoch's losses; while an exception unwinds (a HIP error, KeyboardInterrupt) nothing is read back: a
        # device sync could block or raise again and hide the original failure, and a partial epoch must not reach
        # loss_history or the user's hook
        if ok and pending is not None:
            read_back(*pending)
