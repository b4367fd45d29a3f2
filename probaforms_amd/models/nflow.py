"""Flow container and invertible-layer protocol, MI355X build.

Mirrors `probaforms.models.nflow` (/root/reference/probaforms/models/nflow.py:15-145):
``InvertibleLayer`` is the protocol (``f`` forward + log-det, ``g`` inverse) and
``NormalizingFlow`` chains layers under a prior.  Here a flow made of RealNVP coupling
layers is not evaluated layer by layer: the whole stack runs as one fused HIP kernel
(probaforms_amd/csrc) through ``FlowEngine``.
"""
import math
import os
import threading

import torch
import torch.nn as nn

from .. import _hip
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
    which lets the kernels fold the prior term into the forward pass.

    ``host_rng=False`` (RealNVP(prior_rng='device')) is the build's throughput option: a counter-based stream
    (include/rnvp_hip.h rnvp_prior_normal) in which z[row][j] depends only on (seed, GLOBAL row, j).  One 63-bit
    seed per draw comes from the global CPU generator (``torch.manual_seed`` still controls it); chunks and
    ranks that pass their first global row reproduce the one-shot draw bit for bit, and the flow's inverse
    kernel makes the draws in registers (rnvp_sample).  Not the reference's random stream."""

    def __init__(self, var_size, device, host_rng=True):
        self.var_size = int(var_size)
        self.device = torch.device(device)
        self.host_rng = host_rng
        self.loc = torch.zeros(var_size)
        self.covariance_matrix = torch.eye(var_size)

    def log_prob(self, z):
        return -0.5 * ((z * z).sum(-1) + self.var_size * math.log(2.0 * math.pi))

    @staticmethod
    def next_seed():
        """seed of one counter-based draw: a single int64 from the global CPU generator"""
        return int(torch.empty((), dtype=torch.int64).random_().item())

    def sample(self, sample_shape=(), *, seed=None, row_offset=0):
        shape = tuple(sample_shape) + (self.var_size,)
        if not self.host_rng:
            require_hip(self.device)
            n = 1
            for v in tuple(sample_shape):
                n *= int(v)
            z = torch.empty((n, self.var_size), dtype=torch.float32, device=self.device)
            _hip.prior_normal(self.next_seed() if seed is None else seed, row_offset, n, self.var_size, z)
            return z.reshape(shape)
        n = 1
        for v in shape:
            n *= int(v)
        if n >= 16 and HostStreamOnDevice.usable(self.device):
            return HostStreamOnDevice(self.device).randn(shape)   # the same values, drawn on the device
        return torch.randn(shape).to(self.device)          # host generator: the reference's CPU stream


class HostStreamOnDevice:
    """`torch.randn` of a CPU generator, drawn ON THE DEVICE with the same bits (rnvp_prior_normal_torch_cpu): the reference's
    prior stream (nflow.py:141) without the host's serial 2 ns per number.

    A draw copies the generator's Mersenne-Twister state to the device (2.5 KB), lets the library walk it there (one workgroup
    for the twister, all of them for torch's 16-element Box-Muller blocks), and writes the advanced state back into the
    generator afterwards -- the generator ends exactly where `torch.randn` would have left it, so everything drawn later
    (by this class or by torch) is unchanged.  Several draws can be chained on the device (`begin` / `draw` / `end`) with one
    state round trip.

    The library restates what THIS torch build's CPU kernel computes (mt19937, 24-bit uniforms, 16-element Box-Muller blocks on
    avx_mathfun.h's polynomials with the multiply-adds its compiler contracted); another build could differ, so
    `usable(device)` draws 4117 + 32 + 100 003 numbers both ways from a scratch generator once per process and device and compares
    values and final states bit for bit; on any difference the callers keep the host draw.  That check guards what depends on
    the torch build (the uniform conversion, the Box-Muller arithmetic, the tail rule, the state layout) and one jump-ahead unit;
    the twister's larger jump units and multi-round draws (>= 20M numbers) do not depend on torch and are pinned by the GPU
    test suite against torch.randn instead (tests/test_prior_torch.py: up to 25M numbers)."""

    _ok = {}
    _lock = threading.Lock()

    @staticmethod
    def _unpack(gen):
        import numpy as np
        st = gen.get_state().numpy()
        if st.size != 5056:
            raise RuntimeError("unknown CPU generator state layout (%d bytes)" % st.size)
        left = int(st[8:12].view(np.int32)[0]); nxt = int(st[16:24].view(np.uint64)[0])
        words = st[24:24 + 624 * 8].view(np.uint64).astype(np.uint32)
        pos = 624 if left == 1 else nxt
        if not (0 <= pos <= 624) or (left != 1 and left + nxt != 625):
            raise RuntimeError("inconsistent CPU generator state (left %d, next %d)" % (left, nxt))
        return st, np.concatenate([words, np.array([pos], np.uint32)]).view(np.int32)

    @staticmethod
    def _pack(gen, st, mt):
        import numpy as np
        mt = mt.view(np.uint32)
        pos = int(mt[624])
        new = st.copy()
        new[8:12] = np.array([1 if pos == 624 else 625 - pos], np.int32).view(np.uint8)
        new[16:24] = np.array([pos], np.uint64).view(np.uint8)
        new[24:24 + 624 * 8] = mt[:624].astype(np.uint64).view(np.uint8)
        gen.set_state(torch.from_numpy(new))

    def __init__(self, device, generator=None):
        self.device = torch.device(device)
        self.gen = torch.default_generator if generator is None else generator
        self._st = None

    def begin(self):
        st, mt = self._unpack(self.gen)
        self._st = st
        self._mt = torch.from_numpy(mt.copy()).to(self.device)
        self._tail = torch.empty(16, dtype=torch.float32, device=self.device)
        self._ws = torch.empty(_hip.prior_torch_workspace_bytes(), dtype=torch.uint8, device=self.device)
        return self

    def draw(self, out):
        """out (contiguous float32 device tensor, numel >= 16) <- the generator's next out.numel() normals"""
        _hip.prior_normal_torch_cpu(self._mt, out.numel(), out, self._tail, self._ws)
        return out

    def end(self):
        """hand the advanced state back to the generator (waits for the draws)"""
        self._pack(self.gen, self._st, self._mt.cpu().numpy())
        self._st = None

    def randn(self, shape):
        z = torch.empty(shape, dtype=torch.float32, device=self.device)
        self.begin()
        try:
            self.draw(z)
        finally:
            self.end()          # the window is this one draw; an error leaves the generator advanced by what was enqueued
        return z

    @classmethod
    def usable(cls, device):
        device = torch.device(device)
        if device.type != "cuda" or os.environ.get("RNVP_HOST_PRIOR_ON_DEVICE", "1") == "0":
            return False
        key = device.index
        with cls._lock:
            if key not in cls._ok:
                ok = False
                try:
                    g1 = torch.Generator(); g1.manual_seed(20240717)
                    torch.rand(100, generator=g1)                       # start inside a block
                    g2 = torch.Generator(); g2.set_state(g1.get_state())
                    ref = torch.randn(4117, generator=g1)               # not a multiple of 16: the redrawn tail, 6 twists
                    got = cls(device, g2).randn((4117,)).cpu()
                    ok = bool(torch.equal(ref, got)) and bool(torch.equal(g1.get_state(), g2.get_state()))
                    if ok:                                              # and a second call continues the stream
                        ok = bool(torch.equal(torch.randn(32, generator=g1), cls(device, g2).randn((32,)).cpu()))
                    if ok:                                              # a draw long enough for the jump-ahead segments (64-block unit)
                        ok = (bool(torch.equal(torch.randn(100_003, generator=g1), cls(device, g2).randn((100_003,)).cpu()))
                              and bool(torch.equal(g1.get_state(), g2.get_state())))
                except Exception:
                    ok = False
                cls._ok[key] = ok
            return cls._ok[key]


def row_chunks(n, rows):
    """[(lo, m)] covering n rows in chunks of `rows` (a multiple of 16); a tail shorter than 16 rows is
    merged into the chunk before it.  torch's CPU randn fills blocks of 16 values and redraws the LAST
    16 values of a call whose size is not a multiple of 16, so these chunked draws consume the global
    generator exactly like one randn(n, d) for any d."""
    out = [(lo, min(rows, n - lo)) for lo in range(0, n, rows)]
    if len(out) > 1 and out[-1][1] < 16:
        lo, m = out[-2]
        out[-2:] = [(lo, m + out[-1][1])]
    return out


class NormalizingFlow(nn.Module):
    """Layers + prior (nflow.py:71-145).

    log_prob(X, C) -> 0-dim tensor: mean over the batch of [sum_l log_det_l + prior.log_prob(z)]; differentiable like the
                      reference's (`(-nf.log_prob(X, C)).backward()` fills p.grad through the HIP backward, rnvp_backward),
                      unless called under torch.no_grad().
    sample(C)      -> [n, var_size]: prior draw pushed through the layers' inverses, last layer first.
    log_prob_samples(X, C) -> [n] per-row log-density (build-only addition).
    """

    def __init__(self, layers, prior, *, precision=None, small_calls=None):
        super().__init__()
        self.layers = nn.ModuleList(layers)
        self.prior = prior
        # build-only, keyword-only: arithmetic of the s/t nets' first Linear in the forward / inverse kernels --
        # None / 'auto' (library picks per shape), 'f32', 'bx3' (_hip.PRECISIONS; both meet the 1e-5 parity bar)
        self.precision = precision
        # build-only, keyword-only: None / 'invariant' (a row's log_prob / sample never depends on how the rows are split
        # into calls, chunks or ranks) or 'latency' (calls of at most 4096 rows run the tile-split kernels: 2-4x lower
        # latency, last-bit differences against larger calls) -- _hip.SMALL_CALLS
        self.small_calls = small_calls
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
            self._engine_obj = FlowEngine(list(self.layers), dev, precision=self.precision, small_calls=self.small_calls)
        return self._engine_obj

    def _fused_prior(self):
        return isinstance(self.prior, StandardNormalPrior)

    def _layerwise(self):
        """True when the layers cannot run as ONE fused stack: the reference's container accepts any list of
        InvertibleLayer (nflow.py:85-88) -- coupling layers of differing hidden widths / activations, or user-defined
        layers.  Such flows are evaluated layer by layer through each layer's own f / g (RealNVPLayer.f / .g are the
        HIP kernels with L = 1), exactly the loops of nflow.py:109-114,142-143."""
        ls = list(self.layers)
        if not ls:
            return True
        l0 = ls[0]
        key = lambda l: (getattr(l, "var_size", None), getattr(l, "cond_size", None), tuple(getattr(l, "hidden", ())),
                         getattr(l, "activation", None), hasattr(l, "nn_t") and hasattr(l, "nn_s") and hasattr(l, "mask"))
        return any(key(l) != key(l0) for l in ls) or not key(l0)[4]

    def _layerwise_forward(self, X, C):
        dev = X.device if torch.is_tensor(X) and X.is_cuda else DEVICE
        x = torch.as_tensor(X, dtype=torch.float32).to(dev)
        c = None if C is None else torch.as_tensor(C, dtype=torch.float32).to(dev)
        log_det = torch.zeros(x.shape[0], device=x.device)
        for layer in self.layers:
            x, ld = layer.f(x, c)
            log_det = log_det + ld
        return log_det + self.prior.log_prob(x)

    def _on_device(self, t, eng):
        if t is None:
            return None
        return torch.as_tensor(t, dtype=torch.float32).to(eng.device).contiguous()

    def _wants_graph(self, X, C=None):
        """as in the reference, log_prob / log_prob_samples carry an autograd graph whenever one can be recorded: grad mode
        on and a parameter (or X, or C) requiring grad.  Under torch.no_grad() -- or with frozen parameters -- the fused
        no-graph kernel runs instead (prior folded in, nothing saved for a backward)."""
        if not torch.is_grad_enabled():
            return False
        return ((torch.is_tensor(X) and X.requires_grad) or (torch.is_tensor(C) and C.requires_grad)
                or any(p.requires_grad for p in self.parameters()))

    def _log_prob_graph(self, X, C):
        """per-row log-density with a graph: (z, logdet) from the FlowFunction node (forward = the fused stack, backward
        = rnvp_backward), the prior's log_prob from torch -- any differentiable prior object works (nflow.py:115)"""
        eng = self.engine()
        Xd = X.to(eng.device, torch.float32) if torch.is_tensor(X) else self._on_device(X, eng)
        Cd = C.to(eng.device, torch.float32).contiguous() if torch.is_tensor(C) else self._on_device(C, eng)     # (keeps C's graph)
        z, ld = eng.forward_autograd(Xd.contiguous(), Cd)
        return ld + self.prior.log_prob(z)

    # -- reference API -------------------------------------------------------------------------
    def log_prob(self, X, C):
        if self._layerwise():
            return self._layerwise_forward(X, C).mean()
        if self._wants_graph(X, C):
            return self._log_prob_graph(X, C).mean()
        eng = self.engine()
        X, C = self._on_device(X, eng), self._on_device(C, eng)
        if self._fused_prior():
            _, _, _, tot = eng.forward(X, C, want_z=False, want_logp=False, want_sum=True)
            return (tot / X.shape[0]).reshape(())
        z, ld, _, _ = eng.forward(X, C, want_z=True, want_logdet=True, want_logp=False)
        return (ld + self.prior.log_prob(z)).mean()

    def log_prob_samples(self, X, C=None):
        if self._layerwise():
            return self._layerwise_forward(X, C)
        if self._wants_graph(X, C):
            return self._log_prob_graph(X, C)
        eng = self.engine()
        X, C = self._on_device(X, eng), self._on_device(C, eng)
        if self._fused_prior():
            return eng.forward(X, C, want_z=False, want_logp=True)[2]
        z, ld, _, _ = eng.forward(X, C, want_z=True, want_logdet=True, want_logp=False)
        return ld + self.prior.log_prob(z)

    def sample(self, C):
        if self._layerwise():
            n = C if type(C) == type(1) else len(C)
            c = None if type(C) == type(1) else torch.as_tensor(C, dtype=torch.float32).to(DEVICE)
            x = torch.as_tensor(self.prior.sample((n,)), dtype=torch.float32).to(DEVICE)
            for layer in self.layers[::-1]:
                x = layer.g(x, c)
            return x
        eng = self.engine()
        if type(C) == type(1):            # python int only, as nflow.py:135 (np.int64 is not an int)
            n, C = C, None
        else:
            n = len(C)
            C = self._on_device(C, eng)
        # As in the reference (nflow.py:141-145), the sample carries an autograd graph whenever one can be recorded -- grad mode
        # on and a parameter (or C) requiring grad: x = g(z, c) is then ONE node whose backward is rnvp_inverse_backward
        # (reverse-KL / sample-based losses).  RealNVP.sample detaches, as realnvp.py:280 does, and takes the paths below.
        graph = eng.wants_graph(C)
        if self._fused_prior() and not self.prior.host_rng and not graph:
            return eng.sample(n, C, self.prior.next_seed())       # prior drawn inside the inverse kernel
        z = self.prior.sample((n,))
        z = torch.as_tensor(z, dtype=torch.float32).to(eng.device).contiguous()
        if graph:
            return eng.inverse_autograd(z, C)
        return eng.inverse(z, C, out=z)

    # -- host staging -------------------------------------------------
    # one chunk of output rows: at least PIPELINE_CHUNK_BYTES and at least PIPELINE_MIN_ROWS rows (a launch of fewer
    # rows leaves CUs idle: 256 rows per workgroup).  Measured at API level, C2 sample(1M): 8 MB chunks 3.1 ms (device
    # prior) / 33 ms (host prior) against 9.4 / 60 ms one-shot with a pageable download.
    PIPELINE_CHUNK_BYTES = 8 << 20
    PIPELINE_MIN_ROWS = 131072
    ONE_SHOT_UPLOAD_BYTES = 2 << 30        # conditions larger than this (at their source width) are staged chunk by chunk
    _UPLOAD_DTYPES = frozenset(__import__("numpy").dtype(t) for t in
                               ("float16", "float32", "float64", "int8", "int16", "int32", "int64", "uint8", "bool"))

    def pipelined_rows(self, n):
        """rows per chunk (a multiple of 16, see row_chunks) if sample_to_host() would pipeline n rows, else 0"""
        if not self._fused_prior() or self._layerwise():
            return 0
        rows = max(16, self.PIPELINE_MIN_ROWS, self.PIPELINE_CHUNK_BYTES // (4 * self.prior.var_size)) // 16 * 16
        return rows if n > 2 * rows else 0

    def sample_to_host(self, C):
        """``sample(C).cpu().numpy()`` (realnvp.py:279-282) as a three-stage pipeline over row chunks:
        prior draw on the host generator (or on the device) + H2D of z and C | inverse kernel | D2H
        into a pinned result, each on its own HIP stream, double buffered.  C: python int, numpy array
        or tensor.  Returns a float32 numpy array [n, var_size] (backed by pinned host memory); the
        values are those of the one-shot path."""
        import numpy as np
        eng = self.engine()
        dev = eng.device
        if type(C) == type(1):
            n, Cn, Cd = C, None, None
        else:
            n = len(C)
            on_dev = torch.is_tensor(C) and C.device.type == "cuda"
            Cd = C.to(dev, torch.float32).contiguous() if on_dev else None
            Cn = None if on_dev else (C.detach().numpy() if torch.is_tensor(C) else np.asarray(C))
        rows = self.pipelined_rows(n)
        # (only dtypes torch.from_numpy takes -- numpy's longdouble / float128, uint16..64 on older builds and structured types keep
        # the chunked numpy staging below, whose casting='unsafe' copy accepts them all -- and only while the whole array at its
        # SOURCE width is a modest share of the device: a very large draw stays on bounded per-chunk staging)
        one_shot = (rows and Cn is not None and Cn.flags.c_contiguous and Cn.flags.writeable
                    and Cn.dtype in self._UPLOAD_DTYPES and Cn.nbytes <= self.ONE_SHOT_UPLOAD_BYTES)
        # one_shot: ONE pageable upload through the runtime's own staging (tens of GB/s on these hosts) and the float32 cast ON THE DEVICE
        # (the same round-to-nearest as the host cast; what RealNVP.fit does with X and C) instead of a single-threaded numpy
        # copy / cast into pinned memory per chunk -- that copy, not the GPU, bounded the call (16 MB of float32 conditions:
        # 1.6 ms of host time against 1.1 ms of kernels for sample(1M) at C2; float64 input twice that).  The upload blocks the
        # HOST for about a millisecond, so it is made further down, AFTER the first window of the prior draw has been enqueued:
        # the draw (0.57 ms of GPU time on its own stream) then runs under it instead of behind it.
        if rows == 0:
            if Cn is not None:
                Cd = self._on_device(torch.from_numpy(np.ascontiguousarray(Cn)), eng)
            with torch.no_grad():       # the result is detached on this very line (realnvp.py:280): record no graph, keep the fused-prior path
                return self.sample(n if Cd is None else Cd).cpu().numpy()
        d = self.prior.var_size
        cdim = 0 if type(C) == type(1) else C.shape[1]
        host_rng = self.prior.host_rng
        seed = None if host_rng else self.prior.next_seed()     # one counter-based stream for all chunks
        NB, cap = 3, rows + 15
        out = torch.empty((n, d), dtype=torch.float32, pin_memory=True)
        zdev = [torch.empty((cap, d), dtype=torch.float32, device=dev) for _ in range(NB)]
        # the reference's stream: drawn on the device with the host generator's bits where that is validated (HostStreamOnDevice),
        # on its own stream, a WINDOW of chunks per draw (up to 32M numbers: the twister's workgroups share one stream through
        # jump-ahead, so one large draw costs a fraction of many small ones); else on the host, chunk by chunk
        hs = HostStreamOnDevice(dev).begin() if (host_rng and HostStreamOnDevice.usable(dev)) else None
        chunks = row_chunks(n, rows)
        if hs is not None:
            cpw = max(1, (32 << 20) // (rows * d))                           # chunks per window
            nwin = (len(chunks) + cpw - 1) // cpw
            win = [(chunks[w * cpw][0], sum(m for _, m in chunks[w * cpw:(w + 1) * cpw])) for w in range(nwin)]
            zwin = [torch.empty((max(r for _, r in win), d), dtype=torch.float32, device=dev) for _ in range(min(2, nwin))]
            ev_win = [torch.cuda.Event() for _ in zwin]
            ev_used = [None for _ in zwin]                                    # the window buffer's last reader (an inverse kernel)
        zpin = [torch.empty((cap, d), dtype=torch.float32, pin_memory=True) for _ in range(NB)] if (host_rng and hs is None) else None
        stage_c = cdim > 0 and Cn is not None and not one_shot
        cdev = [torch.empty((cap, cdim), dtype=torch.float32, device=dev) for _ in range(NB)] if stage_c else None
        cpin = [torch.empty((cap, cdim), dtype=torch.float32, pin_memory=True) for _ in range(NB)] if stage_c else None
        cpin_np = [t.numpy() for t in cpin] if stage_c else None
        cur = torch.cuda.current_stream(dev)
        h2d, d2h = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        ev_in = [torch.cuda.Event() for _ in range(NB)]
        ev_k = [torch.cuda.Event() for _ in range(NB)]
        ev_out = [torch.cuda.Event() for _ in range(NB)]
        gen = torch.cuda.Stream(dev) if hs is not None else None
        h2d.wait_stream(cur)

        def draw_window(w):             # the global CPU generator's next normals for window w, made on the device
            b = w % len(zwin)
            with torch.cuda.stream(gen):
                if ev_used[b] is not None:
                    gen.wait_event(ev_used[b])
                hs.draw(zwin[b][:win[w][1]])
                ev_win[b].record(gen)

        # The window from begin() to end() holds a COPY of the global CPU generator's state: end() writes the advanced state back.
        # Whatever happens in between, the numbers the enqueued draws consumed stay consumed (try / finally); draws another thread
        # makes on torch.default_generator inside the window are overwritten by end() -- torch.randn holds the generator only per
        # call, so the reference has no such guarantee across a sample() either.
        try:
            if gen is not None:
                gen.wait_stream(cur)
                draw_window(0)
            if one_shot:
                Cd = torch.from_numpy(Cn).to(dev).to(torch.float32)
                Cn = None
            for k, (lo, m) in enumerate(chunks):
                i = k % NB
                if k >= NB:
                    ev_out[i].synchronize()                 # buffer set i is free again (its D2H has landed)
                if hs is not None:
                    w = k // cpw
                    if k % cpw == 0:
                        cur.wait_event(ev_win[w % len(zwin)])
                        if w + 1 < nwin:
                            draw_window(w + 1)              # overlaps this window's inverse kernels and downloads
                elif host_rng:
                    torch.randn((m, d), out=zpin[i][:m])    # global CPU generator: the reference's stream
                if stage_c:
                    # numpy's single-threaded copy/cast (float32 rounding = torch's): torch's CPU copy_ wakes
                    # its whole thread pool per call, measured 5 ms per 8 MB chunk on a 256-core host
                    np.copyto(cpin_np[i][:m], Cn[lo:lo + m], casting="unsafe")
                with torch.cuda.stream(h2d):
                    if zpin is not None:
                        zdev[i][:m].copy_(zpin[i][:m], non_blocking=True)
                    if stage_c:
                        cdev[i][:m].copy_(cpin[i][:m], non_blocking=True)
                    ev_in[i].record(h2d)
                cur.wait_event(ev_in[i])
                cc = None if cdim == 0 else (cdev[i][:m] if stage_c else Cd[lo:lo + m])
                if hs is not None:
                    w = k // cpw
                    eng.inverse(zwin[w % len(zwin)][lo - win[w][0]:lo - win[w][0] + m], cc, out=zdev[i][:m])
                    if k % cpw == cpw - 1 or k + 1 == len(chunks):
                        ev_used[w % len(zwin)] = torch.cuda.Event()
                        ev_used[w % len(zwin)].record(cur)
                elif host_rng:
                    eng.inverse(zdev[i][:m], cc, out=zdev[i][:m])
                else:
                    eng.sample(m, cc, seed, row_offset=lo, out=zdev[i][:m])   # chunk lo..lo+m of the one-shot draw
                ev_k[i].record(cur)
                d2h.wait_event(ev_k[i])
                with torch.cuda.stream(d2h):
                    out[lo:lo + m].copy_(zdev[i][:m], non_blocking=True)
                    ev_out[i].record(d2h)
            d2h.synchronize()
            cur.wait_stream(d2h)
        finally:
            if hs is not None:
                cur.wait_stream(gen)
                hs.end()                                # the generator ends where the host draws would have left it
        return out.numpy()
