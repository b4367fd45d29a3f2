"""ctypes/numpy front-end of oracle/rnvp_oracle.c -- TEST INFRASTRUCTURE ONLY.

Every method is a thin call into the plain-C restatement of the reference
(/root/reference/probaforms/models/realnvp.py:47-129, nflow.py:90-145,
realnvp.py:205-207,246-251); see the header of rnvp_oracle.c for the citation
of each function.  `Oracle(precision=32)` computes in float like the reference,
`Oracle(precision=64)` is the double-precision referee used to show that the
HIP path is no further from the truth than the reference itself.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")
MAX_HIDDEN = 8


class Shape(C.Structure):
    _fields_ = [("L", C.c_int32), ("d", C.c_int32), ("c", C.c_int32),
                ("n_hidden", C.c_int32), ("hidden", C.c_int32 * MAX_HIDDEN),
                ("act", C.c_int32)]

    @classmethod
    def make(cls, L, d, c, hidden=(10,), activation="tanh"):
        hidden = tuple(int(h) for h in hidden)
        assert 1 <= len(hidden) <= MAX_HIDDEN
        s = cls()
        s.L, s.d, s.c, s.n_hidden = int(L), int(d), int(c), len(hidden)
        for i, h in enumerate(hidden):
            s.hidden[i] = h
        s.act = 0 if activation == "tanh" else 1          # realnvp.py:32-37
        return s


def build(force=False):
    """Compile the oracle with gcc (make -C oracle). Building the checker is not using it."""
    libs = [os.path.join(_BUILD, "librnvp_oracle%d.so" % b) for b in (32, 64)]
    libs.append(os.path.join(_BUILD, "libprior_torch_oracle.so"))
    libs.append(os.path.join(_BUILD, "librandperm_torch_oracle.so"))
    srcs = [os.path.join(_HERE, f) for f in ("rnvp_oracle.c", "cvae_oracle.c", "prior_torch_oracle.c", "randperm_torch_oracle.c")]
    stale = force or any(not os.path.exists(p) or os.path.getmtime(p) < max(os.path.getmtime(f) for f in srcs)
                         for p in libs)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return libs


def default_masks(L, d):
    """mask[l][j] = (j + l) % 2 -- realnvp.py:199."""
    return ((np.arange(d)[None, :] + np.arange(L)[:, None]) % 2).astype(np.uint8)


def flat_from_state_dict(sd, L, n_linear, prefix="layers."):
    """Concatenate a reference `nf.state_dict()` in nf.parameters() order."""
    parts = []
    for l in range(L):
        for net in ("nn_t", "nn_s"):
            for k in range(n_linear):
                for nm in ("weight", "bias"):
                    parts.append(np.asarray(sd["%s%d.%s.%d.%s" % (prefix, l, net, 2 * k, nm)],
                                            dtype=np.float64).ravel())
    return np.concatenate(parts)


class Oracle:
    def __init__(self, precision=32):
        assert precision in (32, 64)
        build()
        self.lib = C.CDLL(os.path.join(_BUILD, "librnvp_oracle%d.so" % precision))
        self.dtype = np.float32 if precision == 32 else np.float64
        assert self.lib.rnvp_oracle_real_bytes() == np.dtype(self.dtype).itemsize
        self.lib.rnvp_oracle_param_count.restype = C.c_size_t

    # -- helpers ---------------------------------------------------------------
    def _a(self, x, shape=None):
        a = np.ascontiguousarray(x, dtype=self.dtype)
        if shape is not None:
            assert a.shape == tuple(shape), (a.shape, shape)
        return a

    @staticmethod
    def _p(a):
        return None if a is None else a.ctypes.data_as(C.c_void_p)

    def param_count(self, shape):
        return int(self.lib.rnvp_oracle_param_count(C.byref(shape)))

    def _cond(self, shape, c, B):
        if shape.c == 0:
            return None
        return self._a(c, (B, shape.c))

    def _masks(self, shape, masks):
        if masks is None:
            masks = default_masks(shape.L, shape.d)
        m = np.ascontiguousarray(masks, dtype=np.uint8)
        assert m.shape == (shape.L, shape.d)
        return m

    # -- RealNVPLayer.f / .g ----------------------------------------------------
    def layer_f(self, shape, p_layer, mask, x, c=None):
        B = x.shape[0]
        x = self._a(x, (B, shape.d)); cc = self._cond(shape, c, B)
        p = self._a(p_layer); m = np.ascontiguousarray(mask, dtype=np.uint8)
        y = np.empty_like(x); ld = np.empty(B, self.dtype)
        self.lib.rnvp_oracle_layer_f(C.byref(shape), self._p(p), self._p(m), self._p(x), self._p(cc),
                                     C.c_int64(B), self._p(y), self._p(ld))
        return y, ld

    def layer_g(self, shape, p_layer, mask, x, c=None):
        B = x.shape[0]
        x = self._a(x, (B, shape.d)); cc = self._cond(shape, c, B)
        p = self._a(p_layer); m = np.ascontiguousarray(mask, dtype=np.uint8)
        y = np.empty_like(x)
        self.lib.rnvp_oracle_layer_g(C.byref(shape), self._p(p), self._p(m), self._p(x), self._p(cc),
                                     C.c_int64(B), self._p(y))
        return y

    # -- NormalizingFlow.log_prob / .sample -------------------------------------
    def log_prob(self, shape, params, x, c=None, masks=None):
        """returns (z [B,d], logp [B], mean)"""
        B = x.shape[0]
        x = self._a(x, (B, shape.d)); cc = self._cond(shape, c, B)
        p = self._a(params, (self.param_count(shape),)); m = self._masks(shape, masks)
        z = np.empty_like(x); lp = np.empty(B, self.dtype); mean = np.zeros(1, self.dtype)
        self.lib.rnvp_oracle_log_prob(C.byref(shape), self._p(p), self._p(m), self._p(x), self._p(cc),
                                      C.c_int64(B), self._p(z), self._p(lp), self._p(mean))
        return z, lp, mean[0]

    def sample(self, shape, params, z, c=None, masks=None):
        n = z.shape[0]
        z = self._a(z, (n, shape.d)); cc = self._cond(shape, c, n)
        p = self._a(params, (self.param_count(shape),)); m = self._masks(shape, masks)
        x = np.empty_like(z)
        self.lib.rnvp_oracle_sample(C.byref(shape), self._p(p), self._p(m), self._p(z), self._p(cc),
                                    C.c_int64(n), self._p(x))
        return x

    # -- the build's counter-based prior (include/rnvp_hip.h rnvp_prior_normal) -------
    def philox4x32_10(self, ctr, key):
        c = (C.c_uint32 * 4)(*[int(v) for v in ctr]); k = (C.c_uint32 * 2)(*[int(v) for v in key])
        out = (C.c_uint32 * 4)()
        self.lib.rnvp_oracle_philox4x32_10(c, k, out)
        return [int(v) for v in out]

    def prior_normal(self, seed, row0, n, d):
        z = np.empty((n, d), self.dtype)
        self.lib.rnvp_oracle_prior_normal(C.c_uint64(int(seed)), C.c_int64(int(row0)), C.c_int64(int(n)),
                                          C.c_int32(int(d)), self._p(z))
        return z

    # -- loss + gradient, Adam ---------------------------------------------------
    def loss_grad(self, shape, params, x, c=None, masks=None, inv_B=None):
        """returns (loss, grad [P]) for loss = -mean log_prob (realnvp.py:246)."""
        B = x.shape[0]
        x = self._a(x, (B, shape.d)); cc = self._cond(shape, c, B)
        p = self._a(params, (self.param_count(shape),)); m = self._masks(shape, masks)
        g = np.empty_like(p); loss = np.zeros(1, self.dtype)
        inv = 1.0 / B if inv_B is None else float(inv_B)
        self.lib.rnvp_oracle_loss_grad(C.byref(shape), self._p(p), self._p(m), self._p(x), self._p(cc),
                                       C.c_int64(B), C.c_double(inv), self._p(g), self._p(loss))
        return loss[0], g

    def adam(self, p, g, m, v, step, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        """in-place on p, m, v (arrays of this oracle's dtype)."""
        for a in (p, m, v):
            assert a.dtype == self.dtype and a.flags.c_contiguous
        g = self._a(g, p.shape)
        self.lib.rnvp_oracle_adam(self._p(p), self._p(g), self._p(m), self._p(v), C.c_int64(p.size),
                                  C.c_double(lr), C.c_double(betas[0]), C.c_double(betas[1]),
                                  C.c_double(eps), C.c_double(weight_decay), C.c_int64(step))


# ---------------------------------------------------------------------------------------------
# CVAE (oracle/cvae_oracle.c; /root/reference/probaforms/models/cvae.py)
# ---------------------------------------------------------------------------------------------
class CvaeShape(C.Structure):
    _fields_ = [("d", C.c_int32), ("c", C.c_int32), ("lat", C.c_int32), ("n_hidden", C.c_int32),
                ("hidden", C.c_int32 * MAX_HIDDEN), ("act", C.c_int32)]

    @classmethod
    def make(cls, d, c, lat, hidden=(10,), activation="tanh"):
        s = cls()
        s.d, s.c, s.lat, s.n_hidden = int(d), int(c), int(lat), len(hidden)
        for i, h in enumerate(hidden):
            s.hidden[i] = int(h)
        s.act = 0 if activation == "tanh" else 1
        return s


def cvae_flat_from_state(enc_sd, dec_sd, n_hidden):
    """reference state_dicts of Encoder / Decoder -> the oracle's flat order (heads as one Linear)"""
    g = lambda sd, k: np.asarray(sd[k], dtype=np.float64).ravel()
    parts = []
    for k in range(n_hidden):
        parts += [g(enc_sd, "model.%d.weight" % (2 * k)), g(enc_sd, "model.%d.bias" % (2 * k))]
    parts += [g(enc_sd, "mu.weight"), g(enc_sd, "log_sigma.weight"), g(enc_sd, "mu.bias"), g(enc_sd, "log_sigma.bias")]
    for k in range(n_hidden + 1):
        parts += [g(dec_sd, "model.%d.weight" % (2 * k)), g(dec_sd, "model.%d.bias" % (2 * k))]
    return np.concatenate(parts)


class CvaeOracle:
    def __init__(self, precision=32):
        build()
        self.lib = C.CDLL(os.path.join(_BUILD, "librnvp_oracle%d.so" % precision))
        self.dtype = np.float32 if precision == 32 else np.float64
        self.lib.cvae_oracle_param_count.restype = C.c_size_t

    def _a(self, x):
        return None if x is None else np.ascontiguousarray(x, dtype=self.dtype)

    @staticmethod
    def _p(a):
        return None if a is None else a.ctypes.data_as(C.c_void_p)

    def param_count(self, s):
        return int(self.lib.cvae_oracle_param_count(C.byref(s)))

    def loss_grad(self, s, params, x, c, eps, kl_weight, inv_B=None, want_grad=True):
        B = x.shape[0]
        x, c, eps, p = self._a(x), self._a(c), self._a(eps), self._a(params)
        g = np.empty_like(p) if want_grad else None
        loss = np.zeros(1, self.dtype)
        self.lib.cvae_oracle_loss_grad(C.byref(s), self._p(p), self._p(x), self._p(c), self._p(eps), C.c_int64(B),
                                       C.c_double(1.0 / B if inv_B is None else inv_B), C.c_double(kl_weight),
                                       self._p(g), self._p(loss))
        return loss[0], g

    def decode(self, s, params, z, c):
        n = z.shape[0]
        z, c, p = self._a(z), self._a(c), self._a(params)
        x = np.empty((n, s.d), self.dtype)
        self.lib.cvae_oracle_decode(C.byref(s), self._p(p), self._p(z), self._p(c), C.c_int64(n), self._p(x))
        return x

    def encode(self, s, params, x, c):
        n = x.shape[0]
        x, c, p = self._a(x), self._a(c), self._a(params)
        mu = np.empty((n, s.lat), self.dtype); ls = np.empty((n, s.lat), self.dtype)
        self.lib.cvae_oracle_encode(C.byref(s), self._p(p), self._p(x), self._p(c), C.c_int64(n), self._p(mu), self._p(ls))
        return mu, ls
