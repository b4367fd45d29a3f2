"""ctypes binding of librnvp_hip.so (C ABI: include/rnvp_hip.h).

The shared library is built in-tree by `make -C probaforms_amd/csrc` (see
__graft_entry__.build).  There is NO fallback: if the library is missing, or a
tensor is not on a HIP device, the call raises.  PyTorch is used only to own
device memory and streams (tensor.data_ptr(), torch.cuda.current_stream()).
"""
import ctypes as C
import os
import threading

import torch

MAX_HIDDEN = 8
ABI_VERSION = 600          # rnvp_version() of the library this binding matches (include/rnvp_hip.h RNVP_HIP_VERSION)
OP_FORWARD, OP_INVERSE, OP_TRAIN = 0, 1, 2
PROFILE_TRAIN, PROFILE_FORWARD, PROFILE_INVERSE = 0, 1, 2
PATH_GENERIC, PATH_MFMA, PATH_LMM = 0, 1, 2
# arithmetic of the first Linear of the s/t nets in the forward / inverse / sampling kernels (rnvp_shape.precision):
# 'f32' = f32-input MFMA; 'bx3' = three-term bf16 split, six products (float32-level accuracy, see rnvp_bx3.h)
# 'auto' lets the library take the faster one for the shape (bx3 for d > 16, cdim > 4 or one hidden layer of more than 64 units)
PRECISIONS = {"auto": 0, "f32": 1, "bx3": 2}
DEFAULT_PRECISION = os.environ.get("RNVP_PRECISION", "auto")

_HERE = os.path.dirname(os.path.abspath(__file__))
# RNVP_HIP_LIB lets a developer A/B an experimental build of the same C ABI (still a HIP library:
# there is no non-HIP implementation to point it at).
LIB_PATH = os.environ.get("RNVP_HIP_LIB") or os.path.join(_HERE, "csrc", "librnvp_hip.so")


# rnvp_shape.small_calls: forward / inverse / sampling calls of at most 4096 rows -- 'invariant' (default): a row's result
# never depends on how the rows are split into calls; 'latency': tile-split kernels, 2-4x lower latency, last-bit differences
SMALL_CALLS = {"invariant": 0, "latency": 1}
# rnvp_shape.family (per call; test / measurement aid): kernels for shapes outside the register-chained MFMA path --
# 'auto' / 'lmm': the any-shape MFMA kernels whenever their LDS image fits; 'valu': always the one-thread-per-row kernels
FAMILIES = {"auto": 0, "valu": 1, "lmm": 2, "lmm16": 3, "lmm64": 4}


class RnvpShape(C.Structure):
    """mirror of `rnvp_shape` (include/rnvp_hip.h)"""
    _fields_ = [("L", C.c_int32), ("d", C.c_int32), ("c", C.c_int32),
                ("n_hidden", C.c_int32), ("hidden", C.c_int32 * MAX_HIDDEN),
                ("act", C.c_int32), ("alt_masks", C.c_int32), ("precision", C.c_int32), ("small_calls", C.c_int32),
                ("family", C.c_int32)]

    @classmethod
    def make(cls, L, d, c, hidden, activation, alt_masks=0, precision=None, small_calls=0, family="auto"):
        hidden = tuple(int(h) for h in hidden)
        if not 1 <= len(hidden) <= MAX_HIDDEN:
            raise ValueError("hidden must have 1..%d entries, got %r" % (MAX_HIDDEN, hidden))
        s = cls()
        s.L, s.d, s.c, s.n_hidden = int(L), int(d), int(c), len(hidden)
        for i, h in enumerate(hidden):
            s.hidden[i] = h
        s.act = 0 if activation == "tanh" else 1     # anything else is ReLU, realnvp.py:32-37
        s.alt_masks = int(alt_masks)
        s.precision = PRECISIONS[DEFAULT_PRECISION if precision is None else precision]
        s.small_calls = int(small_calls)             # SMALL_INVARIANT (0) / SMALL_LATENCY (1)
        s.family = FAMILIES[family]                  # which kernels serve a shape outside the register-chained MFMA path
        return s

    @staticmethod
    def classify_masks(masks):
        """0 arbitrary; 1 if masks[l][j] == (j+l)%2 (realnvp.py:199); 2 if == (j+l+1)%2"""
        import numpy as np
        m = np.asarray(masks).astype(np.int64)
        if m.ndim == 1:
            m = m[None, :]
        L, d = m.shape
        base = (np.arange(d)[None, :] + np.arange(L)[:, None]) % 2
        if np.array_equal(m, base):
            return 1
        if np.array_equal(m, 1 - base):
            return 2
        return 0

    def key(self):
        return (self.L, self.d, self.c, tuple(self.hidden[:self.n_hidden]), self.act)


VARIANTS = {0: "none", 1: "rowpar", 2: "netsplit", 3: "wide", 4: "tilesplit", 5: "bx3_staged", 6: "bx3_direct", 7: "lmm",
            8: "valu", 9: "resident"}


class RnvpDispatch(C.Structure):
    """mirror of `rnvp_dispatch` (include/rnvp_hip.h): what the calling thread's last call of a kind launched"""
    _fields_ = [("variant", C.c_int32), ("row_tiles", C.c_int32), ("waves", C.c_int32), ("grid", C.c_int32),
                ("gemm1_fwd", C.c_int32), ("launches", C.c_int32), ("rows", C.c_int64), ("kernel", C.c_char * 48)]


class CvaeShape(C.Structure):
    """mirror of `cvae_shape` (include/cvae_hip.h)"""
    _fields_ = [("d", C.c_int32), ("c", C.c_int32), ("lat", C.c_int32), ("n_hidden", C.c_int32),
                ("hidden", C.c_int32 * MAX_HIDDEN), ("act", C.c_int32), ("family", C.c_int32)]

    @classmethod
    def make(cls, d, c, lat, hidden, activation, family="auto"):
        hidden = tuple(int(h) for h in hidden)
        if not 1 <= len(hidden) <= MAX_HIDDEN:
            raise ValueError("hidden must have 1..%d entries, got %r" % (MAX_HIDDEN, hidden))
        s = cls()
        s.d, s.c, s.lat, s.n_hidden = int(d), int(c), int(lat), len(hidden)
        for i, h in enumerate(hidden):
            s.hidden[i] = h
        s.act = 0 if activation == "tanh" else 1       # cvae.py:26-32
        # per call (test / measurement aid): "generic" pins one thread per row, "lmm" the any-shape MFMA kernels
        s.family = {"auto": 0, "generic": 1, "lmm": 2}[family]
        return s


class HipLibraryMissing(RuntimeError):
    pass


_lib = None
_lock = threading.Lock()

_VP, _I64, _U64, _F, _D, _SZ = C.c_void_p, C.c_int64, C.c_uint64, C.c_float, C.c_double, C.c_size_t
_SP = C.POINTER(RnvpShape)

_SIGNATURES = {
    "rnvp_version": (C.c_int, []),
    "rnvp_status_string": (C.c_char_p, [C.c_int]),
    "rnvp_param_count": (_SZ, [_SP]),
    "rnvp_workspace_bytes": (_SZ, [_SP, C.c_int, _I64]),
    "rnvp_kernel_path": (C.c_int, [_SP, _VP, C.c_int]),
    "rnvp_fit_epoch_resident": (C.c_int, [_SP, _I64]),
    "rnvp_forward_logprob": (C.c_int, [_VP, _SP, _VP, _VP, _VP, _VP, _VP, _I64, _VP, _VP, _VP, _VP, _VP, _SZ]),
    "rnvp_inverse": (C.c_int, [_VP, _SP, _VP, _VP, _VP, _VP, _I64, _VP, _VP, _SZ]),
    "rnvp_prior_normal": (C.c_int, [_VP, _U64, _I64, _I64, C.c_int32, _VP]),
    "rnvp_prior_normal_torch_cpu": (C.c_int, [_VP, _VP, _I64, _VP, _VP, _VP, _SZ]),
    "rnvp_prior_torch_workspace_bytes": (_SZ, []),
    "rnvp_randperm_workspace_bytes": (_SZ, [_I64]),
    "rnvp_randperm_torch_cpu": (C.c_int, [_VP, _VP, _I64, _VP, _VP, _SZ]),
    "rnvp_sample": (C.c_int, [_VP, _SP, _VP, _VP, _VP, _I64, _U64, _I64, _VP, _VP, _SZ]),
    "rnvp_loss_grad": (C.c_int, [_VP, _SP, _VP, _VP, _VP, _VP, _VP, _I64, _F, _VP, _VP, _VP, _SZ]),
    "rnvp_loss_grad_zseed": (C.c_int, [_VP, _SP, _VP, _VP, _VP, _VP, _VP, _I64, _F, _VP, _VP, _VP, _VP, _SZ]),
    "rnvp_backward": (C.c_int, [_VP, _SP, _VP, _VP, _VP, _VP, _VP, _I64, _VP, _VP, _VP, _VP, _VP, _SZ]),
    "rnvp_backward_cond_workspace_bytes": (_SZ, [_SP, _I64]),
    "rnvp_backward_cond": (C.c_int, [_VP, _SP, _VP, _VP, _VP, _VP, _VP, _I64, _VP, _VP, _VP, _VP, _VP, _VP, _SZ]),
    "rnvp_inverse_backward": (C.c_int, [_VP, _SP, _VP, _VP, _VP, _VP, _I64, _VP, _VP, _VP, _VP, _VP, _SZ]),
    "rnvp_adam_step": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _I64, _D, _D, _D, _D, _D, _I64]),
    "rnvp_dp_finish_step": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _I64, _D, _D, _D, _D, _D, _I64, _VP]),
    "rnvp_train_step": (C.c_int, [_VP, _SP, _VP, _VP, _VP, _VP, _VP, _I64, _F, _VP, _VP, _VP, _VP,
                                  _D, _D, _D, _D, _D, _I64, _VP, _SZ]),
    "rnvp_fit_epoch": (C.c_int, [_VP, _SP, _VP, _VP, _VP, _VP, _VP, _I64, _I64, _VP, _VP, _VP, _VP,
                                 _D, _D, _D, _D, _D, _I64, _VP, _SZ]),
    "rnvp_fit_epochs": (C.c_int, [_VP, _SP, _VP, _VP, _VP, _VP, _VP, _I64, _I64, _I64, _VP, _VP, _VP, _VP,
                                  _D, _D, _D, _D, _D, _I64, _VP, _SZ]),
    "rnvp_dp_unique_id": (C.c_int, [_VP]),
    "rnvp_dp_init": (C.c_int, [_VP, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "rnvp_dp_destroy": (C.c_int, [_VP]),
    "rnvp_dp_all_reduce": (C.c_int, [_VP, _VP, _VP, _I64]),
    "rnvp_fit_epoch_dp": (C.c_int, [_VP, _VP, _SP, _VP, _VP, _VP, _VP, _VP, _I64, _I64, _VP, _VP, _VP, _VP,
                                    _D, _D, _D, _D, _D, _I64, _VP, _SZ]),
    "rnvp_fit_epoch_dp_cb": (C.c_int, [_VP, _VP, _VP, C.c_int, C.c_int, _SP, _VP, _VP, _VP, _VP, _VP, _I64, _I64, _VP, _VP, _VP, _VP,
                                       _D, _D, _D, _D, _D, _I64, _VP, _SZ]),
    "rnvp_dp_set_chunks": (C.c_int, [_VP, C.c_int]),
    "rnvp_fit_epoch_dp_cb_chunked": (C.c_int, [_VP, _VP, _VP, C.c_int, C.c_int, C.c_int, _SP, _VP, _VP, _VP, _VP, _VP, _I64, _I64, _VP, _VP,
                                               _VP, _VP, _D, _D, _D, _D, _D, _I64, _VP, _SZ]),
    "cvae_param_count": (_SZ, [C.POINTER(CvaeShape)]),
    "cvae_workspace_bytes": (_SZ, [C.POINTER(CvaeShape), _I64]),
    "cvae_kernel_path": (C.c_int, [C.POINTER(CvaeShape)]),
    "cvae_loss_grad": (C.c_int, [_VP, C.POINTER(CvaeShape), _VP, _VP, _VP, _VP, _VP, _I64, _F, _F, _VP, _VP, _VP, _SZ]),
    "cvae_train_step": (C.c_int, [_VP, C.POINTER(CvaeShape), _VP, _VP, _VP, _VP, _VP, _I64, _F, _F, _VP, _VP, _VP, _VP,
                                  _D, _D, _D, _D, _D, _I64, _VP, _SZ]),
    "cvae_fit_epoch": (C.c_int, [_VP, C.POINTER(CvaeShape), _VP, _VP, _VP, _VP, _VP, _I64, _I64, _F, _VP, _VP, _VP, _VP,
                                 _D, _D, _D, _D, _D, _I64, _VP, _SZ]),
    "cvae_fit_epoch_resident": (C.c_int, [C.POINTER(CvaeShape), _I64]),
    "cvae_decode": (C.c_int, [_VP, C.POINTER(CvaeShape), _VP, _VP, _VP, _I64, _VP, _VP, _SZ]),
    "cvae_encode": (C.c_int, [_VP, C.POINTER(CvaeShape), _VP, _VP, _VP, _I64, _VP, _VP, _VP, _SZ]),
    "rnvp_profile_enable": (C.c_int, [C.c_int]),
    "rnvp_profile_read": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_float)]),
    "rnvp_last_dispatch": (C.c_int, [C.c_int, C.POINTER(RnvpDispatch)]),
}
EXPORTS = tuple(_SIGNATURES)


def lib():
    """Load librnvp_hip.so once; raise loudly if it has not been built."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise HipLibraryMissing(
                        "%s not found: build it with `make -C probaforms_amd/csrc` "
                        "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
                        "probaforms_amd has no CPU fallback." % LIB_PATH)
                L = C.CDLL(LIB_PATH)
                L.rnvp_version.restype, L.rnvp_version.argtypes = C.c_int, []
                have = int(L.rnvp_version())
                if have != ABI_VERSION:
                    # an older / newer build of the same library (RNVP_HIP_LIB, a stale A/B variant): struct layouts and
                    # argument lists differ between versions, calling through would corrupt memory silently
                    raise HipLibraryMissing("%s reports rnvp_version() = %d, this binding is written for %d: rebuild it "
                                            "(`make -C probaforms_amd/csrc`)" % (LIB_PATH, have, ABI_VERSION))
                for name, (res, args) in _SIGNATURES.items():
                    fn = getattr(L, name)
                    fn.restype, fn.argtypes = res, args
                _lib = L
    return _lib


def check(status, what):
    if status != 0:
        msg = lib().rnvp_status_string(status)
        raise RuntimeError("%s failed: %s (status %d)" % (what, msg.decode() if msg else "?", status))


def _ptr(t, dtype, what):
    if t is None:
        return None
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError("%s must be a tensor on a HIP device (got %s); probaforms_amd has no CPU path"
                           % (what, getattr(t, "device", type(t))))
    if t.dtype != dtype or not t.is_contiguous():
        raise RuntimeError("%s must be contiguous %s (got %s, contiguous=%s)"
                           % (what, dtype, t.dtype, t.is_contiguous()))
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _call(name, args):
    """arguments are validated (device, dtype, contiguity) BEFORE the stream is looked up, so a CPU
    tensor is reported as such even on a box without a GPU"""
    st = getattr(lib(), name)(_stream(), *args)
    check(st, name)


def _ws(ws):
    return _ptr(ws, torch.uint8, "workspace"), (0 if ws is None else ws.numel())


def profile_enable(capacity):
    check(lib().rnvp_profile_enable(int(capacity)), "rnvp_profile_enable")


def profile_read(kind=PROFILE_TRAIN):
    """-> (launches, total_ms) of the hot kernel of `kind` (PROFILE_*) since the last read"""
    n, ms = C.c_int(0), C.c_float(0.0)
    check(lib().rnvp_profile_read(int(kind), C.byref(n), C.byref(ms)), "rnvp_profile_read")
    return n.value, ms.value


def last_dispatch(kind=PROFILE_TRAIN):
    """-> dict(kernel, variant, row_tiles, waves, grid, gemm1_fwd, launches, rows) of this thread's last call of `kind`"""
    d = RnvpDispatch()
    check(lib().rnvp_last_dispatch(int(kind), C.byref(d)), "rnvp_last_dispatch")
    return dict(kernel=d.kernel.decode(), variant=VARIANTS.get(d.variant, str(d.variant)), row_tiles=d.row_tiles, waves=d.waves,
                grid=d.grid, gemm1_fwd={1: "f32", 2: "bx3"}.get(d.gemm1_fwd, "?"), launches=d.launches, rows=d.rows)


def param_count(shape):
    return int(lib().rnvp_param_count(C.byref(shape)))


def workspace_bytes(shape, op, max_rows):
    return int(lib().rnvp_workspace_bytes(C.byref(shape), op, int(max_rows)))


def kernel_path(shape, host_masks_u8, op):
    p = None if host_masks_u8 is None else host_masks_u8.ctypes.data
    return int(lib().rnvp_kernel_path(C.byref(shape), p, op))


def forward_logprob(shape, params, masks, x, c, row_index, n_rows, z_out, logdet_out, logp_out, logp_sum, ws):
    wp, wn = _ws(ws)
    _call("rnvp_forward_logprob", (C.byref(shape), _ptr(params, torch.float32, "params"), _ptr(masks, torch.uint8, "masks"),
        _ptr(x, torch.float32, "x"), _ptr(c, torch.float32, "c"), _ptr(row_index, torch.int64, "row_index"),
        int(n_rows), _ptr(z_out, torch.float32, "z_out"), _ptr(logdet_out, torch.float32, "logdet_out"),
        _ptr(logp_out, torch.float32, "logp_out"), _ptr(logp_sum, torch.float32, "logp_sum"), wp, wn))


def inverse(shape, params, masks, z, c, n_rows, x_out, ws):
    wp, wn = _ws(ws)
    _call("rnvp_inverse", (C.byref(shape), _ptr(params, torch.float32, "params"), _ptr(masks, torch.uint8, "masks"),
        _ptr(z, torch.float32, "z"), _ptr(c, torch.float32, "c"), int(n_rows),
        _ptr(x_out, torch.float32, "x_out"), wp, wn))


def prior_normal(seed, row_offset, n_rows, d, z_out):
    """z_out[r][j] = N(0,1)(seed, row_offset + r, j): the counter-based 'device' prior (rnvp_prior_normal)"""
    _call("rnvp_prior_normal", (int(seed) & 0xFFFFFFFFFFFFFFFF, int(row_offset), int(n_rows), int(d),
                                _ptr(z_out, torch.float32, "z_out")))


def prior_torch_workspace_bytes():
    return int(lib().rnvp_prior_torch_workspace_bytes())


def prior_normal_torch_cpu(mt_state, count, z_out, tail16, ws):
    """z_out[:count] = torch.randn(count) of the CPU generator whose twister state is mt_state ([625] int32 on the device:
    624 words + position), advanced in place (rnvp_prior_normal_torch_cpu); ws: prior_torch_workspace_bytes() bytes"""
    wp, wn = _ws(ws)
    _call("rnvp_prior_normal_torch_cpu", (_ptr(mt_state, torch.int32, "mt_state"), int(count), _ptr(z_out, torch.float32, "z_out"),
                                          _ptr(tail16, torch.float32, "tail16"), wp, wn))


def randperm_workspace_bytes(n):
    return int(lib().rnvp_randperm_workspace_bytes(int(n)))


def randperm_torch_cpu(mt_state, n, perm_out, ws):
    """perm_out[:n] (int64, device) = torch.randperm(n, generator=g) of the CPU generator whose twister state is mt_state ([625] int32 on
    the device), advanced in place (rnvp_randperm_torch_cpu); ws: randperm_workspace_bytes(n) bytes"""
    wp, wn = _ws(ws)
    _call("rnvp_randperm_torch_cpu", (_ptr(mt_state, torch.int32, "mt_state"), int(n), _ptr(perm_out, torch.int64, "perm_out"), wp, wn))


def sample(shape, params, masks, c, n_rows, seed, row_offset, x_out, ws):
    """prior draw fused into the inverse (rnvp_sample)"""
    wp, wn = _ws(ws)
    _call("rnvp_sample", (C.byref(shape), _ptr(params, torch.float32, "params"), _ptr(masks, torch.uint8, "masks"),
        _ptr(c, torch.float32, "c"), int(n_rows), int(seed) & 0xFFFFFFFFFFFFFFFF, int(row_offset),
        _ptr(x_out, torch.float32, "x_out"), wp, wn))


def loss_grad(shape, params, masks, x, c, row_index, n_rows, inv_B, grad_out, loss_out, ws):
    wp, wn = _ws(ws)
    _call("rnvp_loss_grad", (C.byref(shape), _ptr(params, torch.float32, "params"), _ptr(masks, torch.uint8, "masks"),
        _ptr(x, torch.float32, "x"), _ptr(c, torch.float32, "c"), _ptr(row_index, torch.int64, "row_index"),
        int(n_rows), float(inv_B), _ptr(grad_out, torch.float32, "grad_out"),
        _ptr(loss_out, torch.float32, "loss_out"), wp, wn))


def loss_grad_zseed(shape, params, masks, x, c, row_index, n_rows, inv_B, gz, grad_out, loss_out, ws):
    wp, wn = _ws(ws)
    _call("rnvp_loss_grad_zseed", (C.byref(shape), _ptr(params, torch.float32, "params"), _ptr(masks, torch.uint8, "masks"),
        _ptr(x, torch.float32, "x"), _ptr(c, torch.float32, "c"), _ptr(row_index, torch.int64, "row_index"),
        int(n_rows), float(inv_B), _ptr(gz, torch.float32, "gz"), _ptr(grad_out, torch.float32, "grad_out"),
        _ptr(loss_out, torch.float32, "loss_out"), wp, wn))


def backward(shape, params, masks, x, c, row_index, n_rows, gz, gld, grad_out, gx_out, ws):
    """vector-Jacobian product of rnvp_forward_logprob: (d loss / d z, d loss / d logdet) -> (d loss / d params, d loss / d x)"""
    wp, wn = _ws(ws)
    _call("rnvp_backward", (C.byref(shape), _ptr(params, torch.float32, "params"), _ptr(masks, torch.uint8, "masks"),
        _ptr(x, torch.float32, "x"), _ptr(c, torch.float32, "c"), _ptr(row_index, torch.int64, "row_index"),
        int(n_rows), _ptr(gz, torch.float32, "gz"), _ptr(gld, torch.float32, "gld"),
        _ptr(grad_out, torch.float32, "grad_out"), _ptr(gx_out, torch.float32, "gx_out"), wp, wn))


def backward_cond_workspace_bytes(shape, max_rows):
    """bytes rnvp_backward_cond / rnvp_inverse_backward need; 0 when no kernel serves the shape (hidden sizes beyond a CU's LDS)"""
    return int(lib().rnvp_backward_cond_workspace_bytes(C.byref(shape), int(max_rows)))


def backward_cond(shape, params, masks, x, c, row_index, n_rows, gz, gld, grad_out, gx_out, gc_out, ws):
    """rnvp_backward plus d loss / d c (gc_out [n_rows, c], nullable)"""
    wp, wn = _ws(ws)
    _call("rnvp_backward_cond", (C.byref(shape), _ptr(params, torch.float32, "params"), _ptr(masks, torch.uint8, "masks"),
        _ptr(x, torch.float32, "x"), _ptr(c, torch.float32, "c"), _ptr(row_index, torch.int64, "row_index"),
        int(n_rows), _ptr(gz, torch.float32, "gz"), _ptr(gld, torch.float32, "gld"),
        _ptr(grad_out, torch.float32, "grad_out"), _ptr(gx_out, torch.float32, "gx_out"), _ptr(gc_out, torch.float32, "gc_out"), wp, wn))


def inverse_backward(shape, params, masks, z, c, n_rows, gx, grad_out, gz_out, gc_out, ws):
    """vector-Jacobian product of rnvp_inverse: d loss / d x -> (d loss / d params, d loss / d z, d loss / d c)"""
    wp, wn = _ws(ws)
    _call("rnvp_inverse_backward", (C.byref(shape), _ptr(params, torch.float32, "params"), _ptr(masks, torch.uint8, "masks"),
        _ptr(z, torch.float32, "z"), _ptr(c, torch.float32, "c"), int(n_rows), _ptr(gx, torch.float32, "gx"),
        _ptr(grad_out, torch.float32, "grad_out"), _ptr(gz_out, torch.float32, "gz_out"), _ptr(gc_out, torch.float32, "gc_out"), wp, wn))


def adam_step(params, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step):
    _call("rnvp_adam_step", (_ptr(params, torch.float32, "params"), _ptr(grad, torch.float32, "grad"),
        _ptr(exp_avg, torch.float32, "exp_avg"), _ptr(exp_avg_sq, torch.float32, "exp_avg_sq"), int(n),
        float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), int(step)))


def dp_finish_step(params, grad_loss, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step, loss_out):
    _call("rnvp_dp_finish_step", (_ptr(params, torch.float32, "params"), _ptr(grad_loss, torch.float32, "grad_loss"),
        _ptr(exp_avg, torch.float32, "exp_avg"), _ptr(exp_avg_sq, torch.float32, "exp_avg_sq"), int(n),
        float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), int(step),
        _ptr(loss_out, torch.float32, "loss_out")))


def train_step(shape, params, masks, x, c, row_index, n_rows, inv_B, grad_buf, loss_out, exp_avg, exp_avg_sq,
               lr, beta1, beta2, eps, weight_decay, step, ws):
    wp, wn = _ws(ws)
    _call("rnvp_train_step", (C.byref(shape), _ptr(params, torch.float32, "params"), _ptr(masks, torch.uint8, "masks"),
        _ptr(x, torch.float32, "x"), _ptr(c, torch.float32, "c"), _ptr(row_index, torch.int64, "row_index"),
        int(n_rows), float(inv_B), _ptr(grad_buf, torch.float32, "grad_buf"),
        _ptr(loss_out, torch.float32, "loss_out"), _ptr(exp_avg, torch.float32, "exp_avg"),
        _ptr(exp_avg_sq, torch.float32, "exp_avg_sq"), float(lr), float(beta1), float(beta2), float(eps),
        float(weight_decay), int(step), wp, wn))


def fit_epoch_resident(shape, batch_size):
    """True when rnvp_fit_epoch runs this shape / batch size as one persistent launch per epoch (rnvp_resident.hip)"""
    return bool(lib().rnvp_fit_epoch_resident(C.byref(shape), int(batch_size)))


def fit_epoch(shape, params, masks, x, c, perm, n, batch_size, grad_buf, loss_hist, exp_avg, exp_avg_sq,
              lr, beta1, beta2, eps, weight_decay, first_step, ws):
    wp, wn = _ws(ws)
    _call("rnvp_fit_epoch", (C.byref(shape), _ptr(params, torch.float32, "params"), _ptr(masks, torch.uint8, "masks"),
          _ptr(x, torch.float32, "x"), _ptr(c, torch.float32, "c"), _ptr(perm, torch.int64, "perm"), int(n),
          int(batch_size), _ptr(grad_buf, torch.float32, "grad_buf"), _ptr(loss_hist, torch.float32, "loss_hist"),
          _ptr(exp_avg, torch.float32, "exp_avg"), _ptr(exp_avg_sq, torch.float32, "exp_avg_sq"), float(lr),
          float(beta1), float(beta2), float(eps), float(weight_decay), int(first_step), wp, wn))


def fit_epochs(shape, params, masks, x, c, perms, n, batch_size, n_epochs, grad_buf, loss_hist, exp_avg, exp_avg_sq,
               lr, beta1, beta2, eps, weight_decay, first_step, ws):
    """n_epochs epochs in one library call: perms [n_epochs, n], loss_hist [n_epochs, batches per epoch]"""
    wp, wn = _ws(ws)
    _call("rnvp_fit_epochs", (C.byref(shape), _ptr(params, torch.float32, "params"), _ptr(masks, torch.uint8, "masks"),
          _ptr(x, torch.float32, "x"), _ptr(c, torch.float32, "c"), _ptr(perms, torch.int64, "perms"), int(n),
          int(batch_size), int(n_epochs), _ptr(grad_buf, torch.float32, "grad_buf"), _ptr(loss_hist, torch.float32, "loss_hist"),
          _ptr(exp_avg, torch.float32, "exp_avg"), _ptr(exp_avg_sq, torch.float32, "exp_avg_sq"), float(lr),
          float(beta1), float(beta2), float(eps), float(weight_decay), int(first_step), wp, wn))


# ---- data parallel: the library's own RCCL communicator (include/rnvp_hip.h, rnvp_dp_*) --------------------------------
def dp_unique_id():
    """128 bytes identifying a new communicator (call on rank 0, send to every rank)"""
    buf = C.create_string_buffer(128)
    check(lib().rnvp_dp_unique_id(buf), "rnvp_dp_unique_id")
    return buf.raw


def dp_init(uid, rank, world):
    """collective: every rank calls it with rank 0's id; returns the opaque communicator handle"""
    h = C.c_void_p()
    check(lib().rnvp_dp_init(C.create_string_buffer(bytes(uid), 128), int(rank), int(world), C.byref(h)), "rnvp_dp_init")
    return h


def dp_destroy(comm):
    if comm is not None:
        lib().rnvp_dp_destroy(comm)


def dp_all_reduce(comm, buf, count):
    _call("rnvp_dp_all_reduce", (comm, _ptr(buf, torch.float32, "buf"), int(count)))


def fit_epoch_dp(comm, shape, params, masks, x, c, perm, n, batch_size, grad_loss, loss_hist, exp_avg, exp_avg_sq,
                 lr, beta1, beta2, eps, weight_decay, first_step, ws):
    """one epoch of the data-parallel batch loop in one call (comm None: one rank, no exchange)"""
    wp, wn = _ws(ws)
    _call("rnvp_fit_epoch_dp", (comm, C.byref(shape), _ptr(params, torch.float32, "params"), _ptr(masks, torch.uint8, "masks"),
          _ptr(x, torch.float32, "x"), _ptr(c, torch.float32, "c"), _ptr(perm, torch.int64, "perm"), int(n),
          int(batch_size), _ptr(grad_loss, torch.float32, "grad_loss"), _ptr(loss_hist, torch.float32, "loss_hist"),
          _ptr(exp_avg, torch.float32, "exp_avg"), _ptr(exp_avg_sq, torch.float32, "exp_avg_sq"), float(lr),
          float(beta1), float(beta2), float(eps), float(weight_decay), int(first_step), wp, wn))


ALL_REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64)      # rnvp_all_reduce_fn


def dp_set_chunks(comm, chunks):
    """the library's communicator exchanges the step's [gradient | loss] message in `chunks` groups of layers, each group's
    all-reduce on a side stream under the next group's partial sums (rnvp_dp_set_chunks); 1 = one message per step"""
    check(lib().rnvp_dp_set_chunks(comm, int(chunks)), "rnvp_dp_set_chunks")


def fit_epoch_dp_cb(all_reduce, rank, world, shape, params, masks, x, c, perm, n, batch_size, grad_loss, loss_hist, exp_avg,
                    exp_avg_sq, lr, beta1, beta2, eps, weight_decay, first_step, ws, chunks=1):
    """rnvp_fit_epoch_dp with the exchange supplied by the caller: `all_reduce(buf_tensor_view, count)` must sum the view
    over the ranks in place, ordered with the current stream (a torch.distributed all_reduce on the tensor is).  The library
    hands back raw pointers; the wrapper checks that they lie inside `grad_loss` and passes the matching view (the whole
    message, or one chunk of layers with chunks > 1: rnvp_fit_epoch_dp_cb_chunked)."""
    wp, wn = _ws(ws)
    base = grad_loss.data_ptr()
    failure = []

    def _cb(_ctx, _stream, buf, count):
        try:
            off = ((buf or 0) - base) // 4
            if (buf or 0) < base or ((buf or 0) - base) % 4 or off + count > grad_loss.numel():
                raise RuntimeError("rnvp_fit_epoch_dp_cb handed the exchange a buffer that is not inside grad_loss")
            all_reduce(grad_loss[off:off + count], int(count))
            return 0
        except BaseException as e:          # an exception must not unwind through the C frames
            failure.append(e)
            return 1

    cb = ALL_REDUCE_FN(_cb)
    head = (_stream(), C.cast(cb, C.c_void_p), None, int(rank), int(world))
    fn = lib().rnvp_fit_epoch_dp_cb
    if int(chunks) > 1:
        head, fn = head + (int(chunks),), lib().rnvp_fit_epoch_dp_cb_chunked
    st = fn(*head, C.byref(shape),
          _ptr(params, torch.float32, "params"), _ptr(masks, torch.uint8, "masks"),
          _ptr(x, torch.float32, "x"), _ptr(c, torch.float32, "c"), _ptr(perm, torch.int64, "perm"), int(n),
          int(batch_size), _ptr(grad_loss, torch.float32, "grad_loss"), _ptr(loss_hist, torch.float32, "loss_hist"),
          _ptr(exp_avg, torch.float32, "exp_avg"), _ptr(exp_avg_sq, torch.float32, "exp_avg_sq"), float(lr),
          float(beta1), float(beta2), float(eps), float(weight_decay), int(first_step), wp, wn)
    if failure:
        raise failure[0]
    check(st, "rnvp_fit_epoch_dp_cb")


# ---- CVAE (include/cvae_hip.h) ----------------------------------------------------------------
def cvae_param_count(shape):
    return int(lib().cvae_param_count(C.byref(shape)))


def cvae_workspace_bytes(shape, max_rows):
    return int(lib().cvae_workspace_bytes(C.byref(shape), int(max_rows)))


def cvae_kernel_path(shape):
    """PATH_MFMA / PATH_LMM / PATH_GENERIC: the kernels cvae_loss_grad runs for this shape"""
    return int(lib().cvae_kernel_path(C.byref(shape)))


def cvae_loss_grad(shape, params, x, c, row_index, eps, n_rows, inv_B, kl_weight, grad_out, loss_out, ws):
    wp, wn = _ws(ws)
    _call("cvae_loss_grad", (C.byref(shape), _ptr(params, torch.float32, "params"), _ptr(x, torch.float32, "x"),
          _ptr(c, torch.float32, "c"), _ptr(row_index, torch.int64, "row_index"), _ptr(eps, torch.float32, "eps"),
          int(n_rows), float(inv_B), float(kl_weight), _ptr(grad_out, torch.float32, "grad_out"),
          _ptr(loss_out, torch.float32, "loss_out"), wp, wn))


def cvae_train_step(shape, params, x, c, row_index, eps, n_rows, inv_B, kl_weight, grad_buf, loss_out, exp_avg, exp_avg_sq,
                    lr, beta1, beta2, adam_eps, weight_decay, step, ws):
    """cvae_loss_grad + Adam, fused on the MFMA path (one launch fewer)"""
    wp, wn = _ws(ws)
    _call("cvae_train_step", (C.byref(shape), _ptr(params, torch.float32, "params"), _ptr(x, torch.float32, "x"),
          _ptr(c, torch.float32, "c"), _ptr(row_index, torch.int64, "row_index"), _ptr(eps, torch.float32, "eps"),
          int(n_rows), float(inv_B), float(kl_weight), _ptr(grad_buf, torch.float32, "grad_buf"),
          _ptr(loss_out, torch.float32, "loss_out"), _ptr(exp_avg, torch.float32, "exp_avg"),
          _ptr(exp_avg_sq, torch.float32, "exp_avg_sq"), float(lr), float(beta1), float(beta2), float(adam_eps),
          float(weight_decay), int(step), wp, wn))


def cvae_fit_epoch_resident(shape, batch_size):
    """True when cvae_fit_epoch runs this shape / batch size as one persistent launch per epoch (rnvp_resident.hip)"""
    return bool(lib().cvae_fit_epoch_resident(C.byref(shape), int(batch_size)))


def cvae_fit_epoch(shape, params, x, c, perm, eps, n, batch_size, kl_weight, grad_buf, loss_hist, exp_avg, exp_avg_sq,
                   lr, beta1, beta2, adam_eps, weight_decay, first_step, ws):
    """all batches of one epoch of CVAE.fit in one library call (eps [n, latent] in batch order)"""
    wp, wn = _ws(ws)
    _call("cvae_fit_epoch", (C.byref(shape), _ptr(params, torch.float32, "params"), _ptr(x, torch.float32, "x"),
          _ptr(c, torch.float32, "c"), _ptr(perm, torch.int64, "perm"), _ptr(eps, torch.float32, "eps"), int(n), int(batch_size),
          float(kl_weight), _ptr(grad_buf, torch.float32, "grad_buf"), _ptr(loss_hist, torch.float32, "loss_hist"),
          _ptr(exp_avg, torch.float32, "exp_avg"), _ptr(exp_avg_sq, torch.float32, "exp_avg_sq"), float(lr), float(beta1),
          float(beta2), float(adam_eps), float(weight_decay), int(first_step), wp, wn))


def cvae_decode(shape, params, z, c, n_rows, x_out, ws=None):
    """ws (cvae_workspace_bytes) enables the MFMA kernels where the shape allows; None runs the generic ones"""
    wp, wn = _ws(ws) if ws is not None else (None, 0)
    _call("cvae_decode", (C.byref(shape), _ptr(params, torch.float32, "params"), _ptr(z, torch.float32, "z"),
          _ptr(c, torch.float32, "c"), int(n_rows), _ptr(x_out, torch.float32, "x_out"), wp, wn))


def cvae_encode(shape, params, x, c, n_rows, mu_out, ls_out, ws=None):
    wp, wn = _ws(ws) if ws is not None else (None, 0)
    _call("cvae_encode", (C.byref(shape), _ptr(params, torch.float32, "params"), _ptr(x, torch.float32, "x"),
          _ptr(c, torch.float32, "c"), int(n_rows), _ptr(mu_out, torch.float32, "mu_out"),
          _ptr(ls_out, torch.float32, "log_sigma_out"), wp, wn))
