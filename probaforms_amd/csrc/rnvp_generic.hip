// rnvp_generic.hip -- any-shape RealNVP coupling stack for gfx950 (VALU + LDS).
//
// Serves every configuration the reference accepts (odd d, C=None, d=1, several hidden
// layers, ReLU, user masks: /root/reference/probaforms/models/realnvp.py:19-129) that the
// register-chained MFMA path (rnvp_mfma.hip) does not specialise.  Design:
//   * one thread owns one row for the whole L-layer stack; its row state (x, the net input
//     [x*mask || c], hidden activations, t, s) lives in LDS laid out [feature][TBP] with
//     TBP = TB + 1 (odd), so both "thread = row" and "thread = parameter" sweeps are
//     bank-conflict free;
//   * weights are read with wave-uniform addresses straight from L2 (scalar loads);
//   * x/c rows touch HBM once on the way in and z / log-prob once on the way out;
//   * the backward recomputes the nets per layer from the saved layer inputs, forms the
//     weight gradients with a "thread = parameter" sweep over the block's rows and keeps a
//     block-private partial gradient that a second kernel reduces in a fixed order
//     (deterministic: no float atomics).
#include <atomic>

#include "rnvp_common.h"
#include "rnvp_generic_net.h"

namespace rnvp {
namespace {

constexpr int kMaxGrid = 512;          // persistent blocks for forward / inverse
constexpr int kMaxGridTrain = 256;     // blocks holding a private partial gradient
constexpr size_t kLdsSoft = 64 * 1024; // prefer tilings that leave room for >1 block per CU
constexpr size_t kLdsHard = 160 * 1024 - 1024;

struct Tiling {
    int TB;        // rows per block tile
    int TBP;       // padded row stride in LDS
    int threads;   // blockDim.x
    size_t lds;    // dynamic LDS bytes
};

// ---- forward + log-det + prior (realnvp.py:91-101, nflow.py:107-117) ---------------------
__global__ void __launch_bounds__(256)
k_generic_forward(KShape s, const float *__restrict__ params, const uint8_t *__restrict__ masks,
                  const float *__restrict__ x, const float *__restrict__ c,
                  const int64_t *__restrict__ row_index, int64_t n,
                  float *z_out, float *logdet_out, float *logp_out, float *part,
                  int TB, int TBP) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int t = threadIdx.x, d = s.d, cd = s.c;
    float *xcur = lds;
    float *uin = xcur + d * TBP;
    float *hb0 = uin + (d + cd) * TBP;
    float *hb1 = hb0 + s.hmax * TBP;
    float *tout = hb1 + s.hmax * TBP;
    float *sout = tout + d * TBP;
    float *red = sout + d * TBP;
    const int64_t ntiles = (n + TB - 1) / TB;
    const float prior_c = 0.5f * (float)d * kLog2Pi;
    float block_sum = 0.f;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t row = tile * TB + t;
        const bool valid = (t < TB) && (row < n);
        float lp = 0.f;
        if (t < TB) {
            const int64_t src = valid ? (row_index ? row_index[row] : row) : 0;
            for (int j = 0; j < d; ++j) xcur[j * TBP + t] = valid ? x[src * d + j] : 0.f;
            for (int j = 0; j < cd; ++j) uin[(d + j) * TBP + t] = valid ? c[src * cd + j] : 0.f;
            float ld = 0.f;
            for (int l = 0; l < s.L; ++l) {
                const uint8_t *m = masks + l * d;
                const float *pl = params + (size_t)l * 2 * s.npn;
                for (int j = 0; j < d; ++j) uin[j * TBP + t] = xcur[j * TBP + t] * (float)m[j];
                net_forward<false>(pl, s, uin, hb0, hb1, tout, TBP, t);           // nn_t
                net_forward<false>(pl + s.npn, s, uin, hb0, hb1, sout, TBP, t);   // nn_s
                for (int j = 0; j < d; ++j) {
                    if (!m[j]) {
                        const float sv = sout[j * TBP + t];
                        xcur[j * TBP + t] = fmaf(xcur[j * TBP + t], expf(sv), tout[j * TBP + t]);
                        ld += sv;
                    }
                }
            }
            float ss = 0.f;
            for (int j = 0; j < d; ++j) { const float zv = xcur[j * TBP + t]; ss = fmaf(zv, zv, ss); }
            lp = ld + (-0.5f * ss - prior_c);
            if (valid) {
                if (z_out) for (int j = 0; j < d; ++j) z_out[row * d + j] = xcur[j * TBP + t];
                if (logdet_out) logdet_out[row] = ld;
                if (logp_out) logp_out[row] = lp;
            }
        }
        if (part) {
            red[t] = valid ? lp : 0.f;
            __syncthreads();
            if (t == 0) { float a = 0.f; for (int r = 0; r < TB; ++r) a += red[r]; block_sum += a; }
            __syncthreads();
        }
    }
    if (part && t == 0) part[blockIdx.x] = block_sum;
}

__global__ void __launch_bounds__(64)
k_sum_partials(const float *__restrict__ part, int G, float scale, float *out) {
    // one wave: lane i adds part[i], part[i + 64], ... in order, then a fixed butterfly: deterministic
    const int lane = threadIdx.x;
    float a = 0.f;
    for (int i = lane; i < G; i += 64) a += part[i];
    for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off);
    if (lane == 0) out[0] = a * scale;
}

// ---- inverse (realnvp.py:120-129, nflow.py:142-143) ---------------------------------------
__global__ void __launch_bounds__(256)
k_generic_inverse(KShape s, const float *__restrict__ params, const uint8_t *__restrict__ masks,
                  const float *z, const float *__restrict__ c, int64_t n, float *x_out,
                  int TB, int TBP) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int t = threadIdx.x, d = s.d, cd = s.c;
    float *xcur = lds;
    float *uin = xcur + d * TBP;
    float *hb0 = uin + (d + cd) * TBP;
    float *hb1 = hb0 + s.hmax * TBP;
    float *tout = hb1 + s.hmax * TBP;
    float *sout = tout + d * TBP;
    const int64_t ntiles = (n + TB - 1) / TB;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t row = tile * TB + t;
        if (t >= TB || row >= n) continue;       // no barriers in this kernel
        for (int j = 0; j < d; ++j) xcur[j * TBP + t] = z[row * d + j];
        for (int j = 0; j < cd; ++j) uin[(d + j) * TBP + t] = c[row * cd + j];
        for (int l = s.L - 1; l >= 0; --l) {
            const uint8_t *m = masks + l * d;
            const float *pl = params + (size_t)l * 2 * s.npn;
            for (int j = 0; j < d; ++j) uin[j * TBP + t] = xcur[j * TBP + t] * (float)m[j];
            net_forward<false>(pl, s, uin, hb0, hb1, tout, TBP, t);
            net_forward<false>(pl + s.npn, s, uin, hb0, hb1, sout, TBP, t);
            for (int j = 0; j < d; ++j)
                if (!m[j])
                    xcur[j * TBP + t] = (xcur[j * TBP + t] - tout[j * TBP + t]) * expf(-sout[j * TBP + t]);
        }
        for (int j = 0; j < d; ++j) x_out[row * d + j] = xcur[j * TBP + t];
    }
}

// ---- loss + gradient (realnvp.py:246-250; backward derived in SURVEY.md 3.3) ---------------
// Seeds (rnvp_common.h) select the differentiable-seam variants: gz / gld / gx (rnvp_backward), gc with KShape::gcw (the
// conditions' gradient: rnvp_backward_cond -- they enter every net through torch.cat((X * mask, C)), realnvp.py:92), and inv
// (rnvp_inverse_backward: the rows are z; phase 1 runs the INVERSE, realnvp.py:120-128 / nflow.py:141-145, saving behind each
// layer the input of f's layer l; phase 2 walks the layers 0 .. L-1 -- the order in which g applied them last to first -- with
// d/dt = -g_x e^{-s}, d/ds = -g_x x_u as the nets' output gradients and g_y = g_x e^{-s} as the chain; no loss term).  These
// serve the shapes whose tile image the any-shape MFMA kernel (rnvp_lmm.hip) cannot hold.
__global__ void __launch_bounds__(256)
k_generic_train(KShape s, const float *__restrict__ params, const uint8_t *__restrict__ masks,
                const float *__restrict__ x, const float *__restrict__ c,
                const int64_t *__restrict__ row_index, int64_t n, float inv_B,
                float *gpart, float *losspart, float *xsave, int TB, int TBP, Seeds sd) {
    const float *__restrict__ gz = sd.gz;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int t = threadIdx.x, d = s.d, cd = s.c, nthreads = blockDim.x;
    float *xcur = lds;
    float *uin = xcur + d * TBP;
    float *acts = uin + (d + cd) * TBP;
    float *tout = acts + s.hs * TBP;
    float *sout = tout + d * TBP;
    float *gy = sout + d * TBP;
    float *gxb = gy + d * TBP;
    float *gin = gxb + d * TBP;
    float *gA = gin + d * TBP;
    float *gB = gA + s.wmax * TBP;
    float *gcb = gB + s.wmax * TBP;                          // [cd][TBP]: d loss / d c of the tile's rows (KShape::gcw only)
    float *red = gcb + (s.gcw ? cd : 0) * TBP;
    float *const gcond = (s.gcw && cd > 0) ? gcb : nullptr;
    const bool inv = sd.inv != 0;
    const size_t P = (size_t)2 * s.npn * s.L;
    float *gp = gpart + (size_t)blockIdx.x * P;
    float *xs = xsave + (size_t)blockIdx.x * s.L * d * TB;
    const int64_t ntiles = (n + TB - 1) / TB;
    const float prior_c = 0.5f * (float)d * kLog2Pi;
    float block_sum = 0.f;
    bool first = true;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t row = tile * TB + t;
        const bool valid = (t < TB) && (row < n);
        float lp = 0.f;
        if (t < TB) {
            const int64_t src = valid ? (row_index ? row_index[row] : row) : 0;
            for (int j = 0; j < d; ++j) xcur[j * TBP + t] = valid ? x[src * d + j] : 0.f;
            for (int j = 0; j < cd; ++j) uin[(d + j) * TBP + t] = valid ? c[src * cd + j] : 0.f;
            float ld = 0.f;
            if (gcond)
                for (int j = 0; j < cd; ++j) gcond[j * TBP + t] = 0.f;
            for (int l = s.L - 1; inv && l >= 0; --l) {      // x = g(z, c), keeping every layer's OUTPUT (= the input of f's layer l)
                const uint8_t *m = masks + l * d;
                const float *pl = params + (size_t)l * 2 * s.npn;
                for (int j = 0; j < d; ++j) uin[j * TBP + t] = xcur[j * TBP + t] * (float)m[j];
                net_forward<false>(pl, s, uin, gA, gB, tout, TBP, t);
                net_forward<false>(pl + s.npn, s, uin, gA, gB, sout, TBP, t);
                for (int j = 0; j < d; ++j) {
                    if (!m[j]) xcur[j * TBP + t] = (xcur[j * TBP + t] - tout[j * TBP + t]) * expf(-sout[j * TBP + t]);
                    xs[(l * d + j) * TB + t] = xcur[j * TBP + t];
                }
            }
            for (int l = 0; !inv && l < s.L; ++l) {
                const uint8_t *m = masks + l * d;
                const float *pl = params + (size_t)l * 2 * s.npn;
                for (int j = 0; j < d; ++j) {
                    const float xv = xcur[j * TBP + t];
                    xs[(l * d + j) * TB + t] = xv;                 // layer input, for the backward
                    uin[j * TBP + t] = xv * (float)m[j];
                }
                net_forward<false>(pl, s, uin, gA, gB, tout, TBP, t);
                net_forward<false>(pl + s.npn, s, uin, gA, gB, sout, TBP, t);
                for (int j = 0; j < d; ++j) {
                    if (!m[j]) {
                        const float sv = sout[j * TBP + t];
                        xcur[j * TBP + t] = fmaf(xcur[j * TBP + t], expf(sv), tout[j * TBP + t]);
                        ld += sv;
                    }
                }
            }
            float ss = 0.f;
            for (int j = 0; j < d; ++j) { const float zv = xcur[j * TBP + t]; ss = fmaf(zv, zv, ss); }
            lp = inv ? 0.f : (gz ? ld : ld + (-0.5f * ss - prior_c));
            // seed: d(-mean logp)/dz = z / B   (zero for padding rows); with gz the caller's d loss / d z (another prior)
            for (int j = 0; j < d; ++j)
                gy[j * TBP + t] = valid ? (gz ? gz[row * d + j] : xcur[j * TBP + t] * inv_B) : 0.f;
        }
        red[t] = valid ? lp : 0.f;
        __syncthreads();
        if (t == 0) { float a = 0.f; for (int r = 0; r < TB; ++r) a += red[r]; block_sum += a; }
        const float gld = valid ? (sd.gld ? sd.gld[row] : -inv_B) : 0.f;      // d loss / d log_det (rnvp_backward: the caller's)
        for (int li = 0; li < s.L; ++li) {
            const int l = inv ? li : s.L - 1 - li;
            const uint8_t *m = masks + l * d;
            const float *pl = params + (size_t)l * 2 * s.npn;
            float *gl = gp + (size_t)l * 2 * s.npn;
            if (t < TB) {
                for (int j = 0; j < d; ++j) {
                    const float xv = xs[(l * d + j) * TB + t];
                    xcur[j * TBP + t] = xv;
                    uin[j * TBP + t] = xv * (float)m[j];
                    gin[j * TBP + t] = 0.f;
                }
                net_forward<true>(pl + s.npn, s, uin, acts, nullptr, sout, TBP, t);     // nn_s
                for (int j = 0; j < d; ++j) {
                    const float g = gy[j * TBP + t];
                    if (!m[j] && inv) {           // through g: y_u -> x_u = (y_u - t) e^{-s}; xcur holds x_u
                        gxb[j * TBP + t] = g * expf(-sout[j * TBP + t]);
                        gA[j * TBP + t] = -g * xcur[j * TBP + t];
                    } else if (!m[j]) {
                        const float es = expf(sout[j * TBP + t]);
                        gxb[j * TBP + t] = g * es;
                        gA[j * TBP + t] = fmaf(g * xcur[j * TBP + t], es, gld);
                    } else {
                        gxb[j * TBP + t] = g;
                        gA[j * TBP + t] = 0.f;
                    }
                }
            }
            net_backward(pl + s.npn, gl + s.npn, s, uin, acts, gA, gB, gin, TB, TBP, t, nthreads, first, gcond);
            if (t < TB) {
                net_forward<true>(pl, s, uin, acts, nullptr, tout, TBP, t);             // nn_t
                for (int j = 0; j < d; ++j)
                    gA[j * TBP + t] = m[j] ? 0.f : (inv ? -gy[j * TBP + t] * expf(-sout[j * TBP + t]) : gy[j * TBP + t]);
            }
            net_backward(pl, gl, s, uin, acts, gA, gB, gin, TB, TBP, t, nthreads, first, gcond);
            if (t < TB)
                for (int j = 0; j < d; ++j)
                    gy[j * TBP + t] = gxb[j * TBP + t] + (m[j] ? gin[j * TBP + t] : 0.f);
        }
        if (sd.gx && valid)                                           // rnvp_backward: d loss / d x of the batch rows (inv: d loss / d z)
            for (int j = 0; j < d; ++j) sd.gx[row * d + j] = gy[j * TBP + t];
        if (sd.gc && gcond && valid)
            for (int j = 0; j < cd; ++j) sd.gc[row * cd + j] = gcond[j * TBP + t];
        first = false;
        __syncthreads();
    }
    if (t == 0) losspart[blockIdx.x] = block_sum;
}

// grad[p] = sum_b gpart[b][p] in block order; loss = -(sum_b losspart[b]) * inv_B
__global__ void __launch_bounds__(256)
k_reduce_grad(const float *__restrict__ gpart, const float *__restrict__ losspart, int G, size_t P,
              float inv_B, float *grad, float *loss) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < P) {
        float a = 0.f;
        for (int b = 0; b < G; ++b) a += gpart[(size_t)b * P + p];
        grad[p] = a;
    }
}

size_t lds_floats_per_row(const KShape &k, int op) {
    if (op == RNVP_OP_TRAIN)
        return (size_t)k.d + (k.d + k.c) + k.hs + 2 * k.d + 3 * k.d + 2 * k.wmax + (k.gcw ? k.c : 0);
    return (size_t)k.d + (k.d + k.c) + 2 * k.hmax + 2 * k.d;
}

bool pick_tiling(const KShape &k, int op, int64_t n, Tiling *tl) {
    const size_t fpr = lds_floats_per_row(k, op);
    int best = 0;
    for (int TB = 256; TB >= 8; TB >>= 1) {
        const int threads = TB < 64 ? 64 : TB;
        const size_t bytes = (fpr * (TB + 1) + threads) * sizeof(float);
        const size_t limit = (TB > 64) ? kLdsSoft : kLdsHard;
        if (bytes <= limit) { best = TB; break; }
    }
    if (!best) return false;
    // small batches: do not make tiles wider than the batch needs (keeps >1 block busy)
    while (best > 64 && (int64_t)best / 2 >= n) best >>= 1;
    tl->TB = best; tl->TBP = best + 1; tl->threads = best < 64 ? 64 : best;
    tl->lds = (fpr * (best + 1) + tl->threads) * sizeof(float);
    return true;
}

template <typename K>
int allow_lds(K kernel, size_t bytes, std::atomic<uint64_t> &done) {
    if (bytes <= 48 * 1024) return RNVP_OK;
    return allow_big_lds(reinterpret_cast<const void *>(kernel), (int)kLdsHard, done);
}

std::atomic<uint64_t> g_lds_fwd{0}, g_lds_inv{0}, g_lds_train{0};

int grid_for(int64_t n, int TB, int cap) {
    const int64_t ntiles = (n + TB - 1) / TB;
    return (int)(ntiles < cap ? ntiles : cap);
}

}  // namespace

// grad_out[p] = sum_b gpart[b][p] (skipped when gpart is NULL); loss_out = loss_scale * sum_b losspart[b]
int generic_reduce_partials(hipStream_t st, const float *gpart, const float *losspart, int G, size_t P,
                            float loss_scale, float *grad_out, float *loss_out) {
    if (gpart && grad_out) {
        const int rb = 256;
        hipLaunchKernelGGL(k_reduce_grad, dim3((unsigned)((P + rb - 1) / rb)), dim3(rb), 0, st, gpart, losspart, G, P,
                           loss_scale, grad_out, loss_out);
        RNVP_HIP_TRY(hipGetLastError());
    }
    if (loss_out) {
        hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(64), 0, st, losspart, G, loss_scale, loss_out);
        RNVP_HIP_TRY(hipGetLastError());
    }
    return RNVP_OK;
}

size_t generic_workspace_bytes(const KShape &k, int op, int64_t max_rows) {
    Tiling tl;
    if (max_rows < 1) max_rows = 1;
    if (!pick_tiling(k, op, max_rows, &tl)) return 0;
    if (op == RNVP_OP_TRAIN) {
        // a call with fewer rows may pick a narrower tile (more blocks): bound by the cap
        const size_t G = kMaxGridTrain;
        const size_t P = (size_t)2 * k.npn * k.L;
        return align_up(G * P * sizeof(float), 256) + align_up(G * sizeof(float), 256) +
               align_up(G * (size_t)k.L * k.d * 256 * sizeof(float), 256);
    }
    if (op == RNVP_OP_FORWARD) return align_up(kMaxGrid * sizeof(float), 256);
    return 0;
}

int generic_forward(hipStream_t st, const KShape &k, const float *params, const uint8_t *masks,
                    const float *x, const float *c, const int64_t *row_index, int64_t n,
                    float *z_out, float *logdet_out, float *logp_out, float *logp_sum,
                    void *ws, size_t ws_bytes) {
    Tiling tl;
    if (!pick_tiling(k, RNVP_OP_FORWARD, n, &tl)) return RNVP_EUNSUPPORTED;
    float *part = nullptr;
    if (logp_sum) {
        if (!ws || ws_bytes < generic_workspace_bytes(k, RNVP_OP_FORWARD, n)) return RNVP_EWORKSPACE;
        part = static_cast<float *>(ws);
    }
    int rc = allow_lds(k_generic_forward, tl.lds, g_lds_fwd);
    if (rc) return rc;
    const int G = grid_for(n, tl.TB, kMaxGrid);
    note_dispatch(RNVP_PROFILE_FORWARD, "k_generic_forward", RNVP_VARIANT_VALU, 0, tl.threads / 64, G, RNVP_PREC_F32, n);
    hipLaunchKernelGGL(k_generic_forward, dim3(G), dim3(tl.threads), tl.lds, st, k, params, masks, x, c,
                       row_index, n, z_out, logdet_out, logp_out, part, tl.TB, tl.TBP);
    RNVP_HIP_TRY(hipGetLastError());
    if (logp_sum) {
        hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(64), 0, st, part, G, 1.0f, logp_sum);
        RNVP_HIP_TRY(hipGetLastError());
    }
    return RNVP_OK;
}

int generic_inverse(hipStream_t st, const KShape &k, const float *params, const uint8_t *masks,
                    const float *z, const float *c, int64_t n, float *x_out) {
    Tiling tl;
    if (!pick_tiling(k, RNVP_OP_INVERSE, n, &tl)) return RNVP_EUNSUPPORTED;
    int rc = allow_lds(k_generic_inverse, tl.lds, g_lds_inv);
    if (rc) return rc;
    const int G = grid_for(n, tl.TB, kMaxGrid);
    note_dispatch(RNVP_PROFILE_INVERSE, "k_generic_inverse", RNVP_VARIANT_VALU, 0, tl.threads / 64, G, RNVP_PREC_F32, n);
    hipLaunchKernelGGL(k_generic_inverse, dim3(G), dim3(tl.threads), tl.lds, st, k, params, masks, z, c, n,
                       x_out, tl.TB, tl.TBP);
    RNVP_HIP_TRY(hipGetLastError());
    return RNVP_OK;
}

int generic_loss_grad(hipStream_t st, const KShape &k, const float *params, const uint8_t *masks,
                      const float *x, const float *c, const int64_t *row_index, int64_t n,
                      float inv_B, float *grad_out, float *loss_out, void *ws, size_t ws_bytes, Seeds sd) {
    Tiling tl;
    if (!pick_tiling(k, RNVP_OP_TRAIN, n, &tl)) return RNVP_EUNSUPPORTED;
    if (!ws || ws_bytes < generic_workspace_bytes(k, RNVP_OP_TRAIN, n)) return RNVP_EWORKSPACE;
    const size_t P = (size_t)2 * k.npn * k.L;
    const int G = grid_for(n, tl.TB, kMaxGridTrain);
    char *w = static_cast<char *>(ws);
    float *gpart = reinterpret_cast<float *>(w);
    w += align_up((size_t)kMaxGridTrain * P * sizeof(float), 256);
    float *losspart = reinterpret_cast<float *>(w);
    w += align_up((size_t)kMaxGridTrain * sizeof(float), 256);
    float *xsave = reinterpret_cast<float *>(w);
    int rc = allow_lds(k_generic_train, tl.lds, g_lds_train);
    if (rc) return rc;
    note_dispatch(RNVP_PROFILE_TRAIN, "k_generic_train", RNVP_VARIANT_VALU, 0, tl.threads / 64, G, RNVP_PREC_F32, n);
    {
        KernelTimer timer(st, RNVP_PROFILE_TRAIN);      // rnvp_profile_*: brackets exactly this launch when enabled
        hipLaunchKernelGGL(k_generic_train, dim3(G), dim3(tl.threads), tl.lds, st, k, params, masks, x, c,
                           row_index, n, inv_B, gpart, losspart, xsave, tl.TB, tl.TBP, sd);
    }
    RNVP_HIP_TRY(hipGetLastError());
    const int rb = 256;
    const unsigned blocks = (unsigned)((P + 1 + rb - 1) / rb);
    hipLaunchKernelGGL(k_reduce_grad, dim3(blocks), dim3(rb), 0, st, gpart, losspart, G, P, inv_B, grad_out,
                       loss_out);
    RNVP_HIP_TRY(hipGetLastError());
    if (loss_out) {
        hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(64), 0, st, losspart, G, -inv_B, loss_out);
        RNVP_HIP_TRY(hipGetLastError());
    }
    return RNVP_OK;
}

}  // namespace rnvp
