// cvae_generic.hip -- conditional VAE (encoder / reparameterize / decoder / KL + MSE loss / backward)
// for gfx950, built from the same "one thread = one row, row state in LDS" MLP blocks as the generic
// RealNVP path (rnvp_generic_net.h).  C ABI: include/cvae_hip.h.
#include <atomic>

#include "../../include/cvae_hip.h"
#include "rnvp_common.h"
#include "rnvp_generic_net.h"
#include "rnvp_lmm.h"
#include "rnvp_resident.h"

namespace rnvp {
namespace {

constexpr int kMaxGrid = 512, kMaxGridTrain = 256;
constexpr size_t kLdsHard = 160 * 1024 - 1024;

int make_mlp(int nin, const int32_t *hidden, int nh, int nout, int act, int dgrad, KShape *k) {
    std::memset(k, 0, sizeof(*k));
    k->L = 1; k->nh = nh; k->act = act == 0 ? RNVP_ACT_TANH : RNVP_ACT_RELU; k->alt = 0;
    k->d = dgrad; k->c = nin - dgrad;
    int in = nin, off = 0;
    k->wmax = nin > nout ? nin : nout;
    for (int i = 0; i <= nh; ++i) {
        const int out = (i < nh) ? hidden[i] : nout;
        if (out < 1) return RNVP_EINVAL;
        k->nin[i] = in; k->nout[i] = out; k->woff[i] = off; k->boff[i] = off + out * in;
        off += out * in + out;
        if (i < nh) { k->hs += out; if (out > k->hmax) k->hmax = out; }
        if (out > k->wmax) k->wmax = out;
        in = out;
    }
    k->npn = off;
    return RNVP_OK;
}

int make_cvae(const cvae_shape *s, CvaeK *k) {
    if (!s || s->d < 1 || s->c < 0 || s->lat < 1 || s->n_hidden < 1 || s->n_hidden > 8) return RNVP_EINVAL;
    if (s->family < RNVP_FAMILY_AUTO || s->family > RNVP_FAMILY_LMM) return RNVP_EINVAL;
    k->d = s->d; k->c = s->c; k->lat = s->lat;
    int rc = make_mlp(s->d + s->c, s->hidden, s->n_hidden, 2 * s->lat, s->act, 0, &k->enc);
    if (rc) return rc;
    rc = make_mlp(s->lat + s->c, s->hidden, s->n_hidden, s->d, s->act, s->lat, &k->dec);
    k->pe = k->enc.npn;
    return rc;
}

// ---- fused training / loss kernel -------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_cvae_train(CvaeK s, const float *__restrict__ params, const float *__restrict__ x,
             const float *__restrict__ c, const int64_t *__restrict__ row_index,
             const float *__restrict__ eps, int64_t n, float inv_B, float klw,
             float *gpart, float *losspart, int TB, int TBP, int do_grad) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int t = threadIdx.x, d = s.d, cd = s.c, lat = s.lat, nthreads = blockDim.x;
    const int wm = s.enc.wmax > s.dec.wmax ? s.enc.wmax : s.dec.wmax;
    float *ein = lds;                                  // [d + c]
    float *ea = ein + (d + cd) * TBP;                  // [hs]   encoder activations (kept for backward)
    float *eo = ea + s.enc.hs * TBP;                   // [2 lat] mu | log_sigma
    float *din = eo + 2 * lat * TBP;                   // [lat + c]
    float *da = din + (lat + cd) * TBP;                // [hs]   decoder activations
    float *xr = da + s.dec.hs * TBP;                   // [d]    reconstruction
    float *epb = xr + d * TBP;                         // [lat]
    float *gz = epb + lat * TBP;                       // [lat]  d loss / d z
    float *gA = gz + lat * TBP;                        // [wm]
    float *gB = gA + wm * TBP;                         // [wm]
    float *red = gB + wm * TBP;                        // [nthreads]
    const size_t P = (size_t)s.enc.npn + s.dec.npn;
    float *gp = gpart + (size_t)blockIdx.x * P;
    const int64_t ntiles = (n + TB - 1) / TB;
    float block_sum = 0.f;
    bool first = true;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t row = tile * TB + t;
        const bool valid = (t < TB) && (row < n);
        float lrow = 0.f;
        if (t < TB) {
            const int64_t src = valid ? (row_index ? row_index[row] : row) : 0;
            for (int j = 0; j < d; ++j) ein[j * TBP + t] = valid ? x[src * d + j] : 0.f;            // cat(X, C), cvae.py:58
            for (int j = 0; j < cd; ++j) {
                const float cv = valid ? c[src * cd + j] : 0.f;
                ein[(d + j) * TBP + t] = cv; din[(lat + j) * TBP + t] = cv;
            }
            for (int j = 0; j < lat; ++j) epb[j * TBP + t] = valid ? eps[row * lat + j] : 0.f;
            net_forward<true>(params, s.enc, ein, ea, nullptr, eo, TBP, t);                          // mu | log_sigma
            for (int j = 0; j < lat; ++j)                                                            // sample_z, cvae.py:188
                din[j * TBP + t] = fmaf(expf(0.5f * eo[(lat + j) * TBP + t]), epb[j * TBP + t], eo[j * TBP + t]);
            net_forward<true>(params + s.pe, s.dec, din, da, nullptr, xr, TBP, t);                   // decoder
            float kl = 0.f, se = 0.f;
            for (int j = 0; j < lat; ++j) {
                const float mu = eo[j * TBP + t], ls = eo[(lat + j) * TBP + t];
                kl += 1.f + ls - mu * mu - expf(ls);                                                 // cvae.py:191
            }
            for (int j = 0; j < d; ++j) { const float df = ein[j * TBP + t] - xr[j * TBP + t]; se = fmaf(df, df, se); }
            lrow = klw * (-0.5f * kl) + se / (float)d;
        }
        red[t] = valid ? lrow : 0.f;
        __syncthreads();
        if (t == 0) { float a = 0.f; for (int r = 0; r < TB; ++r) a += red[r]; block_sum += a; }
        if (do_grad) {
            const float sc = valid ? inv_B : 0.f;
            if (t < TB) {
                for (int j = 0; j < lat; ++j) gz[j * TBP + t] = 0.f;
                for (int j = 0; j < d; ++j)                                                          // d MSE / d x_rec
                    gA[j * TBP + t] = (2.f * sc / (float)d) * (xr[j * TBP + t] - ein[j * TBP + t]);
            }
            net_backward(params + s.pe, gp + s.pe, s.dec, din, da, gA, gB, gz, TB, TBP, t, nthreads, first);
            if (t < TB) {
                for (int j = 0; j < lat; ++j) {
                    const float mu = eo[j * TBP + t], ls = eo[(lat + j) * TBP + t], g = gz[j * TBP + t];
                    gA[j * TBP + t] = fmaf(klw * sc, mu, g);                                         // d/d mu
                    gA[(lat + j) * TBP + t] = g * epb[j * TBP + t] * 0.5f * expf(0.5f * ls)          // d/d log_sigma
                                              + klw * sc * (-0.5f) * (1.f - expf(ls));
                }
            }
            net_backward(params, gp, s.enc, ein, ea, gA, gB, gz, TB, TBP, t, nthreads, first);
            first = false;
        }
        __syncthreads();
    }
    if (t == 0) losspart[blockIdx.x] = block_sum;
}

template <bool ENCODE>
__global__ void __launch_bounds__(256)
k_cvae_mlp(CvaeK s, const float *__restrict__ params, const float *__restrict__ a, const float *__restrict__ c,
           int64_t n, float *out0, float *out1, int TB, int TBP) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const KShape &m = ENCODE ? s.enc : s.dec;
    const int t = threadIdx.x, na = ENCODE ? s.d : s.lat, cd = s.c, no = m.nout[m.nh];
    float *in = lds;
    float *h0 = in + (na + cd) * TBP;
    float *h1 = h0 + m.hmax * TBP;
    float *o = h1 + m.hmax * TBP;
    const float *p = ENCODE ? params : params + s.pe;
    const int64_t ntiles = (n + TB - 1) / TB;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t row = tile * TB + t;
        if (t >= TB || row >= n) continue;
        for (int j = 0; j < na; ++j) in[j * TBP + t] = a[row * na + j];
        for (int j = 0; j < cd; ++j) in[(na + j) * TBP + t] = c[row * cd + j];
        net_forward<false>(p, m, in, h0, h1, o, TBP, t);
        if (ENCODE) {
            for (int j = 0; j < s.lat; ++j) { out0[row * s.lat + j] = o[j * TBP + t]; out1[row * s.lat + j] = o[(s.lat + j) * TBP + t]; }
        } else {
            for (int j = 0; j < no; ++j) out0[row * no + j] = o[j * TBP + t];
        }
    }
}

size_t train_floats_per_row(const CvaeK &k) {
    const int wm = k.enc.wmax > k.dec.wmax ? k.enc.wmax : k.dec.wmax;
    return (size_t)(k.d + k.c) + k.enc.hs + 2 * k.lat + (k.lat + k.c) + k.dec.hs + k.d + 2 * k.lat + 2 * wm;
}

bool pick_tb(size_t fpr, int64_t n, int *TB, size_t *lds) {
    for (int tb = 256; tb >= 8; tb >>= 1) {
        const int threads = tb < 64 ? 64 : tb;
        const size_t bytes = (fpr * (tb + 1) + threads) * sizeof(float);
        if (bytes <= ((tb > 64) ? (size_t)64 * 1024 : kLdsHard)) {
            int best = tb;
            while (best > 64 && (int64_t)best / 2 >= n) best >>= 1;
            *TB = best; *lds = (fpr * (best + 1) + (best < 64 ? 64 : best)) * sizeof(float);
            return true;
        }
    }
    return false;
}

std::atomic<uint64_t> g_attr_train{0}, g_attr_enc{0}, g_attr_dec{0};
template <typename K> int allow(K kern, std::atomic<uint64_t> &done) {
    return allow_big_lds(reinterpret_cast<const void *>(kern), (int)kLdsHard, done);
}

}  // namespace
}  // namespace rnvp

using namespace rnvp;

extern "C" {

size_t cvae_param_count(const cvae_shape *shape) {
    CvaeK k;
    if (make_cvae(shape, &k) != RNVP_OK) return 0;
    return (size_t)k.enc.npn + k.dec.npn;
}

// shape->family (cvae_hip.h): 0 = the register-chained MFMA kernels where the shape allows, else the any-shape MFMA kernels
// (rnvp_lmm.hip) while a tile's LDS image fits, else one thread per row; 1 pins the latter, 2 the any-shape MFMA kernels
static bool use_mfma(const cvae_shape *shape) { return shape->family == RNVP_FAMILY_AUTO && cvae_mfma::supported(shape); }
static bool use_lmm(const cvae_shape *shape, const CvaeK &k, int op) {
    return shape->family != RNVP_FAMILY_VALU && !use_mfma(shape) && lmm::cvae_fits(k, op);
}

int cvae_kernel_path(const cvae_shape *shape) {
    CvaeK k;
    if (make_cvae(shape, &k) != RNVP_OK) return RNVP_EINVAL;
    if (use_mfma(shape)) return RNVP_PATH_MFMA;
    return use_lmm(shape, k, RNVP_OP_TRAIN) ? RNVP_PATH_LMM : RNVP_PATH_GENERIC;
}

static size_t generic_cvae_workspace(const CvaeK &k) {
    const size_t P = (size_t)k.enc.npn + k.dec.npn;
    return align_up((size_t)kMaxGridTrain * P * sizeof(float), 256) + align_up(kMaxGridTrain * sizeof(float), 256) + 256;
}

size_t cvae_workspace_bytes(const cvae_shape *shape, int64_t max_rows) {
    CvaeK k;
    if (make_cvae(shape, &k) != RNVP_OK) return 0;
    size_t b = generic_cvae_workspace(k);                       // any path may run (shape->family)
    if (cvae_mfma::supported(shape)) { const size_t m = cvae_mfma::workspace_bytes(shape, max_rows) + 256; if (m > b) b = m; }
    if (lmm::cvae_fits(k, RNVP_OP_TRAIN)) { const size_t m = lmm::cvae_workspace_bytes(k, max_rows) + 256; if (m > b) b = m; }
    return b;
}

int cvae_loss_grad(void *stream, const cvae_shape *shape, const float *params, const float *x, const float *c,
                   const int64_t *row_index, const float *eps, int64_t n_rows, float inv_B, float kl_weight,
                   float *grad_out, float *loss_out, void *workspace, size_t workspace_bytes) {
    CvaeK k;
    int rc = make_cvae(shape, &k);
    if (rc) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t P = (size_t)k.enc.npn + k.dec.npn;
    if (n_rows < 0) return RNVP_EINVAL;
    if (n_rows == 0) {
        if (grad_out) RNVP_HIP_TRY(hipMemsetAsync(grad_out, 0, P * sizeof(float), st));
        if (loss_out) RNVP_HIP_TRY(hipMemsetAsync(loss_out, 0, sizeof(float), st));
        return RNVP_OK;
    }
    if (!params || !x || (k.c > 0 && !c) || !eps) return RNVP_EINVAL;
    if (!workspace || workspace_bytes < cvae_workspace_bytes(shape, n_rows)) return RNVP_EWORKSPACE;
    if (use_mfma(shape))
        return cvae_mfma::loss_grad(st, shape, params, x, c, row_index, eps, n_rows, inv_B, kl_weight, grad_out, loss_out,
                                    workspace, workspace_bytes);
    if (use_lmm(shape, k, RNVP_OP_TRAIN))
        return lmm::cvae_loss_grad(st, k, params, x, c, row_index, eps, n_rows, inv_B, kl_weight, grad_out, loss_out, workspace,
                                   workspace_bytes);
    int TB; size_t lds;
    if (!pick_tb(train_floats_per_row(k), n_rows, &TB, &lds)) return RNVP_EUNSUPPORTED;
    rc = allow(k_cvae_train, g_attr_train);
    if (rc) return rc;
    char *w = static_cast<char *>(workspace);
    float *gpart = reinterpret_cast<float *>(w);
    float *losspart = reinterpret_cast<float *>(w + align_up((size_t)kMaxGridTrain * P * sizeof(float), 256));
    const int64_t ntiles = (n_rows + TB - 1) / TB;
    const int G = (int)(ntiles < kMaxGridTrain ? ntiles : kMaxGridTrain);
    const int threads = TB < 64 ? 64 : TB;
    {
        KernelTimer timer(st, RNVP_PROFILE_TRAIN);
        hipLaunchKernelGGL(k_cvae_train, dim3(G), dim3(threads), lds, st, k, params, x, c, row_index, eps, n_rows, inv_B,
                           kl_weight, gpart, losspart, TB, TB + 1, grad_out ? 1 : 0);
    }
    RNVP_HIP_TRY(hipGetLastError());
    return generic_reduce_partials(st, grad_out ? gpart : nullptr, losspart, G, P, inv_B, grad_out, loss_out);
}

int cvae_train_step(void *stream, const cvae_shape *shape, float *params, const float *x, const float *c,
                    const int64_t *row_index, const float *eps, int64_t n_rows, float inv_B, float kl_weight,
                    float *grad_buf, float *loss_out, float *exp_avg, float *exp_avg_sq,
                    double lr, double beta1, double beta2, double adam_eps, double weight_decay, int64_t step,
                    void *workspace, size_t workspace_bytes) {
    CvaeK k;
    int rc = make_cvae(shape, &k);
    if (rc) return rc;
    if (!grad_buf || !exp_avg || !exp_avg_sq || step < 1) return RNVP_EINVAL;
    if (n_rows > 0 && use_mfma(shape)) {
        if (!params || !x || (k.c > 0 && !c) || !eps) return RNVP_EINVAL;
        if (!workspace || workspace_bytes < cvae_workspace_bytes(shape, n_rows)) return RNVP_EWORKSPACE;
        return cvae_mfma::train_step(static_cast<hipStream_t>(stream), shape, params, x, c, row_index, eps, n_rows, inv_B,
                                     kl_weight, grad_buf, loss_out, exp_avg, exp_avg_sq,
                                     make_adam(lr, beta1, beta2, adam_eps, weight_decay, step), workspace, workspace_bytes);
    }
    rc = cvae_loss_grad(stream, shape, params, x, c, row_index, eps, n_rows, inv_B, kl_weight, grad_buf, loss_out, workspace,
                        workspace_bytes);
    if (rc) return rc;
    return rnvp_adam_step(stream, params, grad_buf, exp_avg, exp_avg_sq, (int64_t)cvae_param_count(shape), lr, beta1, beta2,
                          adam_eps, weight_decay, step);
}

int cvae_fit_epoch_resident(const cvae_shape *shape, int64_t batch_size) {
    CvaeK k;
    if (make_cvae(shape, &k) != RNVP_OK) return 0;
    return resident::cvae_fits(k, shape->family, batch_size) ? 1 : 0;
}

int cvae_fit_epoch(void *stream, const cvae_shape *shape, float *params, const float *x, const float *c, const int64_t *perm,
                   const float *eps, int64_t n, int64_t batch_size, float kl_weight, float *grad_buf, float *loss_hist,
                   float *exp_avg, float *exp_avg_sq, double lr, double beta1, double beta2, double adam_eps, double weight_decay,
                   int64_t first_step, void *workspace, size_t workspace_bytes) {
    CvaeK k;
    int rc = make_cvae(shape, &k);
    if (rc) return rc;
    if (n < 0 || batch_size < 1 || first_step < 1 || !loss_hist || !exp_avg || !exp_avg_sq) return RNVP_EINVAL;
    if (n == 0) return RNVP_OK;
    if (!params || !x || (k.c > 0 && !c) || !perm || !eps) return RNVP_EINVAL;
    if (resident::cvae_fits(k, shape->family, batch_size))
        return resident::cvae_fit_epoch(static_cast<hipStream_t>(stream), k, params, x, c, perm, eps, n, batch_size, kl_weight,
                                        loss_hist, exp_avg, exp_avg_sq, lr, beta1, beta2, adam_eps, weight_decay, first_step);
    // register-chained step kernel: the packed fragments live in the workspace for the whole call (packed before the first batch,
    // re-packed by every step's finish kernel)
    const bool chained = use_mfma(shape);
    if (chained && (!workspace || workspace_bytes < cvae_workspace_bytes(shape, batch_size < n ? batch_size : n))) return RNVP_EWORKSPACE;
    if (chained && !grad_buf) return RNVP_EINVAL;
    bool packed_valid = false;
    int64_t kb = 0;
    for (int64_t s0 = 0; s0 < n; s0 += batch_size, ++kb) {
        const int64_t rows = (n - s0 < batch_size) ? n - s0 : batch_size;
        if (chained) {
            rc = cvae_mfma::train_step(static_cast<hipStream_t>(stream), shape, params, x, c, perm + s0, eps + s0 * k.lat, rows,
                                       1.0f / (float)rows, kl_weight, grad_buf, loss_hist + kb, exp_avg, exp_avg_sq,
                                       make_adam(lr, beta1, beta2, adam_eps, weight_decay, first_step + kb), workspace, workspace_bytes,
                                       packed_valid, s0 + rows < n);
            packed_valid = true;
        } else {
            rc = cvae_train_step(stream, shape, params, x, c, perm + s0, eps + s0 * k.lat, rows, 1.0f / (float)rows, kl_weight, grad_buf,
                                 loss_hist + kb, exp_avg, exp_avg_sq, lr, beta1, beta2, adam_eps, weight_decay, first_step + kb, workspace,
                                 workspace_bytes);
        }
        if (rc) return rc;
    }
    return RNVP_OK;
}

int cvae_decode(void *stream, const cvae_shape *shape, const float *params, const float *z, const float *c,
                int64_t n_rows, float *x_out, void *workspace, size_t workspace_bytes) {
    CvaeK k;
    int rc = make_cvae(shape, &k);
    if (rc) return rc;
    if (n_rows < 0) return RNVP_EINVAL;
    if (n_rows == 0) return RNVP_OK;
    if (!params || !z || (k.c > 0 && !c) || !x_out) return RNVP_EINVAL;
    if (workspace && use_mfma(shape))
        return cvae_mfma::forward(static_cast<hipStream_t>(stream), shape, params, false, z, c, n_rows, x_out, nullptr,
                                  workspace, workspace_bytes);
    if (workspace && use_lmm(shape, k, RNVP_OP_INVERSE))
        return lmm::cvae_forward(static_cast<hipStream_t>(stream), k, params, false, z, c, n_rows, x_out, nullptr, workspace,
                                 workspace_bytes);
    int TB; size_t lds;
    const size_t fpr = (size_t)(k.lat + k.c) + 2 * k.dec.hmax + k.d;
    if (!pick_tb(fpr, n_rows, &TB, &lds)) return RNVP_EUNSUPPORTED;
    rc = allow(k_cvae_mlp<false>, g_attr_dec);
    if (rc) return rc;
    const int64_t ntiles = (n_rows + TB - 1) / TB;
    hipLaunchKernelGGL((k_cvae_mlp<false>), dim3((unsigned)(ntiles < kMaxGrid ? ntiles : kMaxGrid)), dim3(TB < 64 ? 64 : TB), lds,
                       static_cast<hipStream_t>(stream), k, params, z, c, n_rows, x_out, nullptr, TB, TB + 1);
    RNVP_HIP_TRY(hipGetLastError());
    return RNVP_OK;
}

int cvae_encode(void *stream, const cvae_shape *shape, const float *params, const float *x, const float *c,
                int64_t n_rows, float *mu_out, float *log_sigma_out, void *workspace, size_t workspace_bytes) {
    CvaeK k;
    int rc = make_cvae(shape, &k);
    if (rc) return rc;
    if (n_rows < 0) return RNVP_EINVAL;
    if (n_rows == 0) return RNVP_OK;
    if (!params || !x || (k.c > 0 && !c) || !mu_out || !log_sigma_out) return RNVP_EINVAL;
    if (workspace && use_mfma(shape))
        return cvae_mfma::forward(static_cast<hipStream_t>(stream), shape, params, true, x, c, n_rows, mu_out, log_sigma_out,
                                  workspace, workspace_bytes);
    if (workspace && use_lmm(shape, k, RNVP_OP_FORWARD))
        return lmm::cvae_forward(static_cast<hipStream_t>(stream), k, params, true, x, c, n_rows, mu_out, log_sigma_out, workspace,
                                 workspace_bytes);
    int TB; size_t lds;
    const size_t fpr = (size_t)(k.d + k.c) + 2 * k.enc.hmax + 2 * k.lat;
    if (!pick_tb(fpr, n_rows, &TB, &lds)) return RNVP_EUNSUPPORTED;
    rc = allow(k_cvae_mlp<true>, g_attr_enc);
    if (rc) return rc;
    const int64_t ntiles = (n_rows + TB - 1) / TB;
    hipLaunchKernelGGL((k_cvae_mlp<true>), dim3((unsigned)(ntiles < kMaxGrid ? ntiles : kMaxGrid)), dim3(TB < 64 ? 64 : TB), lds,
                       static_cast<hipStream_t>(stream), k, params, x, c, n_rows, mu_out, log_sigma_out, TB, TB + 1);
    RNVP_HIP_TRY(hipGetLastError());
    return RNVP_OK;
}

}  // extern "C"
