// rnvp_api.hip -- extern "C" entry points of librnvp_hip.so (declared in include/rnvp_hip.h)
// and the dispatch between the kernel families.
#include <cstring>
#include <mutex>
#include <vector>

#include "rnvp_common.h"
#include "rnvp_lmm.h"
#include "rnvp_resident.h"
#include "rnvp_mfma.h"

using namespace rnvp;

namespace {

bool bad_ptrs(const KShape &k, const float *params, const uint8_t *masks, const float *x, const float *c) {
    return !params || (!masks && !k.alt) || !x || (k.c > 0 && !c);
}

}  // namespace

// ---- rnvp_profile_*: event pairs around the hot kernels ------------------------------------------
namespace rnvp {
namespace {
constexpr int kKinds = 3;
std::mutex g_ev_mu;
std::vector<hipEvent_t> g_ev[kKinds];   // per kind: start0, stop0, start1, stop1, ...
int g_ev_used[kKinds] = {0, 0, 0};      // pairs recorded since the last read
}  // namespace

KernelTimer::KernelTimer(hipStream_t s, int kind_) : st(s), kind(kind_), slot(-1) {
    if (kind < 0 || kind >= kKinds) return;
    std::lock_guard<std::mutex> lk(g_ev_mu);
    if (!g_ev[kind].empty() && (size_t)(2 * g_ev_used[kind] + 1) < g_ev[kind].size()) {
        if (hipEventRecord(g_ev[kind][2 * g_ev_used[kind]], st) == hipSuccess) slot = g_ev_used[kind]++;
    }
}
KernelTimer::~KernelTimer() {
    if (slot >= 0) {
        std::lock_guard<std::mutex> lk(g_ev_mu);
        if ((size_t)(2 * slot + 1) < g_ev[kind].size()) (void)hipEventRecord(g_ev[kind][2 * slot + 1], st);
    }
}
KernelEvents::KernelEvents(int kind) {
    if (kind < 0 || kind >= kKinds) return;
    std::lock_guard<std::mutex> lk(g_ev_mu);
    if (!g_ev[kind].empty() && (size_t)(2 * g_ev_used[kind] + 1) < g_ev[kind].size()) {
        start = g_ev[kind][2 * g_ev_used[kind]];
        stop = g_ev[kind][2 * g_ev_used[kind] + 1];
        ++g_ev_used[kind];
    }
}

namespace {
thread_local rnvp_dispatch g_last[kKinds];
}
void note_dispatch(int kind, const char *kernel, int variant, int row_tiles, int waves, int grid, int gemm1_fwd, int64_t rows) {
    if (kind < 0 || kind >= kKinds) return;
    rnvp_dispatch &d = g_last[kind];
    d.variant = variant; d.row_tiles = row_tiles; d.waves = waves; d.grid = grid; d.gemm1_fwd = gemm1_fwd;
    d.launches = 0; d.rows = rows;
    std::strncpy(d.kernel, kernel, sizeof(d.kernel) - 1);
    d.kernel[sizeof(d.kernel) - 1] = 0;
}
void note_launches(int kind, int launches) {
    if (kind >= 0 && kind < kKinds) g_last[kind].launches = launches;
}
}  // namespace rnvp

extern "C" {

int rnvp_last_dispatch(int kind, rnvp_dispatch *out) {
    if (kind < 0 || kind >= rnvp::kKinds || !out) return RNVP_EINVAL;
    *out = rnvp::g_last[kind];
    return RNVP_OK;
}

int rnvp_profile_enable(int capacity) {
    std::lock_guard<std::mutex> lk(rnvp::g_ev_mu);
    for (int kd = 0; kd < rnvp::kKinds; ++kd) {
        for (hipEvent_t e : rnvp::g_ev[kd]) (void)hipEventDestroy(e);
        rnvp::g_ev[kd].clear();
        rnvp::g_ev_used[kd] = 0;
        for (int i = 0; i < 2 * capacity; ++i) {
            hipEvent_t e;
            RNVP_HIP_TRY(hipEventCreate(&e));
            rnvp::g_ev[kd].push_back(e);
        }
    }
    return RNVP_OK;
}

int rnvp_profile_read(int kind, int *n_launches, float *total_ms) {
    if (kind < 0 || kind >= rnvp::kKinds) return RNVP_EINVAL;
    std::lock_guard<std::mutex> lk(rnvp::g_ev_mu);
    // A slot is taken before its launch is known to have succeeded: a failed launch (or another thread's launch still
    // between its two records) leaves a pair whose stop event was never recorded.  Such pairs are skipped, and the table
    // is reset whatever happened, so one bad pair cannot poison every later read.
    float tot = 0.f;
    int good = 0;
    for (int i = 0; i < rnvp::g_ev_used[kind]; ++i) {
        float ms = 0.f;
        if (hipEventSynchronize(rnvp::g_ev[kind][2 * i + 1]) != hipSuccess ||
            hipEventElapsedTime(&ms, rnvp::g_ev[kind][2 * i], rnvp::g_ev[kind][2 * i + 1]) != hipSuccess) {
            (void)hipGetLastError();
            continue;
        }
        tot += ms;
        ++good;
    }
    if (n_launches) *n_launches = good;
    if (total_ms) *total_ms = tot;
    rnvp::g_ev_used[kind] = 0;
    return RNVP_OK;
}

int rnvp_version(void) { return RNVP_HIP_VERSION; }

const char *rnvp_status_string(int status) {
    switch (status) {
        case RNVP_OK: return "ok";
        case RNVP_EINVAL: return "invalid argument (shape, NULL pointer, alignment or size)";
        case RNVP_EUNSUPPORTED: return "shape too large for the LDS-resident kernels";
        case RNVP_EWORKSPACE: return "workspace missing or smaller than rnvp_workspace_bytes()";
        default: return status > 0 ? hipGetErrorString((hipError_t)status) : "unknown status";
    }
}

size_t rnvp_param_count(const rnvp_shape *shape) {
    KShape k;
    if (make_kshape(shape, &k) != RNVP_OK) return 0;
    return (size_t)2 * k.npn * k.L;
}

int rnvp_kernel_path(const rnvp_shape *shape, const uint8_t *host_masks, int op) {
    KShape k;
    if (make_kshape(shape, &k) != RNVP_OK) return RNVP_EINVAL;
    if (host_masks && k.alt) {          // a declared pattern must agree with the table, if one is given
        for (int l = 0; l < k.L; ++l)
            for (int j = 0; j < k.d; ++j)
                if (host_masks[(size_t)l * k.d + j] != (uint8_t)((j + l + k.alt - 1) & 1)) return RNVP_EINVAL;
    }
    if (op == RNVP_OP_TRAIN ? mfma::train_supported(k) : mfma::supported(k)) return RNVP_PATH_MFMA;
    return lmm::use_lmm(k, op) ? RNVP_PATH_LMM : RNVP_PATH_GENERIC;
}

size_t rnvp_workspace_bytes(const rnvp_shape *shape, int op, int64_t max_rows) {
    KShape k;
    if (make_kshape(shape, &k) != RNVP_OK) return 0;
    const bool use_mfma = (op == RNVP_OP_TRAIN) ? mfma::train_supported(k) : mfma::supported(k);
    const size_t other = lmm::use_lmm(k, op) ? lmm::workspace_bytes(k, op, max_rows < 1 ? 1 : max_rows)
                                             : generic_workspace_bytes(k, op, max_rows);
    size_t b = use_mfma ? mfma::workspace_bytes(k, op, max_rows) : other;
    // rnvp_backward of more rows than the tile-split MFMA kernel takes runs on the any-shape kernels: one size serves both
    if (use_mfma && op == RNVP_OP_TRAIN && !mfma::backward_rows_ok(k, max_rows) && other > b) b = other;
    return b + 256;
}

int rnvp_forward_logprob(void *stream, const rnvp_shape *shape, const float *params, const uint8_t *masks,
                         const float *x, const float *c, const int64_t *row_index, int64_t n_rows,
                         float *z_out, float *logdet_out, float *logp_out, float *logp_sum,
                         void *workspace, size_t workspace_bytes) {
    KShape k;
    int rc = make_kshape(shape, &k);
    if (rc) return rc;
    if (n_rows < 0) return RNVP_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (n_rows == 0) {
        if (logp_sum) RNVP_HIP_TRY(hipMemsetAsync(logp_sum, 0, sizeof(float), st));
        return RNVP_OK;
    }
    if (bad_ptrs(k, params, masks, x, c)) return RNVP_EINVAL;
    if (mfma::supported(k))
        return mfma::forward(st, k, params, x, c, row_index, n_rows, z_out, logdet_out, logp_out, logp_sum,
                             workspace, workspace_bytes);
    if (!masks) return RNVP_EINVAL;
    if (lmm::use_lmm(k, RNVP_OP_FORWARD))
        return lmm::forward(st, k, params, masks, x, c, row_index, n_rows, z_out, logdet_out, logp_out, logp_sum, workspace,
                            workspace_bytes);
    return generic_forward(st, k, params, masks, x, c, row_index, n_rows, z_out, logdet_out, logp_out,
                           logp_sum, workspace, workspace_bytes);
}

int rnvp_inverse(void *stream, const rnvp_shape *shape, const float *params, const uint8_t *masks,
                 const float *z, const float *c, int64_t n_rows, float *x_out,
                 void *workspace, size_t workspace_bytes) {
    KShape k;
    int rc = make_kshape(shape, &k);
    if (rc) return rc;
    if (n_rows < 0) return RNVP_EINVAL;
    if (n_rows == 0) return RNVP_OK;
    if (bad_ptrs(k, params, masks, z, c) || !x_out) return RNVP_EINVAL;
    if (mfma::supported(k))
        return mfma::inverse(static_cast<hipStream_t>(stream), k, params, z, c, n_rows, x_out, workspace,
                             workspace_bytes);
    if (!masks) return RNVP_EINVAL;
    if (lmm::use_lmm(k, RNVP_OP_INVERSE))
        return lmm::inverse(static_cast<hipStream_t>(stream), k, params, masks, z, c, n_rows, x_out, workspace, workspace_bytes);
    return generic_inverse(static_cast<hipStream_t>(stream), k, params, masks, z, c, n_rows, x_out);
}

int rnvp_prior_normal(void *stream, uint64_t seed, int64_t row_offset, int64_t n_rows, int32_t d, float *z_out) {
    if (n_rows < 0 || d < 1 || row_offset < 0) return RNVP_EINVAL;
    if (n_rows == 0) return RNVP_OK;
    if (!z_out) return RNVP_EINVAL;
    return prior_normal(static_cast<hipStream_t>(stream), seed, row_offset, n_rows, d, z_out);
}

int rnvp_sample(void *stream, const rnvp_shape *shape, const float *params, const uint8_t *masks, const float *c,
                int64_t n_rows, uint64_t seed, int64_t row_offset, float *x_out, void *workspace,
                size_t workspace_bytes) {
    KShape k;
    int rc = make_kshape(shape, &k);
    if (rc) return rc;
    if (n_rows < 0 || row_offset < 0) return RNVP_EINVAL;
    if (n_rows == 0) return RNVP_OK;
    if (!params || (!masks && !k.alt) || (k.c > 0 && !c) || !x_out) return RNVP_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (mfma::supported(k)) return mfma::sample(st, k, params, c, n_rows, seed, row_offset, x_out, workspace, workspace_bytes);
    if (!masks) return RNVP_EINVAL;
    // generic kernels: the same draws written to x_out, then the inverse in place
    rc = prior_normal(st, seed, row_offset, n_rows, k.d, x_out);
    if (rc) return rc;
    if (lmm::use_lmm(k, RNVP_OP_INVERSE)) return lmm::inverse(st, k, params, masks, x_out, c, n_rows, x_out, workspace, workspace_bytes);
    return generic_inverse(st, k, params, masks, x_out, c, n_rows, x_out);
}

static int loss_grad_any(void *stream, const rnvp_shape *shape, const float *params, const uint8_t *masks,
                         const float *x, const float *c, const int64_t *row_index, int64_t n_rows, float inv_B,
                         Seeds sd, float *grad_out, float *loss_out, void *workspace, size_t workspace_bytes);

int rnvp_loss_grad(void *stream, const rnvp_shape *shape, const float *params, const uint8_t *masks,
                   const float *x, const float *c, const int64_t *row_index, int64_t n_rows, float inv_B,
                   float *grad_out, float *loss_out, void *workspace, size_t workspace_bytes) {
    return loss_grad_any(stream, shape, params, masks, x, c, row_index, n_rows, inv_B, Seeds{}, grad_out, loss_out,
                         workspace, workspace_bytes);
}

int rnvp_loss_grad_zseed(void *stream, const rnvp_shape *shape, const float *params, const uint8_t *masks,
                         const float *x, const float *c, const int64_t *row_index, int64_t n_rows, float inv_B,
                         const float *gz, float *grad_out, float *loss_out, void *workspace, size_t workspace_bytes) {
    if (!gz && n_rows > 0) return RNVP_EINVAL;
    return loss_grad_any(stream, shape, params, masks, x, c, row_index, n_rows, inv_B, Seeds{gz, nullptr, nullptr}, grad_out,
                         loss_out, workspace, workspace_bytes);
}

int rnvp_backward(void *stream, const rnvp_shape *shape, const float *params, const uint8_t *masks,
                  const float *x, const float *c, const int64_t *row_index, int64_t n_rows,
                  const float *gz, const float *gld, float *grad_out, float *gx_out,
                  void *workspace, size_t workspace_bytes) {
    if ((!gz || !gld) && n_rows > 0) return RNVP_EINVAL;
    // the kernels' loss output is the log-det sum scaled by inv_B: with inv_B = 0 nothing but the caller's seeds drives
    // the backward, and the (unused) loss slot stays finite
    return loss_grad_any(stream, shape, params, masks, x, c, row_index, n_rows, 0.0f, Seeds{gz, gld, gx_out}, grad_out,
                         nullptr, workspace, workspace_bytes);
}

// rnvp_backward_cond / rnvp_inverse_backward: the any-shape 16-row kernel (rnvp_lmm.hip) is the one that carries d loss / d c
// and the backward through the inverse; its transposed first-Linear fragments are packed with the condition columns (gcw)
static int cond_kshape(const rnvp_shape *shape, bool want_gc, KShape *k) {
    int rc = make_kshape(shape, k);
    if (rc) return rc;
    k->family = RNVP_FAMILY_LMM16;
    k->gcw = (want_gc && k->c > 0) ? 1 : 0;
    return lmm::use_lmm(*k, RNVP_OP_TRAIN) ? RNVP_OK : RNVP_EUNSUPPORTED;
}
// ... and behind it, for the shapes whose 16-row tile image exceeds the LDS budget of that kernel (e.g. hidden = (512,) on
// C2-sized rows), the one-thread-per-row VALU kernel (rnvp_generic.hip), which carries the same seeds: slow, but the seam has
// no shape hole (the reference differentiates any shape: realnvp.py:92,120-129, nflow.py:141-145)
static int cond_kshape_any(const rnvp_shape *shape, bool want_gc, KShape *k, bool *valu) {
    *valu = false;
    int rc = cond_kshape(shape, want_gc, k);
    if (rc != RNVP_EUNSUPPORTED) return rc;
    rc = make_kshape(shape, k);
    if (rc) return rc;
    k->gcw = (want_gc && k->c > 0) ? 1 : 0;
    *valu = true;
    return generic_workspace_bytes(*k, RNVP_OP_TRAIN, 1) > 0 ? RNVP_OK : RNVP_EUNSUPPORTED;
}

size_t rnvp_backward_cond_workspace_bytes(const rnvp_shape *shape, int64_t max_rows) {
    KShape k;
    bool valu;
    if (cond_kshape_any(shape, true, &k, &valu) != RNVP_OK) return 0;
    if (valu) return generic_workspace_bytes(k, RNVP_OP_TRAIN, max_rows < 1 ? 1 : max_rows) + 256;
    return lmm::workspace_bytes(k, RNVP_OP_TRAIN, max_rows < 1 ? 1 : max_rows) + 256;
}

static int cond_backward(void *stream, const rnvp_shape *shape, const float *params, const uint8_t *masks, const float *rows,
                         const float *c, const int64_t *row_index, int64_t n_rows, Seeds sd, float *grad_out, void *workspace,
                         size_t workspace_bytes) {
    KShape k;
    bool valu;
    int rc = cond_kshape_any(shape, sd.gc != nullptr, &k, &valu);
    if (rc) return rc;
    if (n_rows < 0 || !grad_out) return RNVP_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (n_rows == 0) {
        RNVP_HIP_TRY(hipMemsetAsync(grad_out, 0, (size_t)2 * k.npn * k.L * sizeof(float), st));
        return RNVP_OK;
    }
    if (!masks || bad_ptrs(k, params, masks, rows, c)) return RNVP_EINVAL;
    if (k.c == 0) sd.gc = nullptr;
    if (valu)
        return generic_loss_grad(st, k, params, masks, rows, c, row_index, n_rows, 0.0f, grad_out, nullptr, workspace, workspace_bytes, sd);
    return lmm::loss_grad(st, k, params, masks, rows, c, row_index, n_rows, 0.0f, grad_out, nullptr, workspace, workspace_bytes, sd);
}

int rnvp_backward_cond(void *stream, const rnvp_shape *shape, const float *params, const uint8_t *masks,
                       const float *x, const float *c, const int64_t *row_index, int64_t n_rows,
                       const float *gz, const float *gld, float *grad_out, float *gx_out, float *gc_out,
                       void *workspace, size_t workspace_bytes) {
    if ((!gz || !gld) && n_rows > 0) return RNVP_EINVAL;
    Seeds sd;
    sd.gz = gz; sd.gld = gld; sd.gx = gx_out; sd.gc = gc_out;
    return cond_backward(stream, shape, params, masks, x, c, row_index, n_rows, sd, grad_out, workspace, workspace_bytes);
}

int rnvp_inverse_backward(void *stream, const rnvp_shape *shape, const float *params, const uint8_t *masks,
                          const float *z, const float *c, int64_t n_rows, const float *gx, float *grad_out,
                          float *gz_out, float *gc_out, void *workspace, size_t workspace_bytes) {
    if (!gx && n_rows > 0) return RNVP_EINVAL;
    Seeds sd;
    sd.gz = gx; sd.gx = gz_out; sd.gc = gc_out; sd.inv = 1;
    return cond_backward(stream, shape, params, masks, z, c, nullptr, n_rows, sd, grad_out, workspace, workspace_bytes);
}

static int loss_grad_any(void *stream, const rnvp_shape *shape, const float *params, const uint8_t *masks,
                         const float *x, const float *c, const int64_t *row_index, int64_t n_rows, float inv_B,
                         Seeds sd, float *grad_out, float *loss_out, void *workspace, size_t workspace_bytes) {
    KShape k;
    int rc = make_kshape(shape, &k);
    if (rc) return rc;
    if (n_rows < 0 || !grad_out) return RNVP_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t P = (size_t)2 * k.npn * k.L;
    if (n_rows == 0) {      // a rank whose shard of a ragged batch is empty contributes zeros
        RNVP_HIP_TRY(hipMemsetAsync(grad_out, 0, P * sizeof(float), st));
        if (loss_out) RNVP_HIP_TRY(hipMemsetAsync(loss_out, 0, sizeof(float), st));
        return RNVP_OK;
    }
    if (bad_ptrs(k, params, masks, x, c)) return RNVP_EINVAL;
    const bool seeded = sd.gld || sd.gx;        // rnvp_backward: only the tile-split MFMA kernel takes those (small calls)
    if (mfma::train_supported(k) && (!seeded || mfma::backward_rows_ok(k, n_rows)))
        return mfma::loss_grad(st, k, params, x, c, row_index, n_rows, inv_B, grad_out, loss_out, workspace,
                               workspace_bytes, sd);
    if (!masks) return RNVP_EINVAL;
    if (lmm::use_lmm(k, RNVP_OP_TRAIN))
        return lmm::loss_grad(st, k, params, masks, x, c, row_index, n_rows, inv_B, grad_out, loss_out, workspace,
                              workspace_bytes, sd);
    return generic_loss_grad(st, k, params, masks, x, c, row_index, n_rows, inv_B, grad_out, loss_out,
                             workspace, workspace_bytes, sd);
}

int rnvp_adam_step(void *stream, float *params, const float *grad, float *exp_avg, float *exp_avg_sq,
                   int64_t n_params, double lr, double beta1, double beta2, double eps,
                   double weight_decay, int64_t step) {
    return adam_step(static_cast<hipStream_t>(stream), params, grad, exp_avg, exp_avg_sq, n_params, lr,
                     beta1, beta2, eps, weight_decay, step);
}

int rnvp_dp_finish_step(void *stream, float *params, const float *grad_loss, float *exp_avg, float *exp_avg_sq,
                        int64_t n_params, double lr, double beta1, double beta2, double eps, double weight_decay,
                        int64_t step, float *loss_out) {
    if (!grad_loss || !loss_out || n_params < 0) return RNVP_EINVAL;
    return adam_step(static_cast<hipStream_t>(stream), params, grad_loss, exp_avg, exp_avg_sq, n_params, lr, beta1,
                     beta2, eps, weight_decay, step, grad_loss + n_params, loss_out);
}

int rnvp_train_step(void *stream, const rnvp_shape *shape, float *params, const uint8_t *masks,
                    const float *x, const float *c, const int64_t *row_index, int64_t n_rows, float inv_B,
                    float *grad_buf, float *loss_out, float *exp_avg, float *exp_avg_sq,
                    double lr, double beta1, double beta2, double eps, double weight_decay, int64_t step,
                    void *workspace, size_t workspace_bytes) {
    KShape k;
    int rc = make_kshape(shape, &k);
    if (rc) return rc;
    if (n_rows > 0 && mfma::train_supported(k)) {
        if (bad_ptrs(k, params, masks, x, c) || !grad_buf || !exp_avg || !exp_avg_sq || step < 1) return RNVP_EINVAL;
        return mfma::train_step(static_cast<hipStream_t>(stream), k, params, x, c, row_index, n_rows, inv_B, grad_buf,
                                loss_out, exp_avg, exp_avg_sq, make_adam(lr, beta1, beta2, eps, weight_decay, step),
                                workspace, workspace_bytes);
    }
    rc = rnvp_loss_grad(stream, shape, params, masks, x, c, row_index, n_rows, inv_B, grad_buf, loss_out,
                        workspace, workspace_bytes);
    if (rc) return rc;
    return rnvp_adam_step(stream, params, grad_buf, exp_avg, exp_avg_sq, (int64_t)rnvp_param_count(shape), lr,
                          beta1, beta2, eps, weight_decay, step);
}

int rnvp_fit_epoch_resident(const rnvp_shape *shape, int64_t batch_size) {
    KShape ks;
    if (make_kshape(shape, &ks) != RNVP_OK) return 0;
    return resident::fits(ks, batch_size) ? 1 : 0;
}

int rnvp_fit_epochs(void *stream, const rnvp_shape *shape, float *params, const uint8_t *masks,
                    const float *x, const float *c, const int64_t *perms, int64_t n, int64_t batch_size, int64_t n_epochs,
                    float *grad_buf, float *loss_hist, float *exp_avg, float *exp_avg_sq,
                    double lr, double beta1, double beta2, double eps, double weight_decay,
                    int64_t first_step, void *workspace, size_t workspace_bytes) {
    if (n < 0 || batch_size < 1 || n_epochs < 0 || !perms || !loss_hist || first_step < 1) return RNVP_EINVAL;
    KShape ks;
    const int rc0 = make_kshape(shape, &ks);
    if (rc0) return rc0;
    const int64_t nb = (n + batch_size - 1) / batch_size;
    // a model that fits one CU's LDS at a batch of at most 128 rows: all epochs in one persistent launch
    if (n > 0 && n_epochs > 0 && resident::fits(ks, batch_size)) {
        if (bad_ptrs(ks, params, masks, x, c) || !exp_avg || !exp_avg_sq) return RNVP_EINVAL;
        return resident::fit_epoch(static_cast<hipStream_t>(stream), ks, params, masks, x, c, perms, n, batch_size, n_epochs, loss_hist,
                                   exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, first_step);
    }
    // the register-chained kernels keep their packed weight fragments in the workspace: packed once at the head of this call,
    // then re-packed by each step's finish kernel (two launches per batch: training kernel + finish)
    const bool chained = n > 0 && mfma::train_supported(ks);
    if (chained && (bad_ptrs(ks, params, masks, x, c) || !grad_buf || !exp_avg || !exp_avg_sq)) return RNVP_EINVAL;
    bool packed_valid = false;
    for (int64_t e = 0; e < n_epochs; ++e) {
        int64_t k = 0;
        for (int64_t s0 = 0; s0 < n; s0 += batch_size, ++k) {
            const int64_t rows = (n - s0 < batch_size) ? n - s0 : batch_size;
            const bool last = e + 1 == n_epochs && s0 + rows >= n;
            int rc;
            if (chained) {
                rc = mfma::train_step(static_cast<hipStream_t>(stream), ks, params, x, c, perms + e * n + s0, rows, 1.0f / (float)rows,
                                      grad_buf, loss_hist + e * nb + k, exp_avg, exp_avg_sq,
                                      make_adam(lr, beta1, beta2, eps, weight_decay, first_step + e * nb + k), workspace,
                                      workspace_bytes, packed_valid, !last);
                packed_valid = true;
            } else {
                rc = rnvp_train_step(stream, shape, params, masks, x, c, perms + e * n + s0, rows, 1.0f / (float)rows,
                                     grad_buf, loss_hist + e * nb + k, exp_avg, exp_avg_sq, lr, beta1, beta2, eps,
                                     weight_decay, first_step + e * nb + k, workspace, workspace_bytes);
            }
            if (rc) return rc;
        }
    }
    return RNVP_OK;
}

int rnvp_fit_epoch(void *stream, const rnvp_shape *shape, float *params, const uint8_t *masks,
                   const float *x, const float *c, const int64_t *perm, int64_t n, int64_t batch_size,
                   float *grad_buf, float *loss_hist, float *exp_avg, float *exp_avg_sq,
                   double lr, double beta1, double beta2, double eps, double weight_decay,
                   int64_t first_step, void *workspace, size_t workspace_bytes) {
    return rnvp_fit_epochs(stream, shape, params, masks, x, c, perm, n, batch_size, 1, grad_buf, loss_hist, exp_avg, exp_avg_sq, lr,
                           beta1, beta2, eps, weight_decay, first_step, workspace, workspace_bytes);
}

}  // extern "C"
