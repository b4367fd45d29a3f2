// rnvp_common.h -- shared host/device declarations of librnvp_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <cstring>

#include "../../include/rnvp_hip.h"

namespace rnvp {

constexpr int kMaxLin = RNVP_MAX_HIDDEN + 1;
constexpr float kLog2Pi = 1.8378770664093453f;   // ln(2*pi)

// Kernel-side view of rnvp_shape with the per-Linear geometry of ONE s/t net
// (gen_network, /root/reference/probaforms/models/realnvp.py:19-43) precomputed.
struct KShape {
    int L, d, c, nh, act;
    int alt;                           // rnvp_shape::alt_masks (0 arbitrary, 1/2 alternating)
    int nin[kMaxLin], nout[kMaxLin];   // Linear k: [nout, nin]
    int woff[kMaxLin], boff[kMaxLin];  // float offsets inside one net's parameter block
    int npn;                           // parameters per net
    int hs;                            // sum(hidden)
    int hmax;                          // max(hidden)
    int wmax;                          // max(d + c, hidden..., d)
};

inline int make_kshape(const rnvp_shape *s, KShape *k) {
    if (!s || s->L < 1 || s->d < 1 || s->c < 0 || s->n_hidden < 1 || s->n_hidden > RNVP_MAX_HIDDEN)
        return RNVP_EINVAL;
    k->L = s->L; k->d = s->d; k->c = s->c; k->nh = s->n_hidden;
    k->act = (s->act == RNVP_ACT_TANH) ? RNVP_ACT_TANH : RNVP_ACT_RELU;
    if (s->alt_masks < 0 || s->alt_masks > 2) return RNVP_EINVAL;
    k->alt = s->alt_masks;
    int in = s->d + s->c, off = 0;
    k->hs = 0; k->hmax = 0; k->wmax = in > s->d ? in : s->d;
    for (int i = 0; i <= s->n_hidden; ++i) {
        int out = (i < s->n_hidden) ? s->hidden[i] : s->d;
        if (out < 1) return RNVP_EINVAL;
        k->nin[i] = in; k->nout[i] = out;
        k->woff[i] = off; k->boff[i] = off + out * in;
        off += out * in + out;
        if (i < s->n_hidden) {
            k->hs += out;
            if (out > k->hmax) k->hmax = out;
            if (out > k->wmax) k->wmax = out;
        }
        in = out;
    }
    for (int i = s->n_hidden + 1; i < kMaxLin; ++i) { k->nin[i] = k->nout[i] = k->woff[i] = k->boff[i] = 0; }
    k->npn = off;
    return RNVP_OK;
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- generic (any-shape) path: rnvp_generic.hip ---------------------------------------
size_t generic_workspace_bytes(const KShape &k, int op, int64_t max_rows);
int generic_forward(hipStream_t st, const KShape &k, const float *params, const uint8_t *masks,
                    const float *x, const float *c, const int64_t *row_index, int64_t n,
                    float *z_out, float *logdet_out, float *logp_out, float *logp_sum,
                    void *ws, size_t ws_bytes);
int generic_inverse(hipStream_t st, const KShape &k, const float *params, const uint8_t *masks,
                    const float *z, const float *c, int64_t n, float *x_out);
int generic_loss_grad(hipStream_t st, const KShape &k, const float *params, const uint8_t *masks,
                      const float *x, const float *c, const int64_t *row_index, int64_t n,
                      float inv_B, float *grad_out, float *loss_out, void *ws, size_t ws_bytes);

int generic_reduce_partials(hipStream_t st, const float *gpart, const float *losspart, int G, size_t P,
                            float loss_scale, float *grad_out, float *loss_out);

// ---- optimizer: rnvp_adam.hip -----------------------------------------------------------
int adam_step(hipStream_t st, float *p, const float *g, float *m, float *v, int64_t n,
              double lr, double beta1, double beta2, double eps, double wd, int64_t step);

// ---- optional event bracket around the dominant kernel (rnvp_profile_*, rnvp_api.hip) ------------
struct KernelTimer {
    hipStream_t st;
    bool on;
    explicit KernelTimer(hipStream_t s);   // records the start event if profiling is enabled
    ~KernelTimer();                        // records the stop event
};

#define RNVP_HIP_TRY(expr)                         \
    do {                                           \
        hipError_t e__ = (expr);                   \
        if (e__ != hipSuccess) return (int)e__;    \
    } while (0)

}  // namespace rnvp
