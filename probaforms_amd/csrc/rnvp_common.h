// rnvp_common.h -- shared host/device declarations of librnvp_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stddef.h>
#include <stdint.h>
#include <atomic>
#include <cstring>

#include "../../include/rnvp_hip.h"
#include "../../include/cvae_hip.h"

namespace rnvp {

constexpr int kMaxLin = RNVP_MAX_HIDDEN + 1;
constexpr float kLog2Pi = 1.8378770664093453f;   // ln(2*pi)

// Kernel-side view of rnvp_shape with the per-Linear geometry of ONE s/t net
// (gen_network, /root/reference/probaforms/models/realnvp.py:19-43) precomputed.
struct KShape {
    int L, d, c, nh, act;
    int alt;                           // rnvp_shape::alt_masks (0 arbitrary, 1/2 alternating)
    int prec;                          // rnvp_shape::precision (RNVP_PREC_*) for the FORWARD PHASE OF THE TRAINING kernels, RNVP_PREC_AUTO resolved
    int prec_flow;                     // ... for the forward / inverse / sampling kernels, RNVP_PREC_AUTO resolved
    int prec_auto;                     // the caller left the choice to the library (RNVP_PREC_AUTO)
    int small_latency;                 // rnvp_shape::small_calls == RNVP_SMALL_LATENCY
    int family;                        // rnvp_shape::family (RNVP_FAMILY_*)
    int gcw;                           // internal (rnvp_backward_cond / rnvp_inverse_backward with gc_out): the transposed first-Linear
                                       // fragments of the any-shape kernels cover the condition columns too (d loss / d c)
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
    if (s->precision < RNVP_PREC_AUTO || s->precision > RNVP_PREC_BX3) return RNVP_EINVAL;
    if (s->small_calls != RNVP_SMALL_INVARIANT && s->small_calls != RNVP_SMALL_LATENCY) return RNVP_EINVAL;
    k->small_latency = s->small_calls == RNVP_SMALL_LATENCY;
    if (s->family < RNVP_FAMILY_AUTO || s->family > RNVP_FAMILY_LMM64) return RNVP_EINVAL;
    k->family = s->family;
    k->gcw = 0;
    k->prec_auto = s->precision == RNVP_PREC_AUTO;
    // auto: bx3 where the tile geometry has 4+ feature slots per lane (d > 16 or cdim > 4: rnvp_mfma.h pick_tiles) -- measured
    // 1.3-1.4x on every operation there.  In the d <= 16 geometry the split inputs cost 36 registers the f32 form does not need:
    //   * the training kernel's forward phase takes them from five hidden tiles on (hidden > 64) -- its spills are cold (none inside
    //     a tile loop: profiles/r04_scratch_isa.txt);
    //   * the flow kernels take them from the same width on SINCE ROUND 6.  Rounds 3-5 measured the barrier-free split-bf16 form
    //     (three row tiles per wave, 42-57 spilled registers: 281-295 vs 88 MB of HBM traffic per 1M rows) at 1-3 % ahead of the f32
    //     kernels and kept f32 -- with a dozen launches from an idle chip, i.e. under its clock ramp.  With warm clocks (>= 50 ms of
    //     warm-up, 30 launches, three repetitions: profiles/r06_c2_flow_bx3_warm.txt) the split form is 8.8 % (forward) / 8.1 %
    //     (inverse, sampling) faster: 0.968 / 0.960 against 1.061 / 1.045 ms per 1M rows; two row tiles (no spill) -3.7 / -2.6 %,
    //     four +20 %.  The counters agree (profiles/r06_flow_pmc_c2.json / _c2_f32.json, both warm): 2.27 against 2.50 M cycles per
    //     launch at 2.32 against 2.37 GHz -- fewer cycles, not a higher clock.  The spill traffic is 0.2 GB per millisecond of an
    //     8 TB/s part.  Log-prob MAE against float64: 2.7e-6 (f32 kernels 2.1e-6, the float32 oracle itself 2.5e-6).
    //   Width sweep at d = 16, c = 4, warm clocks (profiles/r06_d16_width_prec_sweep.txt), bx3 against f32, forward / inverse / loss+grad
    //   call: hidden 16 +33 / +35 / +5 %, 32 +14 / +14 / +1 %, 64 0 / 0 / -1.8 %, 96 -5.2 / -4.8 / -2.8 %, 128 -7.1 / -6.8 / -3.8 %,
    //   256 -7.7 / -7.3 / -5.0 %: the switch sits above 64 units (rounds 3-5: above 96, from cold timings).
    const bool wide = s->d > 16 || s->c > 4;
    k->prec = s->precision == RNVP_PREC_AUTO ? ((wide || (s->n_hidden == 1 && s->hidden[0] > 64)) ? RNVP_PREC_BX3 : RNVP_PREC_F32) : s->precision;
    k->prec_flow = k->prec;
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

// Optional seeds / outputs of the backward pass (all nullable):
//   gz  [n_rows, d]  d loss / d z instead of the fused prior's z / B      (rnvp_loss_grad_zseed, rnvp_backward)
//   gld [n_rows]     d loss / d logdet per row instead of the uniform -1/B (rnvp_backward)
//   gx  [n_rows, d]  OUT: d loss / d x of the batch rows                   (rnvp_backward)
//   gc  [n_rows, c]  OUT: d loss / d c of the batch rows                   (rnvp_backward_cond, rnvp_inverse_backward; any-shape kernels)
//   inv              backward THROUGH THE INVERSE x = g(z, c) (rnvp_inverse_backward; any-shape kernels): the rows handed in are
//                    z, gz holds d loss / d x, gx receives d loss / d z; no loss term
struct Seeds {
    const float *gz = nullptr;
    const float *gld = nullptr;
    float *gx = nullptr;
    float *gc = nullptr;
    int inv = 0;
};

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
                      float inv_B, float *grad_out, float *loss_out, void *ws, size_t ws_bytes,
                      Seeds sd = Seeds{});

int generic_reduce_partials(hipStream_t st, const float *gpart, const float *losspart, int G, size_t P,
                            float loss_scale, float *grad_out, float *loss_out);

// Kernel-side view of cvae_shape: the two MLPs (/root/reference/probaforms/models/cvae.py:35-84) as KShapes with L = 1;
// KShape::d counts the leading inputs that need an input gradient (encoder 0, decoder the latent), KShape::c the rest.
struct CvaeK {
    KShape enc, dec;      // enc: (d+c) -> hidden.. -> 2*lat ; dec: (lat+c) -> hidden.. -> d
    int d, c, lat;
    int pe;               // floats of the encoder block (decoder parameters start here)
};

// ---- CVAE step on MFMA: cvae_mfma.hip (d <= 16, c <= 4, latent <= 4, one tanh hidden layer) ----
namespace cvae_mfma {
bool supported(const ::cvae_shape *s);
size_t workspace_bytes(const ::cvae_shape *s, int64_t max_rows);
int loss_grad(hipStream_t st, const ::cvae_shape *s, const float *params, const float *x, const float *c,
              const int64_t *row_index, const float *eps, int64_t n, float inv_B, float klw, float *grad_out,
              float *loss_out, void *ws, size_t ws_bytes);
int forward(hipStream_t st, const ::cvae_shape *s, const float *params, bool encode, const float *in, const float *c,
            int64_t n, float *out0, float *out1, void *ws, size_t ws_bytes);
}  // namespace cvae_mfma

// ---- Adam arithmetic shared by k_adam (rnvp_adam.hip) and the fused reduce+Adam kernel ------------
// torch.optim.Adam (realnvp.py:205-207,251) as separately rounded tensor ops; written so that
// no build flag can contract them into FMAs: bit-identical to the oracle.
struct AdamK {
    float step_size;    // lr / (1 - beta1^t)
    float bc2_sqrt;     // sqrt(1 - beta2^t)
    float w1;           // 1 - beta1
    float beta2, w2;    // beta2, 1 - beta2
    float wd, eps;
    int use_wd;
};

#ifdef __HIPCC__
// Separately rounded multiply / add / subtract.  HIP's __fmul_rn & co. are plain operators and DO get
// contracted into FMAs under -ffp-contract=fast, so the three ops are pinned with one-instruction asm.
__device__ __forceinline__ float mul_rn(float a, float b) { float r; asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float add_rn(float a, float b) { float r; asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float sub_rn(float a, float b) { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

__device__ __forceinline__ void adam_one(float &p, float g, float &m, float &v, const AdamK &a) {
    if (a.use_wd) g = add_rn(g, mul_rn(a.wd, p));                              // grad = grad + wd * param
    m = add_rn(m, mul_rn(a.w1, sub_rn(g, m)));                                 // exp_avg.lerp_(grad, 1 - beta1)
    v = add_rn(mul_rn(v, a.beta2), mul_rn(a.w2, mul_rn(g, g)));                // mul_(b2).addcmul_(g, g, 1-b2)
    // sqrtf and '/' are IEEE correctly rounded under hipcc's defaults and cannot be contracted
    const float denom = add_rn(sqrtf(v) / a.bc2_sqrt, a.eps);
    p = sub_rn(p, mul_rn(a.step_size, m / denom));                             // addcdiv_(m, denom, -step_size)
}
#endif

AdamK make_adam(double lr, double beta1, double beta2, double eps, double wd, int64_t step);

namespace cvae_mfma {
int train_step(hipStream_t st, const ::cvae_shape *s, float *params, const float *x, const float *c,
               const int64_t *row_index, const float *eps, int64_t n, float inv_B, float klw, float *grad_buf,
               float *loss_out, float *exp_avg, float *exp_avg_sq, const AdamK &adam, void *ws, size_t ws_bytes,
               bool packed_valid = false, bool pack_next = false);
}  // namespace cvae_mfma

// ---- optimizer: rnvp_adam.hip -----------------------------------------------------------
int adam_step(hipStream_t st, float *p, const float *g, float *m, float *v, int64_t n,
              double lr, double beta1, double beta2, double eps, double wd, int64_t step,
              const float *loss_in = nullptr, float *loss_out = nullptr);

// ---- prior draws: rnvp_prior.hip ------------------------------------------------------------
int prior_normal(hipStream_t st, uint64_t seed, int64_t row0, int64_t n, int d, float *z);

// ---- the reference's CPU mt19937 stream on the device: rnvp_prior_torch.hip --------------------------
int mt19937_raw_words(hipStream_t st, uint32_t *mt_state, int64_t count, uint32_t *out, void *workspace);

// ---- optional event bracket around the hot kernels (rnvp_profile_*, rnvp_api.hip) ---------------
// kind: RNVP_PROFILE_TRAIN (fused forward+backward), _FORWARD (log-prob), _INVERSE (sampling)
struct KernelTimer {
    hipStream_t st;
    int kind, slot;
    KernelTimer(hipStream_t s, int kind);  // records the start event if profiling is enabled
    ~KernelTimer();                        // records the stop event
};
// The same event pairs, for launches made with hipExtLaunchKernelGGL(kernel, ..., start, stop, 0, args): the kernel's own
// dispatch packet stamps the two events (no barrier packets around the launch: the bracketing form costs ~6 us of idle
// queue per side on MI355X, 1.8 % of the C2 step).  Both are nullptr while profiling is off = a plain launch.
struct KernelEvents {
    hipEvent_t start = nullptr, stop = nullptr;
    explicit KernelEvents(int kind);
};

// rnvp_last_dispatch (rnvp_api.hip): every launch site of a hot kernel notes what it launched, per thread and kind
void note_dispatch(int kind, const char *kernel, int variant, int row_tiles, int waves, int grid, int gemm1_fwd, int64_t rows);
void note_launches(int kind, int launches);       // launches of the whole call, once known

#define RNVP_HIP_TRY(expr)                         \
    do {                                           \
        hipError_t e__ = (expr);                   \
        if (e__ != hipSuccess) return (int)e__;    \
    } while (0)

// Raise a kernel's dynamic-LDS limit once per DEVICE (the attribute is per device and per kernel): `done`
// keeps one bit per device id of the calling thread's current device.
inline int allow_big_lds(const void *kernel, int bytes, std::atomic<uint64_t> &done) {
    int dev = 0;
    RNVP_HIP_TRY(hipGetDevice(&dev));
    const uint64_t bit = 1ull << (dev & 63);
    if (dev > 63 || !(done.load(std::memory_order_relaxed) & bit)) {
        RNVP_HIP_TRY(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        done.fetch_or(bit, std::memory_order_relaxed);
    }
    return RNVP_OK;
}

}  // namespace rnvp
