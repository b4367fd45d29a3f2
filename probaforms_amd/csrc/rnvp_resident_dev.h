// rnvp_resident_dev.h -- device helpers, plans and launch arguments shared by the resident-fit translation units
// (rnvp_resident.hip: one hidden layer; rnvp_resident_ns.hip: its net-split form for batches of at most 64 rows;
// rnvp_resident_deep.hip: two or three hidden layers; cvae_resident.hip: the conditional VAE).
// Split four ways so that the ~130 kernel instantiations compile in parallel.
#pragma once
#include <cmath>

#include "rnvp_common.h"
#include "rnvp_resident.h"

namespace rnvp {
namespace resident {

// plan of a launch (LDS offsets in floats) and the arguments of an epoch: shared across the translation units
struct RcPlan {
    int W, P, mv_lds;
    int save;                                          // the forward's hidden activations are kept for the backward (nh == 1)
    int stg_net;                                       // floats of one net's stage block: npn + kDump
    int oPAR, oM, oV, oSTG, oRED, oXS, oTT;            // float offsets
    int xs_floats, tt_floats, stg_floats;              // per wave
    int total_floats;
};

struct EpochArgs {
    float *params; const uint8_t *masks; const float *x, *c; const int64_t *perm; int64_t n, batch_size, n_epochs;
    float *loss_hist, *exp_avg, *exp_avg_sq; double lr, beta1, beta2, eps, wd; int64_t first_step;
};

namespace {

using f4 = __attribute__((ext_vector_type(4))) float;
constexpr int kMaxWaves = 16;
constexpr size_t kLdsMax = 160 * 1024;

__device__ __forceinline__ f4 mfma16(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// LDS traffic between lanes of ONE wave: only the compiler has to be kept from reordering the accesses
__device__ __forceinline__ void wfence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- the per-lane layout shared by the three kernels (d <= 16, d + cdim <= 31, hidden tiles of 16 units) ----
// Every vector of a row -- the net input [x * mask | c], a tile of 16 hidden units, s, t and their gradients -- is an f4
// per lane and 16 elements: lane (q = lane >> 4, r = lane & 15) keeps elements 4e + q (e = 0..3) of row r.  An MFMA's D
// operand comes out in exactly that form when the A rows are gathered in the order pi(i) = 4 (i & 3) + (i >> 2), and it IS
// the B operand of the next GEMM's k-step e (K index = lane group) -- so a whole layer, forward and input-gradient chain,
// runs register to register.  LDS holds the weights (A operands gathered straight from the flat parameters through
// per-lane offsets computed once: independent of the data chain, issued early), the saved layer inputs (lane-private) and
// a few transposition tiles per wave for the contractions over the 16 rows (weight gradients; a ones element in the input
// tile yields d b1).  Padding is handled by zeros on ONE side of every product (inputs past d + cdim are zero, hidden
// units past h are multiplied by a 0 / 1 lane mask, output features past d are passed through), so no gather is guarded.
constexpr int TS = 17;
constexpr int kRcMaxWaves = 8;
constexpr int kDump = 16;              // per-net dump zone of the stage: where the padding lanes of a weight-gradient tile write (lane & 15)


// e^x to ~1.5 ulp from the hardware exp2: x log2(e) split into a rounded product and its error
__device__ __forceinline__ float exp_acc(float xv) {
    const float t = xv * 1.4426950408889634f;
    float rr = fmaf(xv, 1.4426950408889634f, -t);
    rr = fmaf(xv, 1.9259629911266175e-8f, rr);
    const float e = __builtin_amdgcn_exp2f(t);
    return fmaf(e, rr * 0.6931471805599453f, e);
}
template <int ACT> __device__ __forceinline__ float actf(float v) {
    if (ACT != RNVP_ACT_TANH) return fmaxf(v, 0.f);
    const float e = __builtin_amdgcn_exp2f(v * 2.8853900817779268f);
    return fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + e), 1.0f);
}
template <int ACT> __device__ __forceinline__ float dactf(float hv) {
    if (ACT != RNVP_ACT_TANH) return hv > 0.f ? 1.f : 0.f;
    return fmaf(-hv, hv, 1.f);
}
// sum over the 16 lanes of a DPP row (the tile's 16 rows of one lane group), result in every lane: four DPP moves
// (quad swaps, half-row mirror, row mirror) instead of four trips through the LDS crossbar (ds_bpermute ~ 100 cycles each)
template <int CTRL> __device__ __forceinline__ float dpp_move(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_move<0xB1>(v);          // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);          // quad_perm [2,3,0,1]
    v += dpp_move<0x141>(v);         // row_half_mirror
    v += dpp_move<0x140>(v);         // row_mirror
    return v;
}
// the lane's elements 4e + q of row r -> tile [element][TS]; operand of a contraction over rows: element i, rows 4ks + q
__device__ __forceinline__ void tile_put(float *T, f4 v, int q, int r) {
#pragma unroll
    for (int e = 0; e < 4; ++e) T[(4 * e + q) * TS + r] = v[e];
}
__device__ __forceinline__ void tile_get(const float *T, int q, int i, float (&o)[4]) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) o[ks] = T[i * TS + 4 * ks + q];
}

// this step's Adam scalars, the bookkeeping in double like torch (make_adam, rnvp_adam.hip); advances beta^t to the next step
__device__ __forceinline__ AdamK step_adam(double lr, double beta1, double beta2, double eps, double wd, double &b1t, double &b2t) {
    AdamK a;
    a.step_size = (float)(lr / (1.0 - b1t)); a.bc2_sqrt = (float)sqrt(1.0 - b2t);
    b1t *= beta1; b2t *= beta2;
    a.w1 = (float)(1.0 - beta1); a.beta2 = (float)beta2; a.w2 = (float)(1.0 - beta2);
    a.wd = (float)wd; a.eps = (float)eps; a.use_wd = wd != 0.0;
    return a;
}

// After the barrier that ends a step's backward: the nw waves' stages are added in wave order (deterministic) and Adam is
// applied in place to the LDS-resident parameters (torch.optim.Adam as separately rounded operations, rnvp_common.h).
// Parameter p of net block p / npn sits at stage index (p / npn) * stg_net + p % npn; two parameters per pass so that their
// loads, divisions and square roots overlap.
__device__ __forceinline__ void adam_phase(const float *st0, int stg_floats, int stg_net, int npn, int P, int nw, float *PAR, float *MM,
                                           float *VV, bool mv_lds, float *__restrict__ exp_avg, float *__restrict__ exp_avg_sq,
                                           const AdamK &a, int tid, int nthreads) {
    const float rnpn = 1.0f / (float)npn;
    // moments in HBM (a model whose moments do not fit LDS): the next pass's are requested before this pass computes
    float nm[2] = {0.f, 0.f}, nv[2] = {0.f, 0.f};
    auto request = [&](int p0) {
        if (mv_lds || p0 >= P) return;
        const int p1 = p0 + nthreads < P ? p0 + nthreads : p0;
        nm[0] = exp_avg[p0]; nv[0] = exp_avg_sq[p0];
        nm[1] = exp_avg[p1]; nv[1] = exp_avg_sq[p1];
    };
    request(tid);
    for (int p0 = tid; p0 < P; p0 += 2 * nthreads) {
        const int p1 = p0 + nthreads;
        const bool two = p1 < P;
        const int pp[2] = {p0, two ? p1 : p0};
        float g[2], pv[2], mv[2], vv[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            int ln = (int)(((float)pp[u] + 0.5f) * rnpn);              // pp / npn (exact after the correction)
            if (ln * npn > pp[u]) --ln;
            else if ((ln + 1) * npn <= pp[u]) ++ln;
            const int sp = ln * stg_net + (pp[u] - ln * npn);
            g[u] = st0[sp];
            for (int w = 1; w < nw; ++w) g[u] += st0[(size_t)w * stg_floats + sp];
            pv[u] = PAR[pp[u]];
            if (mv_lds) { mv[u] = MM[pp[u]]; vv[u] = VV[pp[u]]; }
            else { mv[u] = nm[u]; vv[u] = nv[u]; }
        }
        request(p0 + 2 * nthreads);
#pragma unroll
        for (int u = 0; u < 2; ++u) adam_one(pv[u], g[u], mv[u], vv[u], a);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u == 1 && !two) break;
            PAR[pp[u]] = pv[u];
            if (mv_lds) { MM[pp[u]] = mv[u]; VV[pp[u]] = vv[u]; }
            else { exp_avg[pp[u]] = mv[u]; exp_avg_sq[pp[u]] = vv[u]; }
        }
    }
}

}  // namespace

// k-steps of the net input incl. the ones element behind it
inline int rc_kit(const KShape &k) { const int ki = (k.d + k.c + 1 + 3) / 4; return ki <= 2 ? 2 : (ki <= 4 ? 4 : 8); }
// rnvp_resident_deep.hip: nh = 2 or 3
int launch_deep(hipStream_t st, const KShape &k, const RcPlan &p, const EpochArgs &a);
// rnvp_resident_ns.hip: nh = 1, batches of at most 64 rows, a wave per (row tile, net)
bool ns_applies(const KShape &k, int64_t batch);
int launch_ns(hipStream_t st, const KShape &k, const EpochArgs &a);

}  // namespace resident
}  // namespace rnvp
