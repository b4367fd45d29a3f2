// rnvp_generic_net.h -- one MLP (Linear, act, ..., Linear) evaluated "one thread = one row" with the
// row state in LDS [feature][TBP]; shared by the generic RealNVP kernels (rnvp_generic.hip) and the
// CVAE kernels (cvae_generic.hip).  A KShape describes the MLP: nh activated Linears + one plain
// Linear; `d` is the number of LEADING input columns whose gradient the backward must produce.
#pragma once
#include "rnvp_common.h"

namespace rnvp {

// tanh(v) = 1 - 2 / (1 + e^{2v}) on v_exp_f32 / v_rcp_f32 (~1 ulp each; absolute error ~1e-7, the
// rounding level of values near 1; exact saturation through e = +inf / 0).  ocml's tanhf costs ~40
// instructions and dominated the small-network step time (README config: 303 us -> see DESIGN.md).
__device__ __forceinline__ float act_fwd(float v, int act) {
    if (act != RNVP_ACT_TANH) return fmaxf(v, 0.f);
    const float e = __builtin_amdgcn_exp2f(v * 2.8853900817779268f);
    return fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + e), 1.0f);
}

// One s/t net (Linear, act, ..., Linear) for row t of the block; buffers are LDS [feature][TBP].
// KEEP: every hidden activation vector is kept, consecutively, in buf0 (for the backward).
// Otherwise buf0/buf1 ping-pong.
template <bool KEEP>
static __device__ void net_forward(const float *__restrict__ p, const KShape &s, const float *in,
                            float *buf0, float *buf1, float *out, int TBP, int t) {
    const float *cur = in;
    float *dst = buf0;
    for (int k = 0; k <= s.nh; ++k) {
        const int nin = s.nin[k], nout = s.nout[k];
        const float *__restrict__ W = p + s.woff[k];
        const float *__restrict__ b = p + s.boff[k];
        float *ob = (k < s.nh) ? dst : out;
        for (int o = 0; o < nout; o += 4) {
            const int o1 = min(o + 1, nout - 1), o2 = min(o + 2, nout - 1), o3 = min(o + 3, nout - 1);
            const float *w0 = W + o * nin, *w1 = W + o1 * nin, *w2 = W + o2 * nin, *w3 = W + o3 * nin;
            float a0 = b[o], a1 = b[o1], a2 = b[o2], a3 = b[o3];
            for (int i = 0; i < nin; ++i) {
                const float v = cur[i * TBP + t];
                a0 = fmaf(v, w0[i], a0);
                a1 = fmaf(v, w1[i], a1);
                a2 = fmaf(v, w2[i], a2);
                a3 = fmaf(v, w3[i], a3);
            }
            if (k < s.nh) {
                a0 = act_fwd(a0, s.act); a1 = act_fwd(a1, s.act);
                a2 = act_fwd(a2, s.act); a3 = act_fwd(a3, s.act);
            }
            ob[o * TBP + t] = a0;
            if (o + 1 < nout) ob[(o + 1) * TBP + t] = a1;
            if (o + 2 < nout) ob[(o + 2) * TBP + t] = a2;
            if (o + 3 < nout) ob[(o + 3) * TBP + t] = a3;
        }
        cur = ob;
        if (k < s.nh) {
            if (KEEP) dst += nout * TBP;
            else dst = (dst == buf0) ? buf1 : buf0;
        }
    }
}

// ---- backward of one net for the block's tile ----------------------------------------------
// gA holds d(loss)/d(net output) [d][TBP] on entry.  Weight/bias gradients are summed over the
// tile's rows by a "thread = parameter" sweep and stored (first tile) or added into the
// block-private partial gp (same layout as one net's parameters).  d(loss)/d(net input) for
// the x part is ADDED into gin [d][TBP]; with gcond (nullable: [nin[0] - d][TBP]) the part for the
// remaining input columns -- the conditions, realnvp.py:92 -- is ADDED into gcond as well.
// Uniform control flow: contains barriers.
static __device__ void net_backward(const float *__restrict__ p, float *gp, const KShape &s,
                             const float *uin, const float *acts, float *gA, float *gB, float *gin,
                             int TB, int TBP, int t, int nthreads, bool first, float *gcond = nullptr) {
    float *gcur = gA, *gprev = gB;
    int aoff = s.hs;   // running offset (in features) of Linear k's own activation block
    for (int k = s.nh; k >= 0; --k) {
        const int nin = s.nin[k], nout = s.nout[k];
        if (k < s.nh) aoff -= nout;
        const float *ak = acts + aoff * TBP;                               // act output of Linear k
        const float *inp = (k == 0) ? uin : acts + (aoff - s.nin[k]) * TBP;  // its input
        if (k < s.nh && t < TB) {
            for (int q = 0; q < nout; ++q) {
                const float a = ak[q * TBP + t];
                const float g = gcur[q * TBP + t];
                gcur[q * TBP + t] = (s.act == RNVP_ACT_TANH) ? g * (1.f - a * a) : (a > 0.f ? g : 0.f);
            }
        }
        __syncthreads();
        // weight gradient: thread = (q, i)
        float *gW = gp + s.woff[k], *gb = gp + s.boff[k];
        for (int idx = t; idx < nout * nin; idx += nthreads) {
            const int q = idx / nin, i = idx - q * nin;
            const float *gq = gcur + q * TBP, *xi = inp + i * TBP;
            float a = 0.f;
            for (int r = 0; r < TB; ++r) a = fmaf(gq[r], xi[r], a);
            gW[idx] = first ? a : gW[idx] + a;
        }
        for (int q = t; q < nout; q += nthreads) {
            const float *gq = gcur + q * TBP;
            float a = 0.f;
            for (int r = 0; r < TB; ++r) a += gq[r];
            gb[q] = first ? a : gb[q] + a;
        }
        // input gradient: thread = row.  For Linear 0 only the x part is needed (C gets none) unless the caller asked for
        // the conditions' gradient (gcond).
        if (t < TB) {
            const float *__restrict__ W = p + s.woff[k];
            const int ni = (k == 0) ? (gcond ? nin : s.d) : nin;
            float *dst = (k == 0) ? gin : gprev;
            for (int i = 0; i < ni; i += 4) {
                const int i1 = min(i + 1, ni - 1), i2 = min(i + 2, ni - 1), i3 = min(i + 3, ni - 1);
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
                for (int q = 0; q < nout; ++q) {
                    const float g = gcur[q * TBP + t];
                    const float *w = W + q * nin;
                    a0 = fmaf(g, w[i], a0); a1 = fmaf(g, w[i1], a1);
                    a2 = fmaf(g, w[i2], a2); a3 = fmaf(g, w[i3], a3);
                }
                if (k == 0) {
                    const float av[4] = {a0, a1, a2, a3};
                    for (int u = 0; u < 4 && i + u < ni; ++u) {
                        if (i + u < s.d) dst[(i + u) * TBP + t] += av[u];
                        else gcond[(i + u - s.d) * TBP + t] += av[u];
                    }
                } else {
                    dst[i * TBP + t] = a0;
                    if (i + 1 < ni) dst[(i + 1) * TBP + t] = a1;
                    if (i + 2 < ni) dst[(i + 2) * TBP + t] = a2;
                    if (i + 3 < ni) dst[(i + 3) * TBP + t] = a3;
                }
            }
        }
        float *tmp = gcur; gcur = gprev; gprev = tmp;
    }
    __syncthreads();
}


}  // namespace rnvp
