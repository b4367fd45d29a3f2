// rnvp_resident.hip -- "resident" fit: one epoch of a SMALL flow at a SMALL batch size in ONE persistent workgroup (gfx950).
//
// The reference's defaults -- hidden=(10,), 8 layers, batch_size=32 (/root/reference/probaforms/models/realnvp.py:161-176)
// -- are a flow of ~1000 parameters stepped on a few dozen rows: a step is a chain of ~50 tiny dependent GEMMs, and the
// three launches of the general path (weight re-pack, loss + gradient, reduce + Adam) cost more than the arithmetic
// (41 us per step for d = 2, h = 10).  A CU's 160 KB of LDS holds such a model whole, so the batch loop of RealNVP.fit
// (realnvp.py:237-254) runs here as ONE launch per epoch:
//   * the flat parameters (reference order) live in LDS for the whole epoch, and so do Adam's moments when they fit;
//   * wave w owns rows 16w .. 16w+15 of every batch and runs its forward / backward chain register to register on
//     v_mfma_f32_16x16x4_f32 (layout below): no workgroup barrier inside a step;
//   * the weight gradients of the wave's 16 rows are a third kind of MFMA contraction (over the rows, operands
//     transposed through wave-private LDS tiles), written to the wave's stage in the reference's flat order;
//   * one barrier, then all threads add the stages in wave order (deterministic), apply Adam in place in LDS and write
//     the batch loss; one more barrier and the next batch starts (its rows were requested a step ahead).  Parameters and
//     moments go back to HBM once per epoch.
// A single wave issues one VALU instruction per 4 cycles, so what bounds a step here is its INSTRUCTION COUNT (about
// 180 per layer forward, 900 backward): everything that does not depend on the data -- gather offsets, padding masks,
// stage offsets, mask bits -- is computed once per launch, and the next layer's weight fragments are requested while the
// current layer computes.  Measured (scripts/resident_time.py, profiles/): 21 us per step for the defaults.
// Same arithmetic per element as the other kernel families (tanh through exp2 / rcp, torch.optim.Adam as separately
// rounded operations); the summation ORDER over rows and hidden units differs, so results agree with them to rounding,
// not bit for bit -- run to run this path is bit-reproducible.
#include "rnvp_common.h"
#include "rnvp_resident.h"

#include <cmath>

namespace rnvp {
namespace resident {
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

// ---- layout: ONE hidden layer of at most 32 units, d <= 16, d + cdim <= 31 ----
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
constexpr int kDump = 64;              // per-net dump zone of the stage: where the padding lanes of a weight-gradient tile write

struct RcPlan {
    int W, P, mv_lds;
    int stg_net;                                       // floats of one net's stage block: npn + kDump
    int oPAR, oM, oV, oSTG, oRED, oXS, oTT;            // float offsets
    int xs_floats, tt_floats, stg_floats;              // per wave
    int total_floats;
};

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
            else { mv[u] = exp_avg[pp[u]]; vv[u] = exp_avg_sq[pp[u]]; }
        }
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

// MT hidden tiles, KIT k-steps of the net input (4 KIT >= d + cdim + 1), WMAX waves, DT slots of x that hold features
// (4 DT >= d: 2-d data needs one of the four)
template <int MT, int KIT, int ACT, int WMAX, int DT>
__global__ void __launch_bounds__(64 * WMAX)
k_fit_resident_rc(KShape s, RcPlan pl, float *__restrict__ params, const uint8_t *__restrict__ masks, const float *__restrict__ x,
                  const float *__restrict__ c, const int64_t *__restrict__ perm, int64_t n, int64_t batch,
                  float *__restrict__ loss_hist, float *__restrict__ exp_avg, float *__restrict__ exp_avg_sq, double lr, double beta1,
                  double beta2, double eps, double wd, double b1t, double b2t) {
    constexpr int NIT = KIT > 4 ? 2 : 1;               // 16-element tiles of the net input
    constexpr int KXT = KIT < DT ? KIT : DT;           // k-steps over the features of x
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, nthreads = blockDim.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, r = lane & 15, i = r;
    const int d = s.d, cd = s.c, L = s.L, P = pl.P, h = s.nout[0], nin0 = d + cd, npn = s.npn;
    const int w0 = s.woff[0], b0 = s.boff[0], w1 = s.woff[1], b1o = s.boff[1];
    const int pi = 4 * (i & 3) + (i >> 2);                // the element a fragment's row i stands for
    float *PAR = lds + pl.oPAR, *MM = lds + pl.oM, *VV = lds + pl.oV, *STG = lds + pl.oSTG + (size_t)wave * pl.stg_floats;
    float *RED = lds + pl.oRED;
    f4 *XS = reinterpret_cast<f4 *>(lds + pl.oXS + (size_t)wave * pl.xs_floats);
    float *TT = lds + pl.oTT + (size_t)wave * pl.tt_floats;
    float *T_in = TT, *T_go = TT + NIT * 16 * TS, *T_h = T_go + 2 * 16 * TS;      // [input tiles][g_out t, s][h, g_pre of t; of s]
    for (int e = tid; e < pl.total_floats; e += nthreads) lds[e] = 0.f;
    __syncthreads();
    for (int p = tid; p < P; p += nthreads) {
        PAR[p] = params[p];
        if (pl.mv_lds) { MM[p] = exp_avg[p]; VV[p] = exp_avg_sq[p]; }
    }
    __syncthreads();
    const float prior_c = 0.5f * (float)d * kLog2Pi;
    const int64_t nb = (n + batch - 1) / batch;

    // ---- per-lane constants: gather offsets (floats, relative to a net's parameter block), padding masks, stage offsets ----
    const int gW1 = w0 + pi * nin0 + q;                   // + m 16 nin0 + 4k : W1[16m + pi][4k + q]          (GEMM1)
    const int gW2 = w1 + pi * h + q;                      // + 16m + 4e      : W2[pi][16m + 4e + q]          (GEMM2)
    const int gW1t = w0 + q * nin0 + pi;                  // + (16m + 4e) nin0 : W1[16m + 4e + q][pi]        (g_in)
    const int gB1 = b0 + q, gB2 = b1o + q;                // + 16m + 4e / + 4e
    int gW2t[4];                                          // + 16m           : W2[4e + q][16m + pi]          (g_h)
#pragma unroll
    for (int e = 0; e < 4; ++e) gW2t[e] = w1 + (4 * e + q) * h + pi;
    f4 hm[MT];                                            // 1 for the real hidden units of a tile, 0 for its padding
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) hm[m][e] = 16 * m + 4 * e + q < h ? 1.f : 0.f;
    uint64_t mbits = 0;                                   // bit 4l + e: mask[l][4e + q]; padding features count as masked (passed through)
    for (int l = 0; l < L; ++l)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = 4 * e + q;
            if (j >= d || masks[l * d + j]) mbits |= 1ull << (4 * l + e);
        }
    bool xok[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) xok[e] = 4 * e + q < d;
    // stage offsets (relative to a net's stage block) of the weight-gradient tiles' D registers
    int sS2[MT][4], sS1[MT][NIT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int fo = 4 * q + e, hid_n = 16 * m + i, hid_m = 16 * m + 4 * q + e;
            sS2[m][e] = (fo < d && hid_n < h) ? w1 + fo * h + hid_n : npn + lane;                    // d W2[fo][hid]
#pragma unroll
            for (int nt = 0; nt < NIT; ++nt) {
                const int j = 16 * nt + i;
                sS1[m][nt][e] = hid_m >= h ? npn + lane : (j < nin0 ? w0 + hid_m * nin0 + j : (j == nin0 ? b0 + hid_m : npn + lane));
            }
        }

    // rows of a batch for this lane (zeros past d / cdim / the batch): x, and the condition placed behind x in the net input
    auto row_of = [&](int64_t kb) -> int64_t {
        if (kb >= nb) return -1;
        const int64_t s0 = kb * batch;
        const int64_t rows = (n - s0 < batch) ? n - s0 : batch;
        const int64_t rr = (int64_t)wave * 16 + r;
        return rr < rows ? perm[s0 + rr] : -1;
    };
    auto load_rows = [&](int64_t src, f4 &xo, f4 (&co)[NIT]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) xo[e] = (src >= 0 && xok[e]) ? x[src * d + 4 * e + q] : 0.f;
#pragma unroll
        for (int nt = 0; nt < NIT; ++nt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = 16 * nt + 4 * e + q;
                co[nt][e] = (src >= 0 && j >= d && j < nin0) ? c[src * cd + (j - d)] : 0.f;
            }
    };
    // A fragments of one layer, both nets: forward (biases, W1, W2) and backward (W2^T for g_h, W1^T for g_in)
    struct FwdW { float b1[2][MT][4], a1[2][MT][KIT], b2[2][4], a2[2][MT][4]; };
    struct BwdW { float a2t[2][MT][KXT], a1t[2][MT][4]; };
    auto load_fwd = [&](int l, FwdW &w) {
#pragma unroll
        for (int net = 0; net < 2; ++net) {
            const float *pn = PAR + (size_t)(2 * l + net) * npn;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { w.b1[net][m][e] = pn[gB1 + 16 * m + 4 * e]; w.a2[net][m][e] = pn[gW2 + 16 * m + 4 * e]; }
#pragma unroll
                for (int k = 0; k < KIT; ++k) w.a1[net][m][k] = pn[gW1 + m * 16 * nin0 + 4 * k];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) w.b2[net][e] = pn[gB2 + 4 * e];
        }
    };
    auto load_bwd = [&](int l, BwdW &w) {
#pragma unroll
        for (int net = 0; net < 2; ++net) {
            const float *pn = PAR + (size_t)(2 * l + net) * npn;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
#pragma unroll
                for (int e = 0; e < KXT; ++e) w.a2t[net][m][e] = pn[gW2t[e] + 16 * m];
#pragma unroll
                for (int e = 0; e < 4; ++e) w.a1t[net][m][e] = pn[gW1t + (16 * m + 4 * e) * nin0];
            }
        }
    };
    // both nets of one layer, their two independent chains interleaved instruction by instruction: hidden activations
    // hh[net][m]; outputs o[1] of the s net and (need_t) o[0] of the t net
    auto nets_fwd = [&](const FwdW &w, const f4 (&in)[NIT], f4 (&hh)[2][MT], f4 (&o)[2], auto need_t) {
        constexpr int N0 = decltype(need_t)::value ? 0 : 1;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            f4 acc[2];
#pragma unroll
            for (int net = 0; net < 2; ++net)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[net][e] = w.b1[net][m][e];
#pragma unroll
            for (int k = 0; k < KIT; ++k)
#pragma unroll
                for (int net = 0; net < 2; ++net) acc[net] = mfma16(w.a1[net][m][k], in[k >> 2][k & 3], acc[net]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int net = 0; net < 2; ++net) hh[net][m][e] = actf<ACT>(acc[net][e]) * hm[m][e];
        }
#pragma unroll
        for (int net = N0; net < 2; ++net)
#pragma unroll
            for (int e = 0; e < 4; ++e) o[net][e] = w.b2[net][e];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int net = N0; net < 2; ++net) o[net] = mfma16(w.a2[net][m][e], hh[net][m][e], o[net]);
    };

    // two batches ahead: row indices; one batch ahead: the rows themselves
    int64_t src_next = row_of(0);
    f4 nxq, ncq[NIT];
    load_rows(src_next, nxq, ncq);
    src_next = row_of(1);
    for (int64_t kb = 0; kb < nb; ++kb) {
        const int64_t s0 = kb * batch;
        const int rows = (int)((n - s0 < batch) ? n - s0 : batch);
        const float inv_B = 1.0f / (float)rows;
        const int nw = (rows + 15) >> 4;
        f4 xq = nxq, cin[NIT];
#pragma unroll
        for (int nt = 0; nt < NIT; ++nt) cin[nt] = ncq[nt];
        load_rows(src_next, nxq, ncq);
        src_next = row_of(kb + 2);
#ifdef RC_STAMP
        unsigned long long ts0 = __builtin_readcyclecounter(), ts1 = ts0, ts2 = ts0, ts3 = ts0, ts4 = ts0;
#endif
        if (wave < nw) {
            const bool valid = wave * 16 + r < rows;
            float ld = 0.f;
            // ---- forward (realnvp.py:91-101, nflow.py:107-117); the next layer's fragments are requested a layer ahead ----
            FwdW fw;
            load_fwd(0, fw);
            for (int l = 0; l < L; ++l) {
                const uint32_t mb = (uint32_t)(mbits >> (4 * l)) & 15u;
                XS[l * 64 + lane] = xq;                                    // layer input, for the backward (lane-private)
                f4 in[NIT];
                in[0] = cin[0];
#pragma unroll
                for (int e = 0; e < DT; ++e) in[0][e] = ((mb >> e) & 1u) ? xq[e] + cin[0][e] : cin[0][e];     // [x * mask | c]
                if (NIT > 1) in[NIT - 1] = cin[NIT - 1];
                FwdW fn;
                if (MT == 1) load_fwd(l + 1 < L ? l + 1 : l, fn);
                f4 hh[2][MT], o[2];
                nets_fwd(fw, in, hh, o, std::true_type{});
#pragma unroll
                for (int e = 0; e < DT; ++e) {
                    const bool mk = (mb >> e) & 1u;
                    const float xn = fmaf(xq[e], exp_acc(o[1][e]), o[0][e]);
                    xq[e] = mk ? xq[e] : xn;
                    ld += mk ? 0.f : o[1][e];
                }
                if (MT == 1) fw = fn;
                else if (l + 1 < L) load_fwd(l + 1, fw);
            }
            f4 gy;
            {
                float ss = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) gy[e] = 0.f;
#pragma unroll
                for (int e = 0; e < DT; ++e) { ss = fmaf(xq[e], xq[e], ss); gy[e] = valid ? xq[e] * inv_B : 0.f; }
                ld += __shfl_xor(ld, 16); ld += __shfl_xor(ld, 32);
                ss += __shfl_xor(ss, 16); ss += __shfl_xor(ss, 32);
                float v = (valid && q == 0) ? ld + (-0.5f * ss - prior_c) : 0.f;
                v = row16_sum(v);
                if (lane == 0) RED[wave] = v;
            }
#ifdef RC_STAMP
            __builtin_amdgcn_sched_barrier(0); ts1 = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0);
#endif
            const float gld = valid ? -inv_B : 0.f;
            f4 cinT[NIT];                                  // the condition part of the input TILE: + the ones element behind it (d b1)
#pragma unroll
            for (int nt = 0; nt < NIT; ++nt)
#pragma unroll
                for (int e = 0; e < 4; ++e) cinT[nt][e] = (16 * nt + 4 * e + q == nin0) ? 1.f : cin[nt][e];
            if (NIT > 1) { wfence(); tile_put(T_in + (NIT - 1) * 16 * TS, cinT[NIT - 1], q, r); }
            // ---- backward, the two nets of a layer side by side ----
            FwdW bf;
            BwdW bb;
            load_fwd(L - 1, bf);
            load_bwd(L - 1, bb);
            for (int l = L - 1; l >= 0; --l) {
                float *stg0 = STG + (size_t)l * 2 * pl.stg_net;
                const uint32_t mb = (uint32_t)(mbits >> (4 * l)) & 15u;
                xq = XS[l * 64 + lane];
                FwdW nf;
                BwdW nbw;
                if (MT == 1) { const int lp = l > 0 ? l - 1 : 0; load_fwd(lp, nf); load_bwd(lp, nbw); }
                f4 in[NIT], in0T;
                in[0] = cin[0]; in0T = cinT[0];
#pragma unroll
                for (int e = 0; e < DT; ++e) {
                    const bool mk = (mb >> e) & 1u;
                    in[0][e] = mk ? xq[e] + cin[0][e] : cin[0][e];
                    in0T[e] = mk ? xq[e] + cinT[0][e] : cinT[0][e];
                }
                if (NIT > 1) in[NIT - 1] = cin[NIT - 1];
                wfence();
                tile_put(T_in, in0T, q, r);
                f4 hh[2][MT], o[2];
                nets_fwd(bf, in, hh, o, std::false_type{});
                f4 es, go[2];                               // d loss / d (net output): s: (1-m)(gy x e^s + gld), t: (1-m) gy
                go[0] = f4{0.f, 0.f, 0.f, 0.f}; go[1] = f4{0.f, 0.f, 0.f, 0.f}; es = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < DT; ++e) {
                    const bool mk = (mb >> e) & 1u;
                    es[e] = exp_acc(o[1][e]);
                    go[1][e] = mk ? 0.f : fmaf(gy[e] * xq[e], es[e], gld);
                    go[0][e] = mk ? 0.f : gy[e];
                }
#pragma unroll
                for (int e = 0; e < DT; ++e) {              // the other elements of the two tiles stay at their initial zeros
                    T_go[(4 * e + q) * TS + r] = go[0][e];
                    T_go[16 * TS + (4 * e + q) * TS + r] = go[1][e];
                }
                f4 gp[2][MT], gin[2];
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    f4 gh[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                    for (int e = 0; e < KXT; ++e)
#pragma unroll
                        for (int net = 0; net < 2; ++net) gh[net] = mfma16(bb.a2t[net][m][e], go[net][e], gh[net]);
#pragma unroll
                    for (int net = 0; net < 2; ++net)
#pragma unroll
                        for (int e = 0; e < 4; ++e) gp[net][m][e] = gh[net][e] * dactf<ACT>(hh[net][m][e]) * hm[m][e];
                }
                gin[0] = f4{0.f, 0.f, 0.f, 0.f}; gin[1] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int net = 0; net < 2; ++net) gin[net] = mfma16(bb.a1t[net][m][e], gp[net][m][e], gin[net]);
#pragma unroll
                for (int net = 0; net < 2; ++net)
#pragma unroll
                    for (int e = 0; e < DT; ++e) {          // d b2
                        const float v = row16_sum(go[net][e]);
                        if (r == 0 && xok[e]) stg0[net * pl.stg_net + b1o + 4 * e + q] = v;
                    }
                // weight gradients: contractions over the tile's 16 rows through the transposition tiles
                wfence();
                float inT[NIT][4], goT[2][4];
#pragma unroll
                for (int nt = 0; nt < NIT; ++nt) tile_get(T_in + nt * 16 * TS, q, i, inT[nt]);
                tile_get(T_go, q, i, goT[0]);
                tile_get(T_go + 16 * TS, q, i, goT[1]);
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    wfence();
#pragma unroll
                    for (int net = 0; net < 2; ++net) {
                        tile_put(T_h + net * 32 * TS, hh[net][m], q, r);
                        tile_put(T_h + net * 32 * TS + 16 * TS, gp[net][m], q, r);
                    }
                    wfence();
                    float hT[2][4], gpT[2][4];
#pragma unroll
                    for (int net = 0; net < 2; ++net) {
                        tile_get(T_h + net * 32 * TS, q, i, hT[net]);
                        tile_get(T_h + net * 32 * TS + 16 * TS, q, i, gpT[net]);
                    }
                    f4 dw2[2], dw1[2][NIT];
#pragma unroll
                    for (int net = 0; net < 2; ++net) {
                        dw2[net] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int nt = 0; nt < NIT; ++nt) dw1[net][nt] = f4{0.f, 0.f, 0.f, 0.f};
                    }
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                        for (int net = 0; net < 2; ++net) {
                            dw2[net] = mfma16(goT[net][ks], hT[net][ks], dw2[net]);                        // [out feature 4q+e][hidden 16m + i]
#pragma unroll
                            for (int nt = 0; nt < NIT; ++nt)
                                dw1[net][nt] = mfma16(gpT[net][ks], inT[nt][ks], dw1[net][nt]);            // [hidden 16m + 4q+e][input 16nt + i]
                        }
#pragma unroll
                    for (int net = 0; net < 2; ++net) {
                        float *stg = stg0 + net * pl.stg_net;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            stg[sS2[m][e]] = dw2[net][e];
#pragma unroll
                            for (int nt = 0; nt < NIT; ++nt) stg[sS1[m][nt][e]] = dw1[net][nt][e];
                        }
                    }
                }
#pragma unroll
                for (int e = 0; e < DT; ++e) {
                    const bool mk = (mb >> e) & 1u;
                    gy[e] = xok[e] ? (mk ? gy[e] + (gin[1][e] + gin[0][e]) : gy[e] * es[e]) : 0.f;          // the nets see x * mask
                }
                if (MT == 1) { bf = nf; bb = nbw; }
                else if (l > 0) { load_fwd(l - 1, bf); load_bwd(l - 1, bb); }
            }
        }
#ifdef RC_STAMP
        __builtin_amdgcn_sched_barrier(0); ts2 = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0);
#endif
        __syncthreads();
#ifdef RC_STAMP
        __builtin_amdgcn_sched_barrier(0); ts3 = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0);
#endif
        {
            const AdamK a = step_adam(lr, beta1, beta2, eps, wd, b1t, b2t);
            adam_phase(lds + pl.oSTG, pl.stg_floats, pl.stg_net, npn, P, nw, PAR, MM, VV, pl.mv_lds != 0, exp_avg, exp_avg_sq, a, tid, nthreads);
            if (tid == 0) {
                float acc = 0.f;
                for (int w = 0; w < nw; ++w) acc += RED[w];
                loss_hist[kb] = -acc * inv_B;
            }
        }
#ifdef RC_STAMP
        __builtin_amdgcn_sched_barrier(0); ts4 = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0);
#endif
        __syncthreads();
#ifdef RC_STAMP
        if (kb == 20 && lane == 0)
            printf("RCSTAMP wave %d: fwd %llu bwd %llu barrier %llu adam %llu barrier2+top %llu\n", wave, ts1 - ts0, ts2 - ts1, ts3 - ts2, ts4 - ts3,
                   (unsigned long long)__builtin_readcyclecounter() - ts4);
#endif
    }
    for (int p = tid; p < P; p += nthreads) {
        params[p] = PAR[p];
        if (pl.mv_lds) { exp_avg[p] = MM[p]; exp_avg_sq[p] = VV[p]; }
    }
}

// ---- two or three hidden layers of at most 16 units each (hidden=(10, 10), (16, 16, 16), ...): the same chain, one GEMM longer per
// hidden layer.  Without this form such a flow falls to the any-shape kernels' five launches per step (240 us at batch 32):
// a 14x cliff next to hidden=(10,).  A hidden -> hidden Linear is one 16x16 tile: its D operand is the next Linear's B
// operand exactly like the last Linear's, its weight gradient one more contraction over the rows, its bias gradient a DPP
// row sum.  Fragments are loaded layer by layer (no look-ahead: three Linears' worth per net would not fit 256 registers).
template <int NH, int KIT, int ACT, int WMAX, int DT>
__global__ void __launch_bounds__(64 * WMAX)
k_fit_resident_deep(KShape s, RcPlan pl, float *__restrict__ params, const uint8_t *__restrict__ masks, const float *__restrict__ x,
                    const float *__restrict__ c, const int64_t *__restrict__ perm, int64_t n, int64_t batch,
                    float *__restrict__ loss_hist, float *__restrict__ exp_avg, float *__restrict__ exp_avg_sq, double lr, double beta1,
                    double beta2, double eps, double wd, double b1t, double b2t) {
    constexpr int NIT = KIT > 4 ? 2 : 1;
    constexpr int KXT = KIT < DT ? KIT : DT;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, nthreads = blockDim.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, r = lane & 15, i = r;
    const int d = s.d, cd = s.c, L = s.L, P = pl.P, nin0 = d + cd, npn = s.npn;
    const int pi = 4 * (i & 3) + (i >> 2);
    float *PAR = lds + pl.oPAR, *MM = lds + pl.oM, *VV = lds + pl.oV, *STG = lds + pl.oSTG + (size_t)wave * pl.stg_floats;
    float *RED = lds + pl.oRED;
    f4 *XS = reinterpret_cast<f4 *>(lds + pl.oXS + (size_t)wave * pl.xs_floats);
    float *TT = lds + pl.oTT + (size_t)wave * pl.tt_floats;
    float *T_in = TT, *T_g = TT + NIT * 16 * TS, *T_h = T_g + 2 * 16 * TS;        // [input tiles][g of t, s][h of t, s]
    for (int e = tid; e < pl.total_floats; e += nthreads) lds[e] = 0.f;
    __syncthreads();
    for (int p = tid; p < P; p += nthreads) {
        PAR[p] = params[p];
        if (pl.mv_lds) { MM[p] = exp_avg[p]; VV[p] = exp_avg_sq[p]; }
    }
    __syncthreads();
    const float prior_c = 0.5f * (float)d * kLog2Pi;
    const int64_t nb = (n + batch - 1) / batch;

    // ---- per-lane constants.  Linear k maps nin_k -> nout_k (k = 0: the net input; k = NH: the d outputs) ----
    int gF[NH + 1], gB[NH + 1], gT[NH + 1][4], sS[NH + 1][4], sS0[NIT][4], sBk[NH + 1][4];
    f4 hm[NH];
    const int dump = npn + lane;
#pragma unroll
    for (int k = 0; k <= NH; ++k) {
        const int nin = s.nin[k], nout = s.nout[k], wo = s.woff[k], bo = s.boff[k];
        gF[k] = wo + pi * nin + q;                       // + 4e : W_k[pi][4e + q]           (forward)
        gB[k] = bo + q;                                  // + 4e
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            gT[k][e] = wo + (4 * e + q) * nin + pi;      // W_k[4e + q][pi]                  (gradient of Linear k's input)
            sS[k][e] = (4 * q + e < nout && i < nin) ? wo + (4 * q + e) * nin + i : dump;        // d W_k[4q + e][i]
            sBk[k][e] = (4 * e + q < nout) ? bo + 4 * e + q : dump;                               // d b_k[4e + q] (lanes r == 0)
        }
        if (k < NH)
#pragma unroll
            for (int e = 0; e < 4; ++e) hm[k][e] = 4 * e + q < nout ? 1.f : 0.f;
    }
#pragma unroll
    for (int nt = 0; nt < NIT; ++nt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int u = 4 * q + e, j = 16 * nt + i;
            sS0[nt][e] = u >= s.nout[0] ? dump : (j < nin0 ? s.woff[0] + u * nin0 + j : (j == nin0 ? s.boff[0] + u : dump));
        }
    uint64_t mbits = 0;
    for (int l = 0; l < L; ++l)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = 4 * e + q;
            if (j >= d || masks[l * d + j]) mbits |= 1ull << (4 * l + e);
        }
    bool xok[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) xok[e] = 4 * e + q < d;

    auto row_of = [&](int64_t kb) -> int64_t {
        if (kb >= nb) return -1;
        const int64_t s0 = kb * batch;
        const int64_t rows = (n - s0 < batch) ? n - s0 : batch;
        const int64_t rr = (int64_t)wave * 16 + r;
        return rr < rows ? perm[s0 + rr] : -1;
    };
    auto load_rows = [&](int64_t src, f4 &xo, f4 (&co)[NIT]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) xo[e] = (src >= 0 && xok[e]) ? x[src * d + 4 * e + q] : 0.f;
#pragma unroll
        for (int nt = 0; nt < NIT; ++nt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = 16 * nt + 4 * e + q;
                co[nt][e] = (src >= 0 && j >= d && j < nin0) ? c[src * cd + (j - d)] : 0.f;
            }
    };
    // both nets of one layer, interleaved: hidden activations hh[net][k]; outputs o[1] (s) and, if asked for, o[0] (t)
    auto nets_fwd = [&](const float *pl0, const f4 (&in)[NIT], f4 (&hh)[2][NH], f4 (&o)[2], auto need_t) {
        constexpr int N0 = decltype(need_t)::value ? 0 : 1;
        f4 acc[2];
#pragma unroll
        for (int net = 0; net < 2; ++net)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[net][e] = pl0[net * npn + gB[0] + 4 * e];
#pragma unroll
        for (int k = 0; k < KIT; ++k)
#pragma unroll
            for (int net = 0; net < 2; ++net) acc[net] = mfma16(pl0[net * npn + gF[0] + 4 * k], in[k >> 2][k & 3], acc[net]);
#pragma unroll
        for (int net = 0; net < 2; ++net)
#pragma unroll
            for (int e = 0; e < 4; ++e) hh[net][0][e] = actf<ACT>(acc[net][e]) * hm[0][e];
#pragma unroll
        for (int k = 1; k < NH; ++k) {
#pragma unroll
            for (int net = 0; net < 2; ++net)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[net][e] = pl0[net * npn + gB[k] + 4 * e];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int net = 0; net < 2; ++net) acc[net] = mfma16(pl0[net * npn + gF[k] + 4 * e], hh[net][k - 1][e], acc[net]);
#pragma unroll
            for (int net = 0; net < 2; ++net)
#pragma unroll
                for (int e = 0; e < 4; ++e) hh[net][k][e] = actf<ACT>(acc[net][e]) * hm[k][e];
        }
#pragma unroll
        for (int net = N0; net < 2; ++net)
#pragma unroll
            for (int e = 0; e < 4; ++e) o[net][e] = pl0[net * npn + gB[NH] + 4 * e];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int net = N0; net < 2; ++net) o[net] = mfma16(pl0[net * npn + gF[NH] + 4 * e], hh[net][NH - 1][e], o[net]);
    };

    int64_t src_next = row_of(0);
    f4 nxq, ncq[NIT];
    load_rows(src_next, nxq, ncq);
    src_next = row_of(1);
    for (int64_t kb = 0; kb < nb; ++kb) {
        const int64_t s0 = kb * batch;
        const int rows = (int)((n - s0 < batch) ? n - s0 : batch);
        const float inv_B = 1.0f / (float)rows;
        const int nw = (rows + 15) >> 4;
        f4 xq = nxq, cin[NIT];
#pragma unroll
        for (int nt = 0; nt < NIT; ++nt) cin[nt] = ncq[nt];
        load_rows(src_next, nxq, ncq);
        src_next = row_of(kb + 2);
        if (wave < nw) {
            const bool valid = wave * 16 + r < rows;
            float ld = 0.f;
            for (int l = 0; l < L; ++l) {                                  // forward (realnvp.py:91-101, nflow.py:107-117)
                const uint32_t mb = (uint32_t)(mbits >> (4 * l)) & 15u;
                XS[l * 64 + lane] = xq;
                f4 in[NIT];
                in[0] = cin[0];
#pragma unroll
                for (int e = 0; e < DT; ++e) in[0][e] = ((mb >> e) & 1u) ? xq[e] + cin[0][e] : cin[0][e];
                if (NIT > 1) in[NIT - 1] = cin[NIT - 1];
                f4 hh[2][NH], o[2];
                nets_fwd(PAR + (size_t)l * 2 * npn, in, hh, o, std::true_type{});
#pragma unroll
                for (int e = 0; e < DT; ++e) {
                    const bool mk = (mb >> e) & 1u;
                    const float xn = fmaf(xq[e], exp_acc(o[1][e]), o[0][e]);
                    xq[e] = mk ? xq[e] : xn;
                    ld += mk ? 0.f : o[1][e];
                }
            }
            f4 gy;
            {
                float ss = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) gy[e] = 0.f;
#pragma unroll
                for (int e = 0; e < DT; ++e) { ss = fmaf(xq[e], xq[e], ss); gy[e] = valid ? xq[e] * inv_B : 0.f; }
                ld += __shfl_xor(ld, 16); ld += __shfl_xor(ld, 32);
                ss += __shfl_xor(ss, 16); ss += __shfl_xor(ss, 32);
                float v = (valid && q == 0) ? ld + (-0.5f * ss - prior_c) : 0.f;
                v = row16_sum(v);
                if (lane == 0) RED[wave] = v;
            }
            const float gld = valid ? -inv_B : 0.f;
            f4 cinT[NIT];
#pragma unroll
            for (int nt = 0; nt < NIT; ++nt)
#pragma unroll
                for (int e = 0; e < 4; ++e) cinT[nt][e] = (16 * nt + 4 * e + q == nin0) ? 1.f : cin[nt][e];
            if (NIT > 1) { wfence(); tile_put(T_in + (NIT - 1) * 16 * TS, cinT[NIT - 1], q, r); }
            for (int l = L - 1; l >= 0; --l) {                             // backward
                const float *pl0 = PAR + (size_t)l * 2 * npn;
                float *stg0 = STG + (size_t)l * 2 * pl.stg_net;
                const uint32_t mb = (uint32_t)(mbits >> (4 * l)) & 15u;
                xq = XS[l * 64 + lane];
                f4 in[NIT], in0T;
                in[0] = cin[0]; in0T = cinT[0];
#pragma unroll
                for (int e = 0; e < DT; ++e) {
                    const bool mk = (mb >> e) & 1u;
                    in[0][e] = mk ? xq[e] + cin[0][e] : cin[0][e];
                    in0T[e] = mk ? xq[e] + cinT[0][e] : cinT[0][e];
                }
                if (NIT > 1) in[NIT - 1] = cin[NIT - 1];
                wfence();
                tile_put(T_in, in0T, q, r);
                f4 hh[2][NH], o[2];
                nets_fwd(pl0, in, hh, o, std::false_type{});
                f4 es = f4{0.f, 0.f, 0.f, 0.f}, g[2];
                g[0] = f4{0.f, 0.f, 0.f, 0.f}; g[1] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < DT; ++e) {
                    const bool mk = (mb >> e) & 1u;
                    es[e] = exp_acc(o[1][e]);
                    g[1][e] = mk ? 0.f : fmaf(gy[e] * xq[e], es[e], gld);
                    g[0][e] = mk ? 0.f : gy[e];
                }
                // Linear k = NH .. 1: bias gradient (row sums), weight gradient g^T . h_{k-1}, gradient of h_{k-1}
#pragma unroll
                for (int k = NH; k >= 1; --k) {
                    constexpr int dummy = 0; (void)dummy;
                    const int ke = (k == NH) ? KXT : 4;                    // slots of g that hold units
#pragma unroll
                    for (int net = 0; net < 2; ++net)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (e < ke) {
                                const float v = row16_sum(g[net][e]);
                                if (r == 0) stg0[net * pl.stg_net + sBk[k][e]] = v;
                            }
                    wfence();
#pragma unroll
                    for (int net = 0; net < 2; ++net) {
                        tile_put(T_g + net * 16 * TS, g[net], q, r);
                        tile_put(T_h + net * 16 * TS, hh[net][k - 1], q, r);
                    }
                    f4 gh[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (e < ke)
#pragma unroll
                            for (int net = 0; net < 2; ++net) gh[net] = mfma16(pl0[net * npn + gT[k][e]], g[net][e], gh[net]);
                    wfence();
                    float gT_[2][4], hT_[2][4];
#pragma unroll
                    for (int net = 0; net < 2; ++net) { tile_get(T_g + net * 16 * TS, q, i, gT_[net]); tile_get(T_h + net * 16 * TS, q, i, hT_[net]); }
                    f4 dw[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                        for (int net = 0; net < 2; ++net) dw[net] = mfma16(gT_[net][ks], hT_[net][ks], dw[net]);      // [unit of k: 4q+e][unit of k-1: i]
#pragma unroll
                    for (int net = 0; net < 2; ++net)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            stg0[net * pl.stg_net + sS[k][e]] = dw[net][e];
                            g[net][e] = gh[net][e] * dactf<ACT>(hh[net][k - 1][e]) * hm[k - 1][e];
                        }
                }
                // Linear 0: weight + bias gradient against the input tile(s); input gradient for the x part
                f4 gin[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int net = 0; net < 2; ++net) gin[net] = mfma16(pl0[net * npn + gT[0][e]], g[net][e], gin[net]);
                wfence();
                tile_put(T_g, g[0], q, r);
                tile_put(T_g + 16 * TS, g[1], q, r);
                wfence();
                float g0T[2][4], inT[NIT][4];
                tile_get(T_g, q, i, g0T[0]); tile_get(T_g + 16 * TS, q, i, g0T[1]);
#pragma unroll
                for (int nt = 0; nt < NIT; ++nt) tile_get(T_in + nt * 16 * TS, q, i, inT[nt]);
                f4 dw0[2][NIT];
#pragma unroll
                for (int net = 0; net < 2; ++net)
#pragma unroll
                    for (int nt = 0; nt < NIT; ++nt) dw0[net][nt] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int net = 0; net < 2; ++net)
#pragma unroll
                        for (int nt = 0; nt < NIT; ++nt) dw0[net][nt] = mfma16(g0T[net][ks], inT[nt][ks], dw0[net][nt]);
#pragma unroll
                for (int net = 0; net < 2; ++net)
#pragma unroll
                    for (int nt = 0; nt < NIT; ++nt)
#pragma unroll
                        for (int e = 0; e < 4; ++e) stg0[net * pl.stg_net + sS0[nt][e]] = dw0[net][nt][e];
#pragma unroll
                for (int e = 0; e < DT; ++e) {
                    const bool mk = (mb >> e) & 1u;
                    gy[e] = xok[e] ? (mk ? gy[e] + (gin[1][e] + gin[0][e]) : gy[e] * es[e]) : 0.f;
                }
            }
        }
        __syncthreads();
        {
            const AdamK a = step_adam(lr, beta1, beta2, eps, wd, b1t, b2t);
            adam_phase(lds + pl.oSTG, pl.stg_floats, pl.stg_net, npn, P, nw, PAR, MM, VV, pl.mv_lds != 0, exp_avg, exp_avg_sq, a, tid, nthreads);
            if (tid == 0) {
                float acc = 0.f;
                for (int w = 0; w < nw; ++w) acc += RED[w];
                loss_hist[kb] = -acc * inv_B;
            }
        }
        __syncthreads();
    }
    for (int p = tid; p < P; p += nthreads) {
        params[p] = PAR[p];
        if (pl.mv_lds) { exp_avg[p] = MM[p]; exp_avg_sq[p] = VV[p]; }
    }
}

// k-steps of the net input incl. the ones element behind it
int rc_kit(const KShape &k) { const int ki = (k.d + k.c + 1 + 3) / 4; return ki <= 2 ? 2 : (ki <= 4 ? 4 : 8); }

bool make_rc_plan(const KShape &k, int64_t batch, RcPlan *out) {
    // measured against the batch-by-batch loop (scripts/resident_time.py, profiles/): one hidden tile wins everywhere
    // (21-37 vs 41-53 us per step), two win while the net input fits one tile (38 vs 49 us), beyond that the loop's
    // multi-workgroup kernels are faster (h = 64: 75 vs 37 us)
    if (k.nh < 1 || k.nh > 3 || k.d > 16 || k.d + k.c > 31 || k.L > 16) return false;
    if (k.nh == 1) {
        if (k.nout[0] > 32 || (k.nout[0] > 16 && rc_kit(k) > 4)) return false;
    } else {
        for (int i = 0; i < k.nh; ++i)          // two or three hidden layers: one tile each (k_fit_resident_deep)
            if (k.nout[i] > 16) return false;
    }
    if (batch < 1 || batch > 16 * kRcMaxWaves) return false;       // up to 8 waves: two per SIMD keep 256 registers each
    RcPlan p;
    std::memset(&p, 0, sizeof(p));
    p.W = (int)((batch + 15) / 16);
    p.P = 2 * k.npn * k.L;
    p.stg_net = k.npn + kDump;
    p.stg_floats = 2 * k.L * p.stg_net;
    p.xs_floats = k.L * 64 * 4;
    p.tt_floats = ((rc_kit(k) > 4 ? 2 : 1) + 6) * 16 * TS;
    for (int mv = 1; mv >= 0; --mv) {
        int f = 0;
        p.oPAR = f; f += p.P;
        p.oM = f; p.oV = f;
        if (mv) { p.oM = f; f += p.P; p.oV = f; f += p.P; }
        p.oSTG = f; f += p.W * p.stg_floats;
        p.oRED = f; f += kMaxWaves;
        f = (f + 3) & ~3;                       // the saved layer inputs are read and written as float4
        p.oXS = f; f += p.W * p.xs_floats;
        p.oTT = f; f += p.W * p.tt_floats;
        f += 16 * 64 + 64;                      // the unguarded gathers of padding lanes stay inside the allocation
        p.total_floats = f;
        p.mv_lds = mv;
        if ((size_t)f * sizeof(float) <= kLdsMax) { *out = p; return true; }
    }
    return false;
}

// ---- the same for the conditional VAE: encoder -> reparameterise -> decoder -> KL + MSE -> backward (cvae.py:186-203) -------
// One hidden layer per MLP (the reference's default hidden=(10,), latent 2, batch_size 32: cvae.py:145).  The two nets of a
// step run one after the other (the decoder eats z), nothing is recomputed or saved: a row's activations stay in registers
// from the encoder's first GEMM to the last weight gradient.  The encoder's head is gathered so that lane group q receives
// mu[q], mu[4+q] in slots 0, 1 and log_sigma[q], log_sigma[4+q] in slots 2, 3 (latent <= 8): z = mu + e^{ls/2} eps is
// lane-local and comes out in the element order the decoder's first GEMM wants.
struct CvPlan {
    int W, P, mv_lds;
    int oPAR, oM, oV, oSTG, oRED, oTT;                 // float offsets
    int tt_floats, stg_floats;                         // per wave
    int total_floats;
};

template <int MT, int KIT, int ACT, int WMAX>
__global__ void __launch_bounds__(64 * WMAX)
k_cvae_fit_resident(CvaeK s, CvPlan pl, float *__restrict__ params, const float *__restrict__ x, const float *__restrict__ c,
                    const int64_t *__restrict__ perm, const float *__restrict__ eps, int64_t n, int64_t batch, float klw,
                    float *__restrict__ loss_hist, float *__restrict__ exp_avg, float *__restrict__ exp_avg_sq, double lr, double beta1,
                    double beta2, double adam_eps, double wd, double b1t, double b2t) {
    constexpr int NIT = KIT > 4 ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, nthreads = blockDim.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, r = lane & 15, i = r;
    const int d = s.d, cd = s.c, lat = s.lat, P = pl.P, h = s.enc.nout[0], ne = d + cd, nd = lat + cd, pe = s.pe;
    const int w0e = s.enc.woff[0], b0e = s.enc.boff[0], w1e = s.enc.woff[1], b1e = s.enc.boff[1];
    const int w0d = pe + s.dec.woff[0], b0d = pe + s.dec.boff[0], w1d = pe + s.dec.woff[1], b1d = pe + s.dec.boff[1];
    const int pi = 4 * (i & 3) + (i >> 2);
    float *PAR = lds + pl.oPAR, *MM = lds + pl.oM, *VV = lds + pl.oV, *STG = lds + pl.oSTG + (size_t)wave * pl.stg_floats;
    float *RED = lds + pl.oRED;
    float *TT = lds + pl.oTT + (size_t)wave * pl.tt_floats;
    float *T_ine = TT, *T_ind = TT + NIT * 16 * TS, *T_go = T_ind + NIT * 16 * TS, *T_h = T_go + 2 * 16 * TS;      // [enc in][dec in][g head, g rec][h, g_pre]
    for (int e = tid; e < pl.total_floats; e += nthreads) lds[e] = 0.f;
    __syncthreads();
    for (int p = tid; p < P; p += nthreads) {
        PAR[p] = params[p];
        if (pl.mv_lds) { MM[p] = exp_avg[p]; VV[p] = exp_avg_sq[p]; }
    }
    __syncthreads();
    const int64_t nb = (n + batch - 1) / batch;

    // head position p (0..15) of the encoder's last Linear: p < 8: mu[p], else log_sigma[p - 8]; its row in W / b, or -1
    auto head_row = [&](int p) -> int { return p < 8 ? (p < lat ? p : -1) : (p - 8 < lat ? lat + p - 8 : -1); };
    // ---- per-lane constants ----
    const int gW1e = w0e + pi * ne + q, gW1d = w0d + pi * nd + q;                 // + m 16 nin + 4k
    const int gB1e = b0e + q, gB1d = b0d + q;                                     // + 16m + 4e
    const int hr_a = head_row(pi);
    const int gW2e = w1e + (hr_a < 0 ? 0 : hr_a) * h + q;                         // + 16m + 4e : W2e[row(pi)][16m + 4e + q]
    const int gW2d = w1d + pi * h + q;                                            // W2d[pi][16m + 4e + q]
    const int gB2d = b1d + q;                                                     // + 4e
    const int gW1td = w0d + q * nd + pi;                                          // + (16m + 4e) nd : W1d[16m + 4e + q][pi]   (g_z)
    int gB2e[4], gW2te[4], gW2td[4];
    float hv[4];                                                                  // 1 where head slot e of this lane group is a real unit
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int hr = head_row(4 * e + q);
        hv[e] = hr >= 0 ? 1.f : 0.f;
        gB2e[e] = b1e + (hr < 0 ? 0 : hr);
        gW2te[e] = w1e + (hr < 0 ? 0 : hr) * h + pi;                              // + 16m : W2e[row(4e + q)][16m + pi]      (g_h, encoder)
        gW2td[e] = w1d + (4 * e + q) * h + pi;                                    // + 16m : W2d[4e + q][16m + pi]           (g_h, decoder)
    }
    f4 hm[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) hm[m][e] = 16 * m + 4 * e + q < h ? 1.f : 0.f;
    bool xok[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) xok[e] = 4 * e + q < d;
    const int dump = P + lane;
    int sS2e[MT][4], sS2d[MT][4], sS1e[MT][NIT][4], sS1d[MT][NIT][4], sB2e[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { const int hr = head_row(4 * e + q); sB2e[e] = hr >= 0 ? b1e + hr : dump; }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int po = 4 * q + e, hid_n = 16 * m + i, hid_m = 16 * m + 4 * q + e, hr = head_row(po);
            sS2e[m][e] = (hr >= 0 && hid_n < h) ? w1e + hr * h + hid_n : dump;
            sS2d[m][e] = (po < d && hid_n < h) ? w1d + po * h + hid_n : dump;
#pragma unroll
            for (int nt = 0; nt < NIT; ++nt) {
                const int j = 16 * nt + i;
                sS1e[m][nt][e] = hid_m >= h ? dump : (j < ne ? w0e + hid_m * ne + j : (j == ne ? b0e + hid_m : dump));
                sS1d[m][nt][e] = hid_m >= h ? dump : (j < nd ? w0d + hid_m * nd + j : (j == nd ? b0d + hid_m : dump));
            }
        }

    auto row_of = [&](int64_t kb) -> int64_t {
        if (kb >= nb) return -1;
        const int64_t s0 = kb * batch;
        const int64_t rows = (n - s0 < batch) ? n - s0 : batch;
        const int64_t rr = (int64_t)wave * 16 + r;
        return rr < rows ? s0 + rr : -1;                                          // position in the epoch (eps is in that order)
    };
    // x; the condition behind x (encoder input) and behind z (decoder input); eps in the element order of z
    auto load_rows = [&](int64_t pos, f4 &xo, f4 (&ce)[NIT], f4 (&cdv)[NIT], f4 &eo) {
        const int64_t src = pos >= 0 ? perm[pos] : 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            xo[e] = (pos >= 0 && xok[e]) ? x[src * d + 4 * e + q] : 0.f;
            eo[e] = (pos >= 0 && e < 2 && 4 * e + q < lat) ? eps[pos * lat + 4 * e + q] : 0.f;
        }
#pragma unroll
        for (int nt = 0; nt < NIT; ++nt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = 16 * nt + 4 * e + q;
                ce[nt][e] = (pos >= 0 && j >= d && j < ne) ? c[src * cd + (j - d)] : 0.f;
                cdv[nt][e] = (pos >= 0 && j >= lat && j < nd) ? c[src * cd + (j - lat)] : 0.f;
            }
    };
    // hidden layer of one MLP: act(W1 in + b1), padding units zeroed
    auto hidden_fwd = [&](int gW1, int gB1, int nin, const f4 (&in)[NIT], f4 (&hh)[MT]) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            f4 acc;
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = PAR[gB1 + 16 * m + 4 * e];
#pragma unroll
            for (int k = 0; k < KIT; ++k) acc = mfma16(PAR[gW1 + m * 16 * nin + 4 * k], in[k >> 2][k & 3], acc);
#pragma unroll
            for (int e = 0; e < 4; ++e) hh[m][e] = actf<ACT>(acc[e]) * hm[m][e];
        }
    };

    f4 nx, nce[NIT], ncd[NIT], nep;
    load_rows(row_of(0), nx, nce, ncd, nep);
    for (int64_t kb = 0; kb < nb; ++kb) {
        const int64_t s0 = kb * batch;
        const int rows = (int)((n - s0 < batch) ? n - s0 : batch);
        const float inv_B = 1.0f / (float)rows;
        const int nw = (rows + 15) >> 4;
        const f4 xq = nx, epq = nep;
        f4 cinE[NIT], cinD[NIT];
#pragma unroll
        for (int nt = 0; nt < NIT; ++nt) { cinE[nt] = nce[nt]; cinD[nt] = ncd[nt]; }
        load_rows(row_of(kb + 1), nx, nce, ncd, nep);                             // the next batch's rows, a step ahead
        if (wave < nw) {
            const bool valid = wave * 16 + r < rows;
            const float sc = valid ? inv_B : 0.f;
            // ---- encoder (cvae.py:58-62) ----
            f4 inE[NIT];
            inE[0] = xq + cinE[0];
            if (NIT > 1) inE[NIT - 1] = cinE[NIT - 1];
            f4 hhe[MT];
            hidden_fwd(gW1e, gB1e, ne, inE, hhe);
            f4 oe;                                                                // slots 0, 1: mu; 2, 3: log_sigma
#pragma unroll
            for (int e = 0; e < 4; ++e) oe[e] = PAR[gB2e[e]];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int e = 0; e < 4; ++e) oe = mfma16(PAR[gW2e + 16 * m + 4 * e], hhe[m][e], oe);
            float mu[2], ls[2], el[2], els[2];
            f4 zq = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                mu[e] = oe[e] * hv[e]; ls[e] = oe[2 + e] * hv[2 + e];
                el[e] = exp_acc(0.5f * ls[e]); els[e] = exp_acc(ls[e]);
                zq[e] = hv[e] != 0.f ? fmaf(el[e], epq[e], mu[e]) : 0.f;           // sample_z, cvae.py:188
            }
            // ---- decoder (cvae.py:81-84) ----
            f4 inD[NIT];
            inD[0] = zq + cinD[0];
            if (NIT > 1) inD[NIT - 1] = cinD[NIT - 1];
            f4 hhd[MT];
            hidden_fwd(gW1d, gB1d, nd, inD, hhd);
            f4 xr;
#pragma unroll
            for (int e = 0; e < 4; ++e) xr[e] = PAR[gB2d + 4 * e];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int e = 0; e < 4; ++e) xr = mfma16(PAR[gW2d + 16 * m + 4 * e], hhd[m][e], xr);
            // ---- loss: KL_weight * KL + MSE (cvae.py:190-193) and its seeds ----
            f4 grec;
            {
                const float inv_d = 1.f / (float)d;
                float kl = 0.f, se = 0.f;
#pragma unroll
                for (int e = 0; e < 2; ++e) kl += hv[e] != 0.f ? 1.f + ls[e] - mu[e] * mu[e] - els[e] : 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float df = xok[e] ? xr[e] - xq[e] : 0.f;
                    se = fmaf(df, df, se);
                    grec[e] = (2.f * sc * inv_d) * df;                             // d MSE / d x_rec
                }
                float lrow = klw * (-0.5f * kl) + se * inv_d;
                lrow += __shfl_xor(lrow, 16); lrow += __shfl_xor(lrow, 32);
                float v = (valid && q == 0) ? lrow : 0.f;
                v = row16_sum(v);
                if (lane == 0) RED[wave] = v;
            }
            // ---- decoder backward ----
            f4 gpd[MT], gz = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                f4 gh = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) gh = mfma16(PAR[gW2td[e] + 16 * m], grec[e], gh);
#pragma unroll
                for (int e = 0; e < 4; ++e) gpd[m][e] = gh[e] * dactf<ACT>(hhd[m][e]) * hm[m][e];
            }
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int e = 0; e < 4; ++e) gz = mfma16(PAR[gW1td + (16 * m + 4 * e) * nd], gpd[m][e], gz);
            // ---- d loss / d mu, d log_sigma; encoder backward (no input gradient needed) ----
            f4 goe;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                goe[e] = hv[e] * fmaf(klw * sc, mu[e], gz[e]);
                goe[2 + e] = hv[2 + e] * (gz[e] * epq[e] * 0.5f * el[e] + klw * sc * (-0.5f) * (1.f - els[e]));
            }
            f4 gpe[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                f4 gh = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) gh = mfma16(PAR[gW2te[e] + 16 * m], goe[e], gh);
#pragma unroll
                for (int e = 0; e < 4; ++e) gpe[m][e] = gh[e] * dactf<ACT>(hhe[m][e]) * hm[m][e];
            }
            // ---- biases of the last Linears (row sums), then the weight gradients through the transposition tiles ----
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float ve = row16_sum(goe[e]), vd = row16_sum(grec[e]);
                if (r == 0) { STG[sB2e[e]] = ve; if (xok[e]) STG[b1d + 4 * e + q] = vd; }
            }
            wfence();
#pragma unroll
            for (int nt = 0; nt < NIT; ++nt) {
                f4 te = inE[nt], td = inD[nt];
#pragma unroll
                for (int e = 0; e < 4; ++e) {                                       // the ones element behind the inputs: d b1
                    if (16 * nt + 4 * e + q == ne) te[e] = 1.f;
                    if (16 * nt + 4 * e + q == nd) td[e] = 1.f;
                }
                tile_put(T_ine + nt * 16 * TS, te, q, r);
                tile_put(T_ind + nt * 16 * TS, td, q, r);
            }
            tile_put(T_go, goe, q, r);
            tile_put(T_go + 16 * TS, grec, q, r);
            wfence();
            float inTe[NIT][4], inTd[NIT][4], goT[2][4];
#pragma unroll
            for (int nt = 0; nt < NIT; ++nt) { tile_get(T_ine + nt * 16 * TS, q, i, inTe[nt]); tile_get(T_ind + nt * 16 * TS, q, i, inTd[nt]); }
            tile_get(T_go, q, i, goT[0]);
            tile_get(T_go + 16 * TS, q, i, goT[1]);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                wfence();
                tile_put(T_h, hhe[m], q, r);
                tile_put(T_h + 16 * TS, gpe[m], q, r);
                tile_put(T_h + 32 * TS, hhd[m], q, r);
                tile_put(T_h + 48 * TS, gpd[m], q, r);
                wfence();
                float hT[2][4], gpT[2][4];
                tile_get(T_h, q, i, hT[0]); tile_get(T_h + 16 * TS, q, i, gpT[0]);
                tile_get(T_h + 32 * TS, q, i, hT[1]); tile_get(T_h + 48 * TS, q, i, gpT[1]);
                f4 dw2[2], dw1[2][NIT];
#pragma unroll
                for (int net = 0; net < 2; ++net) {
                    dw2[net] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int nt = 0; nt < NIT; ++nt) dw1[net][nt] = f4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    dw2[0] = mfma16(goT[0][ks], hT[0][ks], dw2[0]);                 // [head position 4q+e][hidden 16m + i]
                    dw2[1] = mfma16(goT[1][ks], hT[1][ks], dw2[1]);                 // [out feature 4q+e][hidden 16m + i]
#pragma unroll
                    for (int nt = 0; nt < NIT; ++nt) {
                        dw1[0][nt] = mfma16(gpT[0][ks], inTe[nt][ks], dw1[0][nt]);  // [hidden 16m + 4q+e][encoder input 16nt + i]
                        dw1[1][nt] = mfma16(gpT[1][ks], inTd[nt][ks], dw1[1][nt]);  // [hidden 16m + 4q+e][decoder input 16nt + i]
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    STG[sS2e[m][e]] = dw2[0][e];
                    STG[sS2d[m][e]] = dw2[1][e];
#pragma unroll
                    for (int nt = 0; nt < NIT; ++nt) { STG[sS1e[m][nt][e]] = dw1[0][nt][e]; STG[sS1d[m][nt][e]] = dw1[1][nt][e]; }
                }
            }
        }
        __syncthreads();
        {
            const AdamK a = step_adam(lr, beta1, beta2, adam_eps, wd, b1t, b2t);
            adam_phase(lds + pl.oSTG, pl.stg_floats, P, P, P, nw, PAR, MM, VV, pl.mv_lds != 0, exp_avg, exp_avg_sq, a, tid, nthreads);
            if (tid == 0 && loss_hist) {
                float acc = 0.f;
                for (int w = 0; w < nw; ++w) acc += RED[w];
                loss_hist[kb] = acc * inv_B;
            }
        }
        __syncthreads();
    }
    for (int p = tid; p < P; p += nthreads) {
        params[p] = PAR[p];
        if (pl.mv_lds) { exp_avg[p] = MM[p]; exp_avg_sq[p] = VV[p]; }
    }
}

int cv_kit(const CvaeK &k) {
    const int a = k.d + k.c + 1, b = k.lat + k.c + 1;
    const int ki = ((a > b ? a : b) + 3) / 4;
    return ki <= 2 ? 2 : (ki <= 4 ? 4 : 8);
}

bool make_cv_plan(const CvaeK &k, int64_t batch, CvPlan *out) {
    if (k.enc.nh != 1 || k.d > 16 || k.d + k.c > 31 || k.lat > 8 || k.lat + k.c > 31 || k.enc.nout[0] > 32) return false;
    if (k.enc.nout[0] > 16 && cv_kit(k) > 4) return false;          // the policy measured for the flows
    if (batch < 1 || batch > 16 * kRcMaxWaves) return false;
    CvPlan p;
    std::memset(&p, 0, sizeof(p));
    p.W = (int)((batch + 15) / 16);
    p.P = k.enc.npn + k.dec.npn;
    p.stg_floats = p.P + kDump;
    p.tt_floats = (2 * (cv_kit(k) > 4 ? 2 : 1) + 6) * 16 * TS;
    for (int mv = 1; mv >= 0; --mv) {
        int f = 0;
        p.oPAR = f; f += p.P;
        p.oM = f; p.oV = f;
        if (mv) { p.oM = f; f += p.P; p.oV = f; f += p.P; }
        p.oSTG = f; f += p.W * p.stg_floats;
        p.oRED = f; f += kMaxWaves;
        p.oTT = f; f += p.W * p.tt_floats;
        f += 16 * 64 + 64;                      // the unguarded gathers of padding lanes stay inside the allocation
        p.total_floats = f;
        p.mv_lds = mv;
        if ((size_t)f * sizeof(float) <= kLdsMax) { *out = p; return true; }
    }
    return false;
}

}  // namespace

bool fits(const KShape &k, int64_t batch_size) {
    if (k.family == RNVP_FAMILY_VALU) return false;
    RcPlan rc;
    if (make_rc_plan(k, batch_size, &rc)) return true;
    return false;
}

namespace {

struct EpochArgs {
    float *params; const uint8_t *masks; const float *x, *c; const int64_t *perm; int64_t n, batch_size;
    float *loss_hist, *exp_avg, *exp_avg_sq; double lr, beta1, beta2, eps, wd; int64_t first_step;
};

template <int MT, int KIT, int ACT, int WMAX, int DT>
int launch_rc_w(hipStream_t st, const KShape &k, const RcPlan &p, const EpochArgs &a) {
    auto kern = k_fit_resident_rc<MT, KIT, ACT, WMAX, DT>;
    static std::atomic<uint64_t> attr_done{0};
    const int rc = allow_big_lds(reinterpret_cast<const void *>(kern), (int)kLdsMax, attr_done);
    if (rc) return rc;
    {
        KernelTimer timer(st, RNVP_PROFILE_TRAIN);
        hipLaunchKernelGGL(kern, dim3(1), dim3(64 * WMAX), (size_t)p.total_floats * sizeof(float), st, k, p, a.params, a.masks, a.x, a.c,
                           a.perm, a.n, a.batch_size, a.loss_hist, a.exp_avg, a.exp_avg_sq, a.lr, a.beta1, a.beta2, a.eps, a.wd,
                           std::pow(a.beta1, (double)a.first_step), std::pow(a.beta2, (double)a.first_step));
    }
    RNVP_HIP_TRY(hipGetLastError());
    return RNVP_OK;
}

template <int MT, int KIT, int DT>
int launch_rc_d(hipStream_t st, const KShape &k, const RcPlan &p, const EpochArgs &a) {
    const bool th = k.act == RNVP_ACT_TANH;
    if (p.W <= 4)       // one wave per SIMD: 512 registers
        return th ? launch_rc_w<MT, KIT, RNVP_ACT_TANH, 4, DT>(st, k, p, a) : launch_rc_w<MT, KIT, RNVP_ACT_RELU, 4, DT>(st, k, p, a);
    return th ? launch_rc_w<MT, KIT, RNVP_ACT_TANH, kRcMaxWaves, DT>(st, k, p, a)
              : launch_rc_w<MT, KIT, RNVP_ACT_RELU, kRcMaxWaves, DT>(st, k, p, a);
}

template <int MT, int KIT>
int launch_rc(hipStream_t st, const KShape &k, const RcPlan &p, const EpochArgs &a) {
    if (k.d <= 4) return launch_rc_d<MT, KIT, 1>(st, k, p, a);      // 2-d toy data, the reference's examples: one x slot
    return launch_rc_d<MT, KIT, 4>(st, k, p, a);
}

template <int NH, int KIT, int ACT, int WMAX, int DT>
int launch_deep_w(hipStream_t st, const KShape &k, const RcPlan &p, const EpochArgs &a) {
    auto kern = k_fit_resident_deep<NH, KIT, ACT, WMAX, DT>;
    static std::atomic<uint64_t> attr_done{0};
    const int rc = allow_big_lds(reinterpret_cast<const void *>(kern), (int)kLdsMax, attr_done);
    if (rc) return rc;
    {
        KernelTimer timer(st, RNVP_PROFILE_TRAIN);
        hipLaunchKernelGGL(kern, dim3(1), dim3(64 * WMAX), (size_t)p.total_floats * sizeof(float), st, k, p, a.params, a.masks, a.x, a.c,
                           a.perm, a.n, a.batch_size, a.loss_hist, a.exp_avg, a.exp_avg_sq, a.lr, a.beta1, a.beta2, a.eps, a.wd,
                           std::pow(a.beta1, (double)a.first_step), std::pow(a.beta2, (double)a.first_step));
    }
    RNVP_HIP_TRY(hipGetLastError());
    return RNVP_OK;
}

template <int NH, int KIT, int DT>
int launch_deep_d(hipStream_t st, const KShape &k, const RcPlan &p, const EpochArgs &a) {
    const bool th = k.act == RNVP_ACT_TANH;
    if (p.W <= 4)
        return th ? launch_deep_w<NH, KIT, RNVP_ACT_TANH, 4, DT>(st, k, p, a) : launch_deep_w<NH, KIT, RNVP_ACT_RELU, 4, DT>(st, k, p, a);
    return th ? launch_deep_w<NH, KIT, RNVP_ACT_TANH, kRcMaxWaves, DT>(st, k, p, a)
              : launch_deep_w<NH, KIT, RNVP_ACT_RELU, kRcMaxWaves, DT>(st, k, p, a);
}

template <int NH>
int launch_deep(hipStream_t st, const KShape &k, const RcPlan &p, const EpochArgs &a) {
    const int kit = rc_kit(k);
    const bool small_d = k.d <= 4;
    if (kit == 2) return small_d ? launch_deep_d<NH, 2, 1>(st, k, p, a) : launch_deep_d<NH, 2, 4>(st, k, p, a);
    if (kit == 4) return small_d ? launch_deep_d<NH, 4, 1>(st, k, p, a) : launch_deep_d<NH, 4, 4>(st, k, p, a);
    return small_d ? launch_deep_d<NH, 8, 1>(st, k, p, a) : launch_deep_d<NH, 8, 4>(st, k, p, a);
}

template <int MT>
int launch_rc_kit(hipStream_t st, const KShape &k, const RcPlan &p, const EpochArgs &a) {
    const int kit = rc_kit(k);
    if (kit == 2) return launch_rc<MT, 2>(st, k, p, a);
    if (kit == 4) return launch_rc<MT, 4>(st, k, p, a);
    return launch_rc<MT, 8>(st, k, p, a);
}

}  // namespace

int fit_epoch(hipStream_t st, const KShape &k, float *params, const uint8_t *masks, const float *x, const float *c,
              const int64_t *perm, int64_t n, int64_t batch_size, float *loss_hist, float *exp_avg, float *exp_avg_sq,
              double lr, double beta1, double beta2, double eps, double weight_decay, int64_t first_step) {
    if (n == 0) return RNVP_OK;
    RcPlan rcp;
    if (make_rc_plan(k, batch_size, &rcp)) {
        const EpochArgs a{params, masks, x, c, perm, n, batch_size, loss_hist, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay,
                          first_step};
        if (k.nh == 2) return launch_deep<2>(st, k, rcp, a);
        if (k.nh == 3) return launch_deep<3>(st, k, rcp, a);
        if (k.nout[0] <= 16) return launch_rc_kit<1>(st, k, rcp, a);
        return launch_rc_kit<2>(st, k, rcp, a);
    }
    return RNVP_EUNSUPPORTED;
}

namespace {

struct CvArgs {
    float *params; const float *x, *c; const int64_t *perm; const float *eps; int64_t n, batch_size; float klw;
    float *loss_hist, *exp_avg, *exp_avg_sq; double lr, beta1, beta2, eps_adam, wd; int64_t first_step;
};

template <int MT, int KIT, int ACT, int WMAX>
int launch_cv_w(hipStream_t st, const CvaeK &k, const CvPlan &p, const CvArgs &a) {
    auto kern = k_cvae_fit_resident<MT, KIT, ACT, WMAX>;
    static std::atomic<uint64_t> attr_done{0};
    const int rc = allow_big_lds(reinterpret_cast<const void *>(kern), (int)kLdsMax, attr_done);
    if (rc) return rc;
    {
        KernelTimer timer(st, RNVP_PROFILE_TRAIN);
        hipLaunchKernelGGL(kern, dim3(1), dim3(64 * WMAX), (size_t)p.total_floats * sizeof(float), st, k, p, a.params, a.x, a.c, a.perm,
                           a.eps, a.n, a.batch_size, a.klw, a.loss_hist, a.exp_avg, a.exp_avg_sq, a.lr, a.beta1, a.beta2, a.eps_adam,
                           a.wd, std::pow(a.beta1, (double)a.first_step), std::pow(a.beta2, (double)a.first_step));
    }
    RNVP_HIP_TRY(hipGetLastError());
    return RNVP_OK;
}

template <int MT, int KIT>
int launch_cv(hipStream_t st, const CvaeK &k, const CvPlan &p, const CvArgs &a) {
    const bool th = k.enc.act == RNVP_ACT_TANH;
    if (p.W <= 4)
        return th ? launch_cv_w<MT, KIT, RNVP_ACT_TANH, 4>(st, k, p, a) : launch_cv_w<MT, KIT, RNVP_ACT_RELU, 4>(st, k, p, a);
    return th ? launch_cv_w<MT, KIT, RNVP_ACT_TANH, kRcMaxWaves>(st, k, p, a) : launch_cv_w<MT, KIT, RNVP_ACT_RELU, kRcMaxWaves>(st, k, p, a);
}

template <int MT>
int launch_cv_kit(hipStream_t st, const CvaeK &k, const CvPlan &p, const CvArgs &a) {
    const int kit = cv_kit(k);
    if (kit == 2) return launch_cv<MT, 2>(st, k, p, a);
    if (kit == 4) return launch_cv<MT, 4>(st, k, p, a);
    return launch_cv<MT, 8>(st, k, p, a);
}

}  // namespace

bool cvae_fits(const CvaeK &k, int family, int64_t batch_size) {
    if (family == RNVP_FAMILY_VALU) return false;
    CvPlan p;
    return make_cv_plan(k, batch_size, &p);
}

int cvae_fit_epoch(hipStream_t st, const CvaeK &k, float *params, const float *x, const float *c, const int64_t *perm,
                   const float *eps, int64_t n, int64_t batch_size, float kl_weight, float *loss_hist, float *exp_avg,
                   float *exp_avg_sq, double lr, double beta1, double beta2, double adam_eps, double weight_decay, int64_t first_step) {
    if (n == 0) return RNVP_OK;
    CvPlan p;
    if (!make_cv_plan(k, batch_size, &p)) return RNVP_EUNSUPPORTED;
    const CvArgs a{params, x, c, perm, eps, n, batch_size, kl_weight, loss_hist, exp_avg, exp_avg_sq, lr, beta1, beta2, adam_eps,
                   weight_decay, first_step};
    if (k.enc.nout[0] <= 16) return launch_cv_kit<1>(st, k, p, a);
    return launch_cv_kit<2>(st, k, p, a);
}

}  // namespace resident
}  // namespace rnvp
