// rnvp_lmm64.hip -- the any-shape training kernel of rnvp_lmm.hip re-blocked: 64 rows per workgroup, weight gradients inside.
//
// k_lmm_train (rnvp_lmm.hip) gives one 16-row tile to a workgroup: every 16 rows re-read all weight fragments from L2 three
// times (forward, recompute, transposed) and dump 35 KB per row of weight-gradient operands for a second kernel.  Here
//   * a workgroup of 8 waves owns BLOCKS of 64 rows (4 row tiles); its LDS image is [feature group of 4][64 rows][4] (+ 4 dwords
//     of padding per group), so that a lane's B operands of four k-steps are ONE ds_read_b128 and an accumulator tile is ONE
//     ds_write_b128 in the layout the next Linear reads; a weight fragment (one 16-byte load per lane and 16 inputs) multiplies
//     four row tiles: 16 MFMAs per fragment load instead of 4;
//   * the WEIGHT gradients are contracted over the block's 64 rows by the same workgroup, straight from the image (the
//     pre-activation gradient replaces the activation it belongs to, in place): a unit = 16 outputs x 64 inputs (or 64 x 16 for a
//     Linear with few inputs), four accumulator tiles in registers;
//   * the backward runs LAYER-major over the workgroup's blocks -- for layer l and net s / t: for every block (reload its layer
//     input and output gradient: 64 B per row, saved by the forward / the previous layer) recompute, chain, accumulate -- so a
//     unit's accumulators live in registers across all blocks of the workgroup and are flushed ONCE per (workgroup, layer, net):
//     256 x P floats per step instead of 35 KB per row; k_lmm64_reduce adds the workgroups' partials in order (no float atomics:
//     bitwise reproducible) and scatters into the reference's flat order.
// Math: /root/reference/probaforms/models/realnvp.py:22-38 (the nets), :91-101 (the coupling), :246-250 (the loss); the
// hand-derived backward is rnvp_lmm.hip's.
#include "rnvp_common.h"
#include "rnvp_generic_net.h"
#include "rnvp_lmm.h"

namespace rnvp {
namespace lmm {
namespace {

using f4 = __attribute__((ext_vector_type(4))) float;
constexpr int kW8 = 8;                 // waves per workgroup
constexpr int BR = 64;                 // rows per block
constexpr int FS = 260;                // dwords per feature group: 64 rows x 4 features + 4 (bank spread for the row-contraction reads)
constexpr int kSlots = 4;              // weight-gradient units a wave keeps in registers (per net)
constexpr int kUnitFloats = 4 * 256 + 64;   // one unit's partial: four accumulator tiles + its bias sums
constexpr int kSlackFg = 16;           // zeroed feature groups past the last region (padded k-groups / strided unit reads end here)

// RNVP_STAMP: diagnostic build that accumulates cycle-counter deltas per kind of work and printf()s them for workgroup 0 (read
// the SHARES, not the absolute time: the stamps serialise the wave)
struct Stamps64 { unsigned long long t0, gemm, epi, bar, wgrad, other, load, elem, fwdx; };
#ifdef RNVP_STAMP
#define STAMP64(field) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t1__ = __builtin_readcyclecounter(); \
                            __builtin_amdgcn_sched_barrier(0); stp.field += t1__ - stp.t0; stp.t0 = t1__; } while (0)
#else
#define STAMP64(field) do { } while (0)
#endif
#define SYNC64() do { STAMP64(other); __syncthreads(); STAMP64(bar); } while (0)

__device__ __forceinline__ f4 mfma16(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

struct G64 {
    int nlin, nnets;
    int nin[kMaxLin], nout[kMaxLin];
    int MT[kMaxLin], KG[kMaxLin];          // forward: out tiles (16), groups of 16 inputs
    int MTt[kMaxLin], KGt[kMaxLin];        // transposed (input gradient): in tiles, groups of 16 outputs
    int offF[kMaxLin], offT[kMaxLin];      // float offsets of the fragment blocks inside one net's packed image
    int net_floats;
    int ukind[kMaxLin];                    // weight-gradient units of Linear k: 0 = 16 outputs x 64 inputs, 1 = 64 outputs x 16 inputs,
                                           // 2 = 16 x 16 (a Linear of at most 16 outputs: eight small units instead of two large ones)
    int unA[kMaxLin], unB[kMaxLin], uoff[kMaxLin], nunits;
    int fgA[kMaxLin];                      // feature-group offset of hidden activation k in the image
    int fgXC, fgT, fgS, fgGY, fgGYB, fgGO, fg_total, xc_fgs, d_fgs;
    int h_fgs;                             // feature groups of all hidden regions (contiguous from fgA[0]): one net's saved activations
    size_t lds_bytes;
};

// ---- weight packing: forward fragment of out tile m, input group G, lane (q, i): W[16m + i][16G + 4q + e], e = 0..3 (one
// 16-byte load; k-step e of the group stands for input 16G + 4q + e on both operands); transposed: W[16G + 4q + e][16m + i]
// (Linear 0: x columns only).  The layer's mask is folded into Linear 0's x columns.
__global__ void __launch_bounds__(256)
k_lmm64_pack(KShape s, G64 g, const float *__restrict__ params, const uint8_t *__restrict__ masks, float *__restrict__ packed) {
    const int64_t total = (int64_t)g.nnets * g.net_floats;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int ln = (int)(t / g.net_floats), o = (int)(t - (int64_t)ln * g.net_floats);
        const int l = ln >> 1;
        const float *pn = params + (size_t)ln * s.npn;
        float v = 0.f;
        for (int k = 0; k < g.nlin; ++k) {
            const int nin = g.nin[k], nout = g.nout[k];
            const float *W = pn + s.woff[k];
            if (o >= g.offF[k] && o < g.offF[k] + g.MT[k] * g.KG[k] * 256) {
                const int oo = o - g.offF[k], e = oo & 3, lane = (oo >> 2) & 63, mg = oo >> 8;
                const int m = mg / g.KG[k], G = mg - m * g.KG[k];
                const int row = 16 * m + (lane & 15), col = 16 * G + 4 * (lane >> 4) + e;
                if (row < nout && col < nin) v = W[row * nin + col] * ((k == 0 && col < s.d) ? (float)masks[l * s.d + col] : 1.f);
                break;
            }
            if (o >= g.offT[k] && o < g.offT[k] + g.MTt[k] * g.KGt[k] * 256) {
                const int oo = o - g.offT[k], e = oo & 3, lane = (oo >> 2) & 63, mg = oo >> 8;
                const int m = mg / g.KGt[k], G = mg - m * g.KGt[k];
                const int out = 16 * G + 4 * (lane >> 4) + e, in = 16 * m + (lane & 15);
                const int nin_eff = k == 0 ? s.d : nin;
                if (out < nout && in < nin_eff) v = W[out * nin + in] * (k == 0 ? (float)masks[l * s.d + in] : 1.f);
                break;
            }
        }
        packed[t] = v;
    }
}

// The first two fragment groups of a wave's first unit of a Linear, requested BEFORE the barrier that ends the previous phase
// (an L2 round trip is ~1 000 cycles; after a barrier both waves of a SIMD would wait for it together).
struct Pre { f4 a0, a1; };
struct Lin { const float *frag; int MT, KG; };                   // fragments of one Linear (forward or transposed)
__device__ __forceinline__ int units_per_tile(int MT) { return MT >= 8 ? 1 : (MT >= 4 ? 2 : 4); }
__device__ __forceinline__ Pre prefetch(const Lin &L, int lane, int wave) {
    Pre p;
    p.a0 = p.a1 = f4{0.f, 0.f, 0.f, 0.f};
    const int per = units_per_tile(L.MT);
    if (L.frag && wave < L.MT * per) {
        const float *pa = L.frag + (size_t)(wave / per) * L.KG * 256 + lane * 4;
        p.a0 = *reinterpret_cast<const f4 *>(pa);
        if (L.KG > 1) p.a1 = *reinterpret_cast<const f4 *>(pa + 256);
    }
    return p;
}

// acc[t] (t < RG) = W tile m . in^T for row tiles rt0 .. rt0 + RG - 1: `ftile` = the tile's fragments [group][lane][4], `inb` the
// LDS region of the input.  Fragments (L2) are requested two groups ahead, the LDS operands one; the first two fragment groups
// come from `pre` when use_pre.
template <int RG>
__device__ __forceinline__ void gemm_acc(const float *__restrict__ ftile, int KG, const float *inb, int rt0, int lane, f4 (&acc)[RG],
                                         const Pre &pre, bool use_pre) {
    const int q = lane >> 4, i = lane & 15;
    const float *pa = ftile + lane * 4;                          // + G * 256
    const float *pb = inb + q * FS + (16 * rt0 + i) * 4;         // + G * 4 * FS + t * 64
#pragma unroll
    for (int t = 0; t < RG; ++t) acc[t] = f4{0.f, 0.f, 0.f, 0.f};
    f4 a[3], b[2][RG];
    auto fetch_b = [&](f4 (&bb)[RG], int G) {
#pragma unroll
        for (int t = 0; t < RG; ++t) bb[t] = *reinterpret_cast<const f4 *>(pb + G * 4 * FS + t * 64);
    };
    a[0] = pre.a0; a[1] = pre.a1;
    if (!use_pre) {
        a[0] = *reinterpret_cast<const f4 *>(pa);
        a[1] = *reinterpret_cast<const f4 *>(pa + min(1, KG - 1) * 256);
    }
    fetch_b(b[0], 0);
#pragma unroll 1
    for (int G0 = 0; G0 < KG; G0 += 6) {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int G = G0 + j;
            if (G >= KG) break;
            // unconditional (the last groups re-request the last one): a load under a branch makes the compiler wait for ALL
            // outstanding loads at the join, i.e. for the groups just requested
            a[(j + 2) % 3] = *reinterpret_cast<const f4 *>(pa + min(G + 2, KG - 1) * 256);
            fetch_b(b[(j + 1) % 2], min(G + 1, KG - 1));
            __builtin_amdgcn_sched_barrier(0);                    // or the scheduler sinks the requests down to their uses
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int t = 0; t < RG; ++t) acc[t] = mfma16(a[j % 3][e], b[j % 2][t][e], acc[t]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

enum { EP_FWD_ACT = 0, EP_FWD_LIN, EP_GRAD, EP_GRAD0 };

// one Linear for the block: the (out tile, row-tile group) units go round-robin to the waves.  Epilogues (`ep`, wave-uniform):
//   EP_FWD_ACT  out = act(acc + bias)           EP_FWD_LIN  out = acc + bias
//   EP_GRAD     out = acc * act'(out)  (in place: `out` holds the activation, and receives the pre-activation gradient)
//   EP_GRAD0    out += acc             (input gradient of Linear 0, added to the layer's running d loss / d x)
// Outputs past nvalid are written as 0.  `pre` = prefetch(this Linear); returns prefetch(next), requested between the wave's last
// product and its epilogue.  One instantiation per RG serves every call: the kernel's code has to stay inside the instruction cache.
template <int RG>
__device__ __forceinline__ Pre gemm_units(const Lin &L, const float *inb, float *outb, const float *__restrict__ bias, int nvalid,
                                          int ep, int act, int lane, int wave, const Pre &pre, const Lin &next, float *__restrict__ gsave,
                                          Stamps64 &stp) {
    constexpr int per = 4 / RG;
    const int q = lane >> 4, i = lane & 15;
    Pre nx;
    bool requested = false;
    STAMP64(other);
#pragma unroll 1
    for (int u = wave; u < L.MT * per; u += kW8) {
        const int m = u / per, rt0 = (u - m * per) * RG;
        const int o0 = 16 * m + 4 * q;
        f4 bv = f4{0.f, 0.f, 0.f, 0.f};
        if (ep <= EP_FWD_LIN) {                                   // requested ahead of the products
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[e] = o0 + e < nvalid ? bias[o0 + e] : 0.f;
        }
        f4 acc[RG];
        gemm_acc<RG>(L.frag + (size_t)m * L.KG * 256, L.KG, inb, rt0, lane, acc, pre, u == wave);
        if (u + kW8 >= L.MT * per) { nx = prefetch(next, lane, wave); requested = true; }
        STAMP64(gemm);
#pragma unroll
        for (int t = 0; t < RG; ++t) {
            f4 *po = reinterpret_cast<f4 *>(outb + (4 * m + q) * FS + (16 * (rt0 + t) + i) * 4);
            f4 v = acc[t];
            if (ep == EP_FWD_ACT) {
                v += bv;
                if (act == RNVP_ACT_TANH) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = act_fwd(v[e], RNVP_ACT_TANH);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                if (16 * m + 16 > nvalid) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (o0 + e >= nvalid) v[e] = 0.f;
                }
                // the backward reads the activation back instead of recomputing the net: [feature group][64 rows][4], as the region
                if (gsave) *reinterpret_cast<f4 *>(gsave + ((size_t)(4 * m + q) * BR + 16 * (rt0 + t) + i) * 4) = v;
            } else if (ep == EP_FWD_LIN) {
                v += bv;
            } else {
                const f4 cur = *po;
                if (ep == EP_GRAD0) v += cur;
                else if (act == RNVP_ACT_TANH) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= fmaf(-cur[e], cur[e], 1.f);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = cur[e] > 0.f ? v[e] : 0.f;
                }
            }
            if (16 * m + 16 > nvalid) {                           // a ragged last tile (wave-uniform): outputs past nvalid are 0
                const f4 keep = ep == EP_GRAD0 ? *po : f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) if (o0 + e >= nvalid) v[e] = keep[e];
            }
            *po = v;
        }
        STAMP64(epi);
    }
    if (!requested) nx = prefetch(next, lane, wave);
    return nx;
}

__device__ __forceinline__ Pre gemm(const Lin &L, const float *inb, float *outb, const float *__restrict__ bias, int nvalid, int ep,
                                    int act, int lane, int wave, const Pre &pre, const Lin &next, float *__restrict__ gsave, Stamps64 &stp) {
    if (L.MT >= 8) return gemm_units<4>(L, inb, outb, bias, nvalid, ep, act, lane, wave, pre, next, gsave, stp);
    if (L.MT >= 4) return gemm_units<2>(L, inb, outb, bias, nvalid, ep, act, lane, wave, pre, next, gsave, stp);
    return gemm_units<1>(L, inb, outb, bias, nvalid, ep, act, lane, wave, pre, next, gsave, stp);
}

__device__ __forceinline__ Lin lin_fwd(const float *pk, const G64 &g, int k) { return Lin{pk + g.offF[k], g.MT[k], g.KG[k]}; }
__device__ __forceinline__ Lin lin_t(const float *pk, const G64 &g, int k) { return Lin{pk + g.offT[k], g.MTt[k], g.KGt[k]}; }
__device__ __forceinline__ Lin lin_none() { return Lin{nullptr, 0, 0}; }

// weight-gradient unit, accumulated over the block's 64 rows (k-step ks of lane group q stands for row 4 ks + q):
//   KIND 0: outputs 16 ua .. + 15 (lane i: one value per k-step) x inputs 64 ub + 4 i + e (one b128: four tiles e)
//   KIND 1: outputs 64 ua + 4 i + e (one b128: four tiles e) x inputs 16 ub .. + 15
//   KIND 2: outputs 16 ua .. + 15 x inputs 16 ub .. + 15 (one dword each, one tile)
// bs collects the sums over rows of the A operand (the bias gradient, wanted from the unit with ub == 0).
template <int KIND>
__device__ __forceinline__ void wgrad_unit(const float *ldsA, const float *ldsB, int ua, int ub, int lane, f4 (&acc)[4], f4 &bs) {
    const int q = lane >> 4, i = lane & 15;
    if (KIND == 2) {
        const float *pa = ldsA + (4 * ua + (i >> 2)) * FS + q * 4 + (i & 3), *pb = ldsB + (4 * ub + (i >> 2)) * FS + q * 4 + (i & 3);
        float a[16], b[16];
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) { a[ks] = pa[ks * 16]; b[ks] = pb[ks * 16]; }
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) { bs[0] += a[ks]; acc[0] = mfma16(a[ks], b[ks], acc[0]); }
        return;
    }
    // the four-tile operand: one b128 per k-step at [16 u + i][4 ks + q]; the one-tile operand: one dword at feature 16 u + i
    const float *p4 = (KIND == 0 ? ldsB + (16 * ub + i) * FS : ldsA + (16 * ua + i) * FS) + q * 4;
    const float *p1 = (KIND == 0 ? ldsA + (4 * ua + (i >> 2)) * FS : ldsB + (4 * ub + (i >> 2)) * FS) + q * 4 + (i & 3);
    f4 w0 = *reinterpret_cast<const f4 *>(p4), w1;
    float n0 = p1[0], n1;
    auto mul = [&](const f4 &w, float nv) {
        if (KIND == 0) {
            bs[0] += nv;
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = mfma16(nv, w[e], acc[e]);
        } else {
            bs += w;
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = mfma16(w[e], nv, acc[e]);
        }
    };
#pragma unroll
    for (int ks = 0; ks < 14; ks += 2) {                           // no load under a branch (see gemm_acc): the last pair is peeled
        w1 = *reinterpret_cast<const f4 *>(p4 + (ks + 1) * 16); n1 = p1[(ks + 1) * 16];
        __builtin_amdgcn_sched_barrier(0);
        mul(w0, n0);
        __builtin_amdgcn_sched_barrier(0);
        w0 = *reinterpret_cast<const f4 *>(p4 + (ks + 2) * 16); n0 = p1[(ks + 2) * 16];
        __builtin_amdgcn_sched_barrier(0);
        mul(w1, n1);
        __builtin_amdgcn_sched_barrier(0);
    }
    w1 = *reinterpret_cast<const f4 *>(p4 + 15 * 16); n1 = p1[15 * 16];
    __builtin_amdgcn_sched_barrier(0);
    mul(w0, n0);
    mul(w1, n1);
}

// block-wide copies between a region of the image and its [feature group][64 rows][4] copy in global memory
__device__ __forceinline__ void region_load(float *reg, const float *__restrict__ src, int fgs, int tid) {
    for (int e = tid; e < fgs * BR; e += 64 * kW8)
        *reinterpret_cast<f4 *>(reg + (e >> 6) * FS + (e & 63) * 4) = *reinterpret_cast<const f4 *>(src + (size_t)e * 4);
}
__device__ __forceinline__ void region_store(const float *reg, float *__restrict__ dst, int fgs, int tid) {
    for (int e = tid; e < fgs * BR; e += 64 * kW8)
        *reinterpret_cast<f4 *>(dst + (size_t)e * 4) = *reinterpret_cast<const f4 *>(reg + (e >> 6) * FS + (e & 63) * 4);
}

// the same copy global -> image as LDS-DMA: one wave instruction moves a feature group (64 lanes x 16 bytes to a wave-uniform LDS
// base + lane * 16), no registers, completion counted by vmcnt
__device__ __forceinline__ void region_dma(float *reg, const float *__restrict__ src, int fgs, int lane, int wave) {
    for (int fg = wave; fg < fgs; fg += kW8)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) uint32_t *)(src + ((size_t)fg * BR + lane) * 4),
                                         (__attribute__((address_space(3))) uint32_t *)(reg + fg * FS), 16, 0, 0);
}

// one net forward for the block, every hidden activation kept in its own region (and saved to `asave`, the net's
// [hidden feature groups][64 rows][4] block, for the backward); the last Linear writes `out`.
// `pre` = prefetch(Linear 0 of this net); returns prefetch(`after`).
__device__ __forceinline__ Pre net_fwd64(const float *__restrict__ pk, const float *__restrict__ pn, const KShape &s, const G64 &g,
                                         float *lds, float *out, float *__restrict__ asave, int lane, int wave, Pre pre, const Lin &after,
                                         Stamps64 &stp) {
    const float *cur = lds + g.fgXC * FS;
    const int nl = g.nlin;
#pragma unroll 1
    for (int k = 0; k < nl; ++k) {
        const bool last = k == g.nlin - 1;
        float *ob = last ? out : lds + g.fgA[k] * FS;
        pre = gemm(lin_fwd(pk, g, k), cur, ob, pn + s.boff[k], g.nout[k], last ? EP_FWD_LIN : EP_FWD_ACT, s.act, lane, wave, pre,
                   k + 1 < nl ? lin_fwd(pk, g, k + 1) : after, last ? nullptr : asave + (size_t)(g.fgA[k] - g.fgA[0]) * BR * 4, stp);
        SYNC64();
        cur = ob;
    }
    return pre;
}

__global__ void __launch_bounds__(64 * kW8)
k_lmm_train64(KShape s, G64 g, const float *__restrict__ packed, const float *__restrict__ params,
              const uint8_t *__restrict__ masks, const float *__restrict__ x, const float *__restrict__ c,
              const int64_t *__restrict__ row_index, int64_t n, float inv_B, Seeds sd, float *__restrict__ xsave,
              float *__restrict__ gysave, float *__restrict__ gybsave, float *__restrict__ asave, float *__restrict__ ssave,
              float *__restrict__ gpart, float *__restrict__ losspart, int first_chunk) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), d = s.d, cd = s.c;
    const int row = lane;                                   // elementwise passes: thread = (row, feature group wave, wave + 8, ..)
    float *XC = lds + g.fgXC * FS, *T = lds + g.fgT * FS, *S = lds + g.fgS * FS, *GY = lds + g.fgGY * FS;
    float *GYB = lds + g.fgGYB * FS, *GO = lds + g.fgGO * FS;
    float *RED = lds + g.fg_total * FS;                     // 2 x 8 x 64 floats of cross-wave reduction scratch
    for (int e = tid; e < g.fg_total * FS; e += 64 * kW8) lds[e] = 0.f;      // padded k-groups read past a region: must be finite
    __syncthreads();
    const int64_t nblocks = (n + BR - 1) / BR;
    const float prior_c = 0.5f * (float)d * kLog2Pi;
    const float *__restrict__ gz = sd.gz;
    const size_t xs_blk = (size_t)s.L * g.xc_fgs * BR * 4, gy_blk = (size_t)g.d_fgs * BR * 4, as_net = (size_t)g.h_fgs * BR * 4;
    float wave_sum = 0.f;
    Stamps64 stp = {};
#ifdef RNVP_STAMP
    const unsigned long long tk0 = __builtin_readcyclecounter();
    stp.t0 = tk0;
#endif

    // ---- forward over the workgroup's blocks: saves every layer's input image and the seed of the backward ----
#pragma unroll 1
    for (int64_t b = blockIdx.x; b < nblocks; b += gridDim.x) {
        const int64_t base = b * BR;
        STAMP64(other);
        for (int e = tid; e < BR * d; e += 64 * kW8) {
            const int rr = e / d, j = e - rr * d;
            const int64_t r = base + rr;
            XC[(j >> 2) * FS + rr * 4 + (j & 3)] = r < n ? x[(row_index ? row_index[r] : r) * d + j] : 0.f;
        }
        for (int e = tid; e < BR * cd; e += 64 * kW8) {
            const int rr = e / cd, j = e - rr * cd, jj = d + j;
            const int64_t r = base + rr;
            XC[(jj >> 2) * FS + rr * 4 + (jj & 3)] = r < n ? c[(row_index ? row_index[r] : r) * cd + j] : 0.f;
        }
        Pre pre = prefetch(lin_fwd(packed, g, 0), lane, wave);
        STAMP64(fwdx);
        SYNC64();
        float ld = 0.f;
#pragma unroll 1
        for (int ln = 0; ln < 2 * s.L; ++ln) {                    // (layer, net): t then s
            const int l = ln >> 1, net = ln & 1;
            if (net == 0) region_store(XC, xsave + (size_t)b * xs_blk + (size_t)l * g.xc_fgs * BR * 4, g.xc_fgs, tid);
            pre = net_fwd64(packed + (size_t)ln * g.net_floats, params + (size_t)ln * s.npn, s, g, lds, net ? S : T,
                            asave + ((size_t)b * g.nnets + ln) * as_net, lane, wave, pre,
                            ln + 1 < 2 * s.L ? lin_fwd(packed + (size_t)(ln + 1) * g.net_floats, g, 0) : lin_none(), stp);
            if (net == 0) continue;
            region_store(S, ssave + ((size_t)b * s.L + l) * gy_blk, g.d_fgs, tid);
            const uint8_t *m = masks + l * d;
            for (int fg = wave; fg < g.d_fgs; fg += kW8) {       // the coupling, realnvp.py:91-101
                f4 *px = reinterpret_cast<f4 *>(XC + fg * FS + row * 4);
                f4 xv = *px;
                const f4 sv = *reinterpret_cast<const f4 *>(S + fg * FS + row * 4), tv = *reinterpret_cast<const f4 *>(T + fg * FS + row * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int j = 4 * fg + e;
                    if (j < d && !m[j]) { xv[e] = fmaf(xv[e], expf(sv[e]), tv[e]); ld += sv[e]; }
                }
                *px = xv;
            }
            SYNC64();
        }
        {   // loss terms of the block's rows and the seed d loss / d z
            const int64_t r = base + row;
            const bool valid = r < n;
            float ss = 0.f;
            for (int fg = wave; fg < g.d_fgs; fg += kW8) {
                const f4 zv = *reinterpret_cast<const f4 *>(XC + fg * FS + row * 4);
                f4 gv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int j = 4 * fg + e;
                    if (j < d) ss = fmaf(zv[e], zv[e], ss);
                    gv[e] = (valid && j < d) ? (gz ? gz[r * d + j] : zv[e] * inv_B) : 0.f;
                }
                *reinterpret_cast<f4 *>(gysave + (size_t)b * gy_blk + ((size_t)fg * BR + row) * 4) = gv;
            }
            RED[wave * 64 + row] = ld; RED[(kW8 + wave) * 64 + row] = ss;
            SYNC64();
            if (wave == 0) {
                float lds_ = 0.f, sss = 0.f;
#pragma unroll
                for (int w = 0; w < kW8; ++w) { lds_ += RED[w * 64 + row]; sss += RED[(kW8 + w) * 64 + row]; }
                float v = valid ? (gz ? lds_ : lds_ + (-0.5f * sss - prior_c)) : 0.f;
                for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
                wave_sum += v;
            }
            SYNC64();
        }
    }
    if (tid == 0) losspart[blockIdx.x] = first_chunk ? wave_sum : losspart[blockIdx.x] + wave_sum;      // chunks run in order on one stream

    // ---- backward: layer-major over the blocks; a wave's weight-gradient units stay in registers across its blocks.  A visit
    // (layer, net, block) reloads the block's layer input, d loss / d (layer output), the s net's output and the net's hidden
    // activations; the activations arrive by LDS-DMA, each region requested as soon as the previous visit is done with it ----
    float *H0 = lds + g.fgA[0] * FS;
    if (blockIdx.x < nblocks) region_dma(H0, asave + ((size_t)blockIdx.x * g.nnets + 2 * s.L - 1) * as_net, g.h_fgs, lane, wave);
#pragma unroll 1
    for (int ln = 2 * s.L - 1; ln >= 0; --ln) {                  // (layer, net): s first -- t only adds to what s leaves in GYB
        const int l = ln >> 1, net = ln & 1;
        const uint8_t *m = masks + l * d;
        const float *pkn = packed + (size_t)ln * g.net_floats;
        f4 acc[kSlots][4], bs[kSlots];
#pragma unroll
        for (int sl = 0; sl < kSlots; ++sl) {
            bs[sl] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[sl][e] = f4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll 1
        for (int64_t b = blockIdx.x; b < nblocks; b += gridDim.x) {
            const int64_t r = b * BR + row;
            const bool valid = r < n;
            // the visit after this one: the next block of the same (layer, net), else the first block of the next net down
            STAMP64(other);
            const bool more = b + gridDim.x < nblocks;
            const float *anext = (more || ln > 0) ? asave + ((size_t)(more ? b + gridDim.x : blockIdx.x) * g.nnets + (more ? ln : ln - 1)) * as_net
                                                  : nullptr;
            region_load(XC, xsave + (size_t)b * xs_blk + (size_t)l * g.xc_fgs * BR * 4, g.xc_fgs, tid);
            region_load(GY, gysave + (size_t)b * gy_blk, g.d_fgs, tid);
            if (net == 0) region_load(GYB, gybsave + (size_t)b * gy_blk, g.d_fgs, tid);
            else region_load(S, ssave + ((size_t)b * s.L + l) * gy_blk, g.d_fgs, tid);
            Pre pre = prefetch(lin_t(pkn, g, g.nlin - 1), lane, wave);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the DMA'd activations of this visit
            STAMP64(load);
            SYNC64();
            const float gld = valid ? (sd.gld ? sd.gld[r] : -inv_B) : 0.f;
            for (int fg = wave; fg < g.d_fgs; fg += kW8) {
                const f4 gy = *reinterpret_cast<const f4 *>(GY + fg * FS + row * 4);
                f4 go, gyb;
                if (net) {
                    const f4 xv = *reinterpret_cast<const f4 *>(XC + fg * FS + row * 4), sv = *reinterpret_cast<const f4 *>(S + fg * FS + row * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int j = 4 * fg + e;
                        const bool tr = j < d && !m[j];             // a transformed feature
                        const float es = expf(sv[e]);
                        go[e] = tr ? fmaf(gy[e] * xv[e], es, gld) : 0.f;      // d / d s: (1-m)(gy x e^s + gld)
                        gyb[e] = j < d ? (tr ? gy[e] * es : gy[e]) : 0.f;     // the direct part of d loss / d x
                    }
                    *reinterpret_cast<f4 *>(GYB + fg * FS + row * 4) = gyb;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const int j = 4 * fg + e; go[e] = (j < d && !m[j]) ? gy[e] : 0.f; }     // d / d t: (1-m) gy
                }
                *reinterpret_cast<f4 *>(GO + fg * FS + row * 4) = go;
            }
            STAMP64(elem);
            SYNC64();
#pragma unroll 1
            for (int k = g.nlin - 1; k >= 0; --k) {
                const float *ga = k == g.nlin - 1 ? GO : lds + g.fgA[k] * FS;              // pre-activation gradient of Linear k
                float *in = k == 0 ? XC : lds + g.fgA[k - 1] * FS;                         // its input
                STAMP64(other);
#pragma unroll
                for (int sl = 0; sl < kSlots; ++sl) {
                    const int ul = kW8 * sl + wave - g.uoff[k];
                    if (ul >= 0 && ul < g.unA[k] * g.unB[k]) {
                        const int ua = ul / g.unB[k], ub = ul - ua * g.unB[k];
                        if (g.ukind[k] == 0) wgrad_unit<0>(ga, in, ua, ub, lane, acc[sl], bs[sl]);
                        else if (g.ukind[k] == 1) wgrad_unit<1>(ga, in, ua, ub, lane, acc[sl], bs[sl]);
                        else wgrad_unit<2>(ga, in, ua, ub, lane, acc[sl], bs[sl]);
                    }
                }
                STAMP64(wgrad);
                if (k) SYNC64();              // Linear 0's input gradient goes to GYB, not into its input: it runs beside the units
                pre = gemm(lin_t(pkn, g, k), ga, k ? in : GYB, nullptr, k ? g.nin[k] : d, k ? EP_GRAD : EP_GRAD0, s.act, lane, wave, pre,
                           k ? lin_t(pkn, g, k - 1) : lin_none(), nullptr, stp);
                SYNC64();
                // hidden region k is dead from here on: request the next visit's copy of it
                if (k < g.nlin - 1 && anext)
                    region_dma(lds + g.fgA[k] * FS, anext + (size_t)(g.fgA[k] - g.fgA[0]) * BR * 4, (g.nout[k] + 15) / 16 * 4, lane, wave);
            }
            region_store(GYB, (net ? gybsave : gysave) + (size_t)b * gy_blk, g.d_fgs, tid);
            if (net == 0 && l == 0 && sd.gx) {                      // rnvp_backward: d loss / d x of the batch rows
                for (int fg = wave; fg < g.d_fgs; fg += kW8) {
                    const f4 gv = *reinterpret_cast<const f4 *>(GYB + fg * FS + row * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const int j = 4 * fg + e; if (valid && j < d) sd.gx[r * d + j] = gv[e]; }
                }
            }
            SYNC64();
        }
        // flush this (layer, net)'s units: [workgroup][layer, net][unit][4 tiles x 64 lanes x 4 | 64 bias sums]
        float *dst = gpart + (((size_t)blockIdx.x * g.nnets + ln) * g.nunits) * kUnitFloats;
#pragma unroll
        for (int sl = 0; sl < kSlots; ++sl) {
            const int u = kW8 * sl + wave;
            if (u < g.nunits) {
                float *du = dst + (size_t)u * kUnitFloats;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    f4 *o = reinterpret_cast<f4 *>(du + e * 256 + lane * 4);
                    *o = first_chunk ? acc[sl][e] : *o + acc[sl][e];             // row chunks of one call, in order: deterministic
                }
                f4 bv = bs[sl];
#pragma unroll
                for (int e = 0; e < 4; ++e) { bv[e] += __shfl_xor(bv[e], 16); bv[e] += __shfl_xor(bv[e], 32); }
                if (lane < 16) {
                    f4 *o = reinterpret_cast<f4 *>(du + 1024 + lane * 4);
                    *o = first_chunk ? bv : *o + bv;
                }
            }
        }
    }
#ifdef RNVP_STAMP
    STAMP64(other);
    if (blockIdx.x == 0 && lane == 0)
        printf("STAMP64 wave %d total %llu gemm %llu epilogue %llu barrier %llu wgrad %llu visit-loads %llu elementwise %llu fwd-rows %llu other %llu\n",
               wave, __builtin_readcyclecounter() - tk0, stp.gemm, stp.epi, stp.bar, stp.wgrad, stp.load, stp.elem, stp.fwdx, stp.other);
#endif
}

// flat reference-order gradient: thread = one float of one (layer, net)'s unit partials; sums the workgroups in index order and
// scatters (every parameter has exactly one source slot); loss = -(sum of the workgroups' partials) * inv_B
__global__ void __launch_bounds__(256)
k_lmm64_reduce(KShape s, G64 g, const float *__restrict__ gpart, int G, const uint8_t *__restrict__ masks,
               const float *__restrict__ losspart, float inv_B, float *__restrict__ grad, float *loss) {
    const size_t per_net = (size_t)g.nunits * kUnitFloats, total = (size_t)g.nnets * per_net;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) {
        if (loss && blockIdx.x == gridDim.x - 1 && threadIdx.x >= 192) {
            const int lane = threadIdx.x - 192;
            float a = 0.f;
            for (int w = lane; w < G; w += 64) a += losspart[w];
            for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off);
            if (lane == 0) loss[0] = -a * inv_B;
        }
        return;
    }
    const int ln = (int)(t / per_net), rest = (int)(t - (size_t)ln * per_net);
    const int u = rest / kUnitFloats, j = rest - u * kUnitFloats;
    if (u < g.uoff[0]) return;                                  // unit numbers left free (make_g64)
    int k = 0;
    while (k + 1 < g.nlin && u >= g.uoff[k + 1]) ++k;
    const int ul = u - g.uoff[k], ua = ul / g.unB[k], ub = ul - ua * g.unB[k];
    const int nin = g.nin[k], nout = g.nout[k];
    int out, in;
    if (j < 1024) {
        const int tile = j >> 8, lane = (j >> 2) & 63, e = j & 3, q = lane >> 4, i = lane & 15;
        if (g.ukind[k] == 0) { out = 16 * ua + 4 * q + e; in = 64 * ub + 4 * i + tile; }
        else if (g.ukind[k] == 1) { out = 64 * ua + 4 * (4 * q + e) + tile; in = 16 * ub + i; }
        else { if (tile) return; out = 16 * ua + 4 * q + e; in = 16 * ub + i; }
        if (out >= nout || in >= nin) return;
    } else {
        if (ub != 0) return;
        const int jj = j - 1024;
        if (g.ukind[k] != 1) { if (jj & 3) return; out = 16 * ua + (jj >> 2); }      // lane i wrote (sum, 0, 0, 0)
        else out = 64 * ua + jj;                                                       // lane i wrote outputs 64 ua + 4 i + e
        if (out >= nout) return;
        in = nin;
    }
    const float *src = gpart + (size_t)ln * per_net + rest;
    const size_t stride = total;
    float a = 0.f;
    int w = 0;
    for (; w + 8 <= G; w += 8) {                                   // eight requests in flight, added in index order
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = src[(size_t)(w + j) * stride];
#pragma unroll
        for (int j = 0; j < 8; ++j) a += v[j];
    }
    for (; w < G; ++w) a += src[(size_t)w * stride];
    float scale = 1.f;
    if (k == 0 && in < s.d) scale = (float)masks[(ln >> 1) * s.d + in];             // W'[:, j] = W[:, j] * mask_j
    const size_t p = (size_t)ln * s.npn + (in < nin ? (size_t)s.woff[k] + (size_t)out * nin + in : (size_t)s.boff[k] + out);
    grad[p] = a * scale;
}

G64 make_g64(const KShape &k) {
    G64 g;
    std::memset(&g, 0, sizeof(g));
    g.nlin = k.nh + 1;
    g.nnets = 2 * k.L;
    int oW = 0, u = 0;
    // Linear 0's input gradient (one tile for d <= 16: four row-tile units, waves 0-3) runs beside its weight-gradient units: when
    // those are at most four they take waves 4-7, i.e. unit numbers 4 .. 7 (0 .. 3 stay free)
    {
        const int nin0 = k.nin[0], nout0 = k.nout[0];
        const int n0 = (nin0 <= 32 && nout0 > nin0) ? ((nout0 + 63) / 64) * ((nin0 + 15) / 16) : ((nout0 + 15) / 16) * ((nin0 + 63) / 64);
        if (k.d <= 16 && n0 <= 4) u = 4;
    }
    for (int i = 0; i < g.nlin; ++i) {
        g.nin[i] = k.nin[i]; g.nout[i] = k.nout[i];
        g.MT[i] = (k.nout[i] + 15) / 16; g.KG[i] = (k.nin[i] + 15) / 16;
        g.MTt[i] = ((i == 0 ? k.d : k.nin[i]) + 15) / 16; g.KGt[i] = (k.nout[i] + 15) / 16;
        g.offF[i] = oW; oW += g.MT[i] * g.KG[i] * 256;
        g.offT[i] = oW; oW += g.MTt[i] * g.KGt[i] * 256;
        // few inputs: 64 outputs x 16 inputs per unit wastes less of the four-tile operand than 16 x 64
        g.ukind[i] = k.nin[i] <= 32 && k.nout[i] > k.nin[i] ? 1 : ((k.nout[i] <= 16 && k.nin[i] > 16) ? 2 : 0);
        if (g.ukind[i] == 0) { g.unA[i] = (k.nout[i] + 15) / 16; g.unB[i] = (k.nin[i] + 63) / 64; }
        else if (g.ukind[i] == 1) { g.unA[i] = (k.nout[i] + 63) / 64; g.unB[i] = (k.nin[i] + 15) / 16; }
        else { g.unA[i] = 1; g.unB[i] = (k.nin[i] + 15) / 16; }
        g.uoff[i] = u; u += g.unA[i] * g.unB[i];
    }
    for (int i = g.nlin; i < kMaxLin; ++i) g.uoff[i] = u;
    g.nunits = u;
    g.net_floats = oW;
    g.xc_fgs = (k.d + k.c + 3) / 4; g.d_fgs = (k.d + 3) / 4;
    // every region is a whole number of 16-feature tiles: an out tile's epilogue writes all four of its feature groups
    auto tiles = [](int feats) { return (feats + 15) / 16 * 4; };
    int fg = 0;
    g.fgXC = fg; fg += tiles(k.d + k.c);
    for (int i = 0; i < k.nh; ++i) { g.fgA[i] = fg; fg += tiles(k.nout[i]); }
    g.h_fgs = fg - g.fgA[0];
    g.fgT = fg; fg += tiles(k.d);
    g.fgS = fg; fg += tiles(k.d);
    g.fgGY = fg; fg += tiles(k.d);
    g.fgGYB = fg; fg += tiles(k.d);
    g.fgGO = fg; fg += tiles(k.d);
    fg += kSlackFg;
    g.fg_total = fg;
    g.lds_bytes = ((size_t)fg * FS + 2 * kW8 * 64) * sizeof(float);
    return g;
}

std::atomic<uint64_t> g_attr_train64{0};

int grid64(int64_t n) {
    const int64_t nblocks = (n + BR - 1) / BR;
    return (int)(nblocks < 256 ? nblocks : 256);
}
// rows per pass: the saved layer inputs take 4 L (d + c) bytes per row, so a very large call goes through in chunks (every chunk on
// the same grid: workgroup w's partial always holds the same rows' sums)
constexpr int64_t kChunkRows = 262144;
// ... and the saved hidden activations 2 L x 16 bytes per hidden feature group and row: the chunk shrinks (in steps of a full grid
// of blocks) until they fit ~1 GiB
int64_t chunk_rows64(const KShape &k, const G64 &g) {
    const double per_row = (double)g.nnets * g.h_fgs * 16.0;
    int64_t r = (int64_t)((double)(1u << 30) / per_row) / 16384 * 16384;
    if (r < 16384) r = 16384;
    return r < kChunkRows ? r : kChunkRows;
}

}  // namespace

// the 64-row form serves a call when the image of a block fits one CU's LDS, a wave's register slots hold the net's units, and
// the batch is large enough to give most CUs a block (below that the 16-row form's 4x finer grain wins)
static bool train64_fits(const KShape &k) {
    const G64 g = make_g64(k);
    return g.lds_bytes <= 160 * 1024 && g.nunits <= kSlots * kW8;
}
bool use_train64(const KShape &k, int64_t n) {
    if (k.family == RNVP_FAMILY_LMM16 || !train64_fits(k)) return false;
    return k.family == RNVP_FAMILY_LMM64 || n >= 8192;
}

size_t train64_workspace_bytes(const KShape &k, int64_t max_rows) {
    const G64 g = make_g64(k);
    if (max_rows > chunk_rows64(k, g)) max_rows = chunk_rows64(k, g);
    const int64_t nblocks = (max_rows + BR - 1) / BR;
    size_t b = align_up((size_t)g.nnets * g.net_floats * sizeof(float), 256);
    b += align_up((size_t)256 * sizeof(float), 256);
    b += align_up((size_t)nblocks * k.L * g.xc_fgs * BR * 4 * sizeof(float), 256);
    b += 2 * align_up((size_t)nblocks * g.d_fgs * BR * 4 * sizeof(float), 256);
    b += align_up((size_t)nblocks * g.nnets * g.h_fgs * BR * 4 * sizeof(float), 256);                // hidden activations
    b += align_up((size_t)nblocks * k.L * g.d_fgs * BR * 4 * sizeof(float), 256);                    // s outputs
    b += align_up((size_t)grid64(max_rows) * g.nnets * g.nunits * kUnitFloats * sizeof(float), 256);
    return b;
}

int loss_grad64(hipStream_t st, const KShape &k, const float *params, const uint8_t *masks, const float *x, const float *c,
                const int64_t *row_index, int64_t n, float inv_B, float *grad_out, float *loss_out, void *ws, size_t ws_bytes,
                Seeds sd) {
    if (!ws || ws_bytes < train64_workspace_bytes(k, n)) return RNVP_EWORKSPACE;
    const G64 g = make_g64(k);
    const int64_t cr = n < chunk_rows64(k, g) ? n : chunk_rows64(k, g);
    const int64_t nblocks = (cr + BR - 1) / BR;
    const int G = grid64(cr);
    char *w = static_cast<char *>(ws);
    float *packed = reinterpret_cast<float *>(w); w += align_up((size_t)g.nnets * g.net_floats * sizeof(float), 256);
    float *losspart = reinterpret_cast<float *>(w); w += align_up((size_t)256 * sizeof(float), 256);
    float *xsave = reinterpret_cast<float *>(w); w += align_up((size_t)nblocks * k.L * g.xc_fgs * BR * 4 * sizeof(float), 256);
    float *gysave = reinterpret_cast<float *>(w); w += align_up((size_t)nblocks * g.d_fgs * BR * 4 * sizeof(float), 256);
    float *gybsave = reinterpret_cast<float *>(w); w += align_up((size_t)nblocks * g.d_fgs * BR * 4 * sizeof(float), 256);
    float *asave = reinterpret_cast<float *>(w); w += align_up((size_t)nblocks * g.nnets * g.h_fgs * BR * 4 * sizeof(float), 256);
    float *ssave = reinterpret_cast<float *>(w); w += align_up((size_t)nblocks * k.L * g.d_fgs * BR * 4 * sizeof(float), 256);
    float *gpart = reinterpret_cast<float *>(w);
    {
        const int64_t total = (int64_t)g.nnets * g.net_floats;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(k_lmm64_pack, dim3(blocks), dim3(256), 0, st, k, g, params, masks, packed);
        RNVP_HIP_TRY(hipGetLastError());
    }
    int rc = allow_big_lds(reinterpret_cast<const void *>(k_lmm_train64), 160 * 1024, g_attr_train64);
    if (rc) return rc;
    int launches = 2;
    for (int64_t r0 = 0; r0 < n; r0 += cr, ++launches) {
        const int64_t rows = n - r0 < cr ? n - r0 : cr;
        note_dispatch(RNVP_PROFILE_TRAIN, "k_lmm_train64", RNVP_VARIANT_LMM, 4, kW8, G, RNVP_PREC_F32, rows);
        KernelTimer timer(st, RNVP_PROFILE_TRAIN);
        hipLaunchKernelGGL(k_lmm_train64, dim3(G), dim3(64 * kW8), g.lds_bytes, st, k, g, packed, params, masks,
                           row_index ? x : x + r0 * k.d, (row_index || !c) ? c : c + r0 * k.c, row_index ? row_index + r0 : nullptr,
                           rows, inv_B,
                           Seeds{sd.gz ? sd.gz + r0 * k.d : nullptr, sd.gld ? sd.gld + r0 : nullptr, sd.gx ? sd.gx + r0 * k.d : nullptr},
                           xsave, gysave, gybsave, asave, ssave, gpart, losspart, r0 == 0 ? 1 : 0);
        RNVP_HIP_TRY(hipGetLastError());
    }
    const size_t total = (size_t)g.nnets * g.nunits * kUnitFloats;
    hipLaunchKernelGGL(k_lmm64_reduce, dim3((unsigned)(total / 256 + 2)), dim3(256), 0, st, k, g, gpart, G, masks, losspart, inv_B,
                       grad_out, loss_out);
    RNVP_HIP_TRY(hipGetLastError());
    note_launches(RNVP_PROFILE_TRAIN, launches);
    return RNVP_OK;
}

}  // namespace lmm
}  // namespace rnvp
