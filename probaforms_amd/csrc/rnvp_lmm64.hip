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

__device__ __forceinline__ f4 mfma16(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

struct G64 {
    int nlin, nnets;
    int nin[kMaxLin], nout[kMaxLin];
    int MT[kMaxLin], KG[kMaxLin];          // forward: out tiles (16), groups of 16 inputs
    int MTt[kMaxLin], KGt[kMaxLin];        // transposed (input gradient): in tiles, groups of 16 outputs
    int offF[kMaxLin], offT[kMaxLin];      // float offsets of the fragment blocks inside one net's packed image
    int net_floats;
    int ukind[kMaxLin];                    // weight-gradient units of Linear k: 0 = 16 outputs x 64 inputs, 1 = 64 outputs x 16 inputs
    int unA[kMaxLin], unB[kMaxLin], uoff[kMaxLin], nunits;
    int fgA[kMaxLin];                      // feature-group offset of hidden activation k in the image
    int fgXC, fgT, fgS, fgGY, fgGYB, fgGO, fg_total, xc_fgs, d_fgs;
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

// acc[t] (t < RG) = W tile m . in^T for row tiles rt0 .. rt0 + RG - 1: `ftile` = the tile's fragments [group][lane][4], `inb` the
// LDS region of the input.  Three operand sets rotate: two groups of fragments (L2) in flight while one multiplies.
template <int RG>
__device__ __forceinline__ void gemm_acc(const float *__restrict__ ftile, int KG, const float *inb, int rt0, int lane, f4 (&acc)[RG]) {
    const int q = lane >> 4, i = lane & 15;
    const float *pa = ftile + lane * 4;                          // + G * 256
    const float *pb = inb + q * FS + (16 * rt0 + i) * 4;         // + G * 4 * FS + t * 64
#pragma unroll
    for (int t = 0; t < RG; ++t) acc[t] = f4{0.f, 0.f, 0.f, 0.f};
    f4 a0, a1, a2, b0[RG], b1[RG], b2[RG];
    auto fetch = [&](f4 &a, f4 (&b)[RG], int G) {
        a = *reinterpret_cast<const f4 *>(pa + G * 256);
#pragma unroll
        for (int t = 0; t < RG; ++t) b[t] = *reinterpret_cast<const f4 *>(pb + G * 4 * FS + t * 64);
    };
    auto mul = [&](const f4 &a, const f4 (&b)[RG]) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int t = 0; t < RG; ++t) acc[t] = mfma16(a[e], b[t][e], acc[t]);
    };
    fetch(a0, b0, 0);
    if (1 < KG) fetch(a1, b1, 1);
    for (int G = 0; G < KG; G += 3) {
        if (G + 2 < KG) fetch(a2, b2, G + 2);
        mul(a0, b0);
        if (G + 1 >= KG) break;
        if (G + 3 < KG) fetch(a0, b0, G + 3);
        mul(a1, b1);
        if (G + 2 >= KG) break;
        if (G + 4 < KG) fetch(a1, b1, G + 4);
        mul(a2, b2);
    }
}

enum { EP_FWD_ACT = 0, EP_FWD_LIN, EP_GRAD, EP_GRAD0 };

// one Linear for the block: the (out tile, row-tile group) units go round-robin to the waves.  Epilogues:
//   EP_FWD_ACT  out = act(acc + bias)           EP_FWD_LIN  out = acc + bias
//   EP_GRAD     out = acc * act'(out)  (in place: `out` holds the activation, and receives the pre-activation gradient)
//   EP_GRAD0    out += acc             (input gradient of Linear 0, added to the layer's running d loss / d x)
// Outputs past nvalid are written as 0.
template <int RG, int EP>
__device__ __forceinline__ void gemm_units(const float *__restrict__ frag, int MT, int KG, const float *inb, float *outb,
                                           const float *__restrict__ bias, int nvalid, int act, int lane, int wave) {
    constexpr int per = 4 / RG;
    const int q = lane >> 4, i = lane & 15;
    for (int u = wave; u < MT * per; u += kW8) {
        const int m = u / per, rt0 = (u - m * per) * RG;
        f4 acc[RG];
        gemm_acc<RG>(frag + (size_t)m * KG * 256, KG, inb, rt0, lane, acc);
        const int o0 = 16 * m + 4 * q;
        f4 bv = f4{0.f, 0.f, 0.f, 0.f};
        if (EP == EP_FWD_ACT || EP == EP_FWD_LIN) {
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[e] = o0 + e < nvalid ? bias[o0 + e] : 0.f;
        }
#pragma unroll
        for (int t = 0; t < RG; ++t) {
            f4 *po = reinterpret_cast<f4 *>(outb + (4 * m + q) * FS + (16 * (rt0 + t) + i) * 4);
            f4 v, cur = f4{0.f, 0.f, 0.f, 0.f};
            if (EP == EP_GRAD || EP == EP_GRAD0) cur = *po;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float r = acc[t][e];
                if (EP == EP_FWD_ACT) r = act_fwd(r + bv[e], act);
                else if (EP == EP_FWD_LIN) r = r + bv[e];
                else if (EP == EP_GRAD) r = (act == RNVP_ACT_TANH) ? r * (1.f - cur[e] * cur[e]) : (cur[e] > 0.f ? r : 0.f);
                if (o0 + e >= nvalid) r = 0.f;
                v[e] = EP == EP_GRAD0 ? cur[e] + r : r;
            }
            *po = v;
        }
    }
}

template <int EP>
__device__ __forceinline__ void gemm(const float *__restrict__ frag, int MT, int KG, const float *inb, float *outb,
                                     const float *__restrict__ bias, int nvalid, int act, int lane, int wave) {
    if (MT >= 8) gemm_units<4, EP>(frag, MT, KG, inb, outb, bias, nvalid, act, lane, wave);
    else if (MT >= 4) gemm_units<2, EP>(frag, MT, KG, inb, outb, bias, nvalid, act, lane, wave);
    else gemm_units<1, EP>(frag, MT, KG, inb, outb, bias, nvalid, act, lane, wave);
}

// one net forward for the block, every hidden activation kept in its own region; the last Linear writes `out` (skipped when null)
__device__ __forceinline__ void net_fwd64(const float *__restrict__ pk, const float *__restrict__ pn, const KShape &s, const G64 &g,
                                          float *lds, float *out, int lane, int wave) {
    const float *cur = lds + g.fgXC * FS;
    for (int k = 0; k < g.nlin; ++k) {
        const bool last = k == g.nlin - 1;
        if (last && !out) break;
        float *ob = last ? out : lds + g.fgA[k] * FS;
        if (last) gemm<EP_FWD_LIN>(pk + g.offF[k], g.MT[k], g.KG[k], cur, ob, pn + s.boff[k], g.nout[k], s.act, lane, wave);
        else gemm<EP_FWD_ACT>(pk + g.offF[k], g.MT[k], g.KG[k], cur, ob, pn + s.boff[k], g.nout[k], s.act, lane, wave);
        __syncthreads();
        cur = ob;
    }
}

// weight-gradient unit, accumulated over the block's 64 rows (k-step ks of lane group q stands for row 4 ks + q):
//   KIND 0: outputs 16 ua .. + 15 (lane i: one value per k-step) x inputs 64 ub + 4 i + e (one b128: four tiles e)
//   KIND 1: outputs 64 ua + 4 i + e (one b128: four tiles e) x inputs 16 ub .. + 15
// bs collects the sums over rows of the A operand (the bias gradient, wanted from the unit with ub == 0).
template <int KIND>
__device__ __forceinline__ void wgrad_unit(const float *ldsA, const float *ldsB, int ua, int ub, int lane, f4 (&acc)[4], f4 &bs) {
    const int q = lane >> 4, i = lane & 15;
    if (KIND == 0) {
        const float *pa = ldsA + (4 * ua + (i >> 2)) * FS + q * 4 + (i & 3);
        const float *pb = ldsB + (16 * ub + i) * FS + q * 4;
        float a[2]; f4 b[2];
        a[0] = pa[0]; b[0] = *reinterpret_cast<const f4 *>(pb);
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            if (ks + 1 < 16) { a[(ks + 1) & 1] = pa[(ks + 1) * 16]; b[(ks + 1) & 1] = *reinterpret_cast<const f4 *>(pb + (ks + 1) * 16); }
            const float av = a[ks & 1]; const f4 bv = b[ks & 1];
            bs[0] += av;
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = mfma16(av, bv[e], acc[e]);
        }
    } else {
        const float *pa = ldsA + (16 * ua + i) * FS + q * 4;
        const float *pb = ldsB + (4 * ub + (i >> 2)) * FS + q * 4 + (i & 3);
        f4 a[2]; float b[2];
        a[0] = *reinterpret_cast<const f4 *>(pa); b[0] = pb[0];
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            if (ks + 1 < 16) { a[(ks + 1) & 1] = *reinterpret_cast<const f4 *>(pa + (ks + 1) * 16); b[(ks + 1) & 1] = pb[(ks + 1) * 16]; }
            const f4 av = a[ks & 1]; const float bv = b[ks & 1];
            bs += av;
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = mfma16(av[e], bv, acc[e]);
        }
    }
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

__global__ void __launch_bounds__(64 * kW8)
k_lmm_train64(KShape s, G64 g, const float *__restrict__ packed, const float *__restrict__ params,
              const uint8_t *__restrict__ masks, const float *__restrict__ x, const float *__restrict__ c,
              const int64_t *__restrict__ row_index, int64_t n, float inv_B, Seeds sd, float *__restrict__ xsave,
              float *__restrict__ gysave, float *__restrict__ gybsave, float *__restrict__ gpart, float *__restrict__ losspart,
              int first_chunk) {
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
    const size_t xs_blk = (size_t)s.L * g.xc_fgs * BR * 4, gy_blk = (size_t)g.d_fgs * BR * 4;
    float wave_sum = 0.f;

    // ---- forward over the workgroup's blocks: saves every layer's input image and the seed of the backward ----
    for (int64_t b = blockIdx.x; b < nblocks; b += gridDim.x) {
        const int64_t base = b * BR;
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
        __syncthreads();
        float ld = 0.f;
        for (int l = 0; l < s.L; ++l) {
            const float *pk = packed + (size_t)l * 2 * g.net_floats, *pn = params + (size_t)l * 2 * s.npn;
            region_store(XC, xsave + (size_t)b * xs_blk + (size_t)l * g.xc_fgs * BR * 4, g.xc_fgs, tid);
            net_fwd64(pk, pn, s, g, lds, T, lane, wave);
            net_fwd64(pk + g.net_floats, pn + s.npn, s, g, lds, S, lane, wave);
            const uint8_t *m = masks + l * d;
            for (int fg = wave; fg < g.d_fgs; fg += kW8) {
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
            __syncthreads();
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
            __syncthreads();
            if (wave == 0) {
                float lds_ = 0.f, sss = 0.f;
#pragma unroll
                for (int w = 0; w < kW8; ++w) { lds_ += RED[w * 64 + row]; sss += RED[(kW8 + w) * 64 + row]; }
                float v = valid ? (gz ? lds_ : lds_ + (-0.5f * sss - prior_c)) : 0.f;
                for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
                wave_sum += v;
            }
            __syncthreads();
        }
    }
    if (tid == 0) losspart[blockIdx.x] = first_chunk ? wave_sum : losspart[blockIdx.x] + wave_sum;      // chunks run in order on one stream

    // ---- backward: layer-major over the blocks; a wave's weight-gradient units stay in registers across its blocks ----
    for (int l = s.L - 1; l >= 0; --l) {
        const uint8_t *m = masks + l * d;
        for (int net = 1; net >= 0; --net) {                     // s first: t only adds to what s leaves in GYB
            const float *pkn = packed + ((size_t)l * 2 + net) * g.net_floats, *pnn = params + ((size_t)l * 2 + net) * s.npn;
            f4 acc[kSlots][4], bs[kSlots];
#pragma unroll
            for (int sl = 0; sl < kSlots; ++sl) {
                bs[sl] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[sl][e] = f4{0.f, 0.f, 0.f, 0.f};
            }
            for (int64_t b = blockIdx.x; b < nblocks; b += gridDim.x) {
                const int64_t r = b * BR + row;
                const bool valid = r < n;
                region_load(XC, xsave + (size_t)b * xs_blk + (size_t)l * g.xc_fgs * BR * 4, g.xc_fgs, tid);
                region_load(GY, gysave + (size_t)b * gy_blk, g.d_fgs, tid);
                if (net == 0) region_load(GYB, gybsave + (size_t)b * gy_blk, g.d_fgs, tid);
                __syncthreads();
                net_fwd64(pkn, pnn, s, g, lds, net ? S : nullptr, lane, wave);
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
                __syncthreads();
                for (int k = g.nlin - 1; k >= 0; --k) {
                    const float *ga = k == g.nlin - 1 ? GO : lds + g.fgA[k] * FS;              // pre-activation gradient of Linear k
                    float *in = k == 0 ? XC : lds + g.fgA[k - 1] * FS;                         // its input
#pragma unroll
                    for (int sl = 0; sl < kSlots; ++sl) {
                        const int ul = kW8 * sl + wave - g.uoff[k];
                        if (ul >= 0 && ul < g.unA[k] * g.unB[k]) {
                            const int ua = ul / g.unB[k], ub = ul - ua * g.unB[k];
                            if (g.ukind[k] == 0) wgrad_unit<0>(ga, in, ua, ub, lane, acc[sl], bs[sl]);
                            else wgrad_unit<1>(ga, in, ua, ub, lane, acc[sl], bs[sl]);
                        }
                    }
                    __syncthreads();
                    if (k == 0) gemm<EP_GRAD0>(pkn + g.offT[0], g.MTt[0], g.KGt[0], ga, GYB, nullptr, d, s.act, lane, wave);
                    else gemm<EP_GRAD>(pkn + g.offT[k], g.MTt[k], g.KGt[k], ga, in, nullptr, g.nin[k], s.act, lane, wave);
                    __syncthreads();
                }
                region_store(GYB, (net ? gybsave : gysave) + (size_t)b * gy_blk, g.d_fgs, tid);
                if (net == 0 && l == 0 && sd.gx) {                      // rnvp_backward: d loss / d x of the batch rows
                    for (int fg = wave; fg < g.d_fgs; fg += kW8) {
                        const f4 gv = *reinterpret_cast<const f4 *>(GYB + fg * FS + row * 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { const int j = 4 * fg + e; if (valid && j < d) sd.gx[r * d + j] = gv[e]; }
                    }
                }
                __syncthreads();
            }
            // flush this (layer, net)'s units: [workgroup][layer, net][unit][4 tiles x 64 lanes x 4 | 64 bias sums]
            float *dst = gpart + (((size_t)blockIdx.x * g.nnets + (size_t)l * 2 + net) * g.nunits) * kUnitFloats;
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
    }
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
    int k = 0;
    while (k + 1 < g.nlin && u >= g.uoff[k + 1]) ++k;
    const int ul = u - g.uoff[k], ua = ul / g.unB[k], ub = ul - ua * g.unB[k];
    const int nin = g.nin[k], nout = g.nout[k];
    int out, in;
    if (j < 1024) {
        const int tile = j >> 8, lane = (j >> 2) & 63, e = j & 3, q = lane >> 4, i = lane & 15;
        if (g.ukind[k] == 0) { out = 16 * ua + 4 * q + e; in = 64 * ub + 4 * i + tile; }
        else { out = 64 * ua + 4 * (4 * q + e) + tile; in = 16 * ub + i; }
        if (out >= nout || in >= nin) return;
    } else {
        if (ub != 0) return;
        const int jj = j - 1024;
        if (g.ukind[k] == 0) { if (jj & 3) return; out = 16 * ua + (jj >> 2); }       // lane i wrote (sum, 0, 0, 0)
        else out = 64 * ua + jj;                                                       // lane i wrote outputs 64 ua + 4 i + e
        if (out >= nout) return;
        in = nin;
    }
    const float *src = gpart + (size_t)ln * per_net + rest;
    const size_t stride = total;
    float a = 0.f;
    for (int w = 0; w < G; ++w) a += src[(size_t)w * stride];
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
    for (int i = 0; i < g.nlin; ++i) {
        g.nin[i] = k.nin[i]; g.nout[i] = k.nout[i];
        g.MT[i] = (k.nout[i] + 15) / 16; g.KG[i] = (k.nin[i] + 15) / 16;
        g.MTt[i] = ((i == 0 ? k.d : k.nin[i]) + 15) / 16; g.KGt[i] = (k.nout[i] + 15) / 16;
        g.offF[i] = oW; oW += g.MT[i] * g.KG[i] * 256;
        g.offT[i] = oW; oW += g.MTt[i] * g.KGt[i] * 256;
        // few inputs: 64 outputs x 16 inputs per unit wastes less of the four-tile operand than 16 x 64
        g.ukind[i] = k.nin[i] <= 32 && k.nout[i] > k.nin[i] ? 1 : 0;
        if (g.ukind[i] == 0) { g.unA[i] = (k.nout[i] + 15) / 16; g.unB[i] = (k.nin[i] + 63) / 64; }
        else { g.unA[i] = (k.nout[i] + 63) / 64; g.unB[i] = (k.nin[i] + 15) / 16; }
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
    if (max_rows > kChunkRows) max_rows = kChunkRows;
    const int64_t nblocks = (max_rows + BR - 1) / BR;
    size_t b = align_up((size_t)g.nnets * g.net_floats * sizeof(float), 256);
    b += align_up((size_t)256 * sizeof(float), 256);
    b += align_up((size_t)nblocks * k.L * g.xc_fgs * BR * 4 * sizeof(float), 256);
    b += 2 * align_up((size_t)nblocks * g.d_fgs * BR * 4 * sizeof(float), 256);
    b += align_up((size_t)grid64(max_rows) * g.nnets * g.nunits * kUnitFloats * sizeof(float), 256);
    return b;
}

int loss_grad64(hipStream_t st, const KShape &k, const float *params, const uint8_t *masks, const float *x, const float *c,
                const int64_t *row_index, int64_t n, float inv_B, float *grad_out, float *loss_out, void *ws, size_t ws_bytes,
                Seeds sd) {
    if (!ws || ws_bytes < train64_workspace_bytes(k, n)) return RNVP_EWORKSPACE;
    const G64 g = make_g64(k);
    const int64_t cr = n < kChunkRows ? n : kChunkRows;
    const int64_t nblocks = (cr + BR - 1) / BR;
    const int G = grid64(cr);
    char *w = static_cast<char *>(ws);
    float *packed = reinterpret_cast<float *>(w); w += align_up((size_t)g.nnets * g.net_floats * sizeof(float), 256);
    float *losspart = reinterpret_cast<float *>(w); w += align_up((size_t)256 * sizeof(float), 256);
    float *xsave = reinterpret_cast<float *>(w); w += align_up((size_t)nblocks * k.L * g.xc_fgs * BR * 4 * sizeof(float), 256);
    float *gysave = reinterpret_cast<float *>(w); w += align_up((size_t)nblocks * g.d_fgs * BR * 4 * sizeof(float), 256);
    float *gybsave = reinterpret_cast<float *>(w); w += align_up((size_t)nblocks * g.d_fgs * BR * 4 * sizeof(float), 256);
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
                           xsave, gysave, gybsave, gpart, losspart, r0 == 0 ? 1 : 0);
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
