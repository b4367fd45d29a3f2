// cvae_mfma.hip -- conditional VAE training step on f32 MFMA (gfx950): encoder -> reparameterise ->
// decoder -> KL + MSE loss -> backward, one kernel, built from the register-chained blocks of the
// RealNVP path (rnvp_mfma_layer.h).  Replaces CVAE.compute_loss + loss.backward()
// (/root/reference/probaforms/models/cvae.py:186-203,243-246) for d <= 16, cdim <= 4, latent <= 4, one
// hidden layer of any width, tanh (BASELINE.json configs[4] and the reference's defaults); other
// shapes run on cvae_generic.hip.  C ABI: include/cvae_hip.h (cvae_loss_grad).
//
// Lane (q = lane >> 4, r = lane & 15) holds, for row r: x[4q..4q+3], c[q], eps[q], and -- after a
// reduce-scatter over the lane groups -- mu[q], log_sigma[q], z[q].
//   encoder : GEMM1 (K = 4 x-steps + 1 c-step) -> tanh -> heads as 4x4x1 blocks (mu | log_sigma)
//   z       : mu + exp(log_sigma / 2) * eps, per lane
//   decoder : GEMM1 (k-steps: z, c) -> tanh -> GEMM2 (one 16-row out tile: lane gets x_rec[4q..4q+3])
//   backward: per hidden tile recompute GEMM1 + tanh; g_h by MFMA from register-resident g_out;
//             d loss / d z as 4x4x1 blocks; weight gradients contract over rows through wave-private
//             LDS transposition tiles, a ones column yields the hidden biases; per-wave LDS slots are
//             added in wave order into the workgroup's partial (deterministic, no float atomics).
#include <atomic>

#include "../../include/cvae_hip.h"
#include "rnvp_mfma_layer.h"

#ifndef RNVP_WPE
#define RNVP_WPE 2
#endif

namespace rnvp {
namespace cvae_mfma {
namespace {

using mfma::f4;
using mfma::mfma16;
using mfma::mfma4;
using mfma::tanh4;
using mfma::swap_add32;
using mfma::swap_add16;
using mfma::opaque;
using mfma::transpose16;
using mfma::wave_lds_fence;
using mfma::kTS;
using mfma::kTanhScale;

#ifndef CVAE_R
#define CVAE_R 4
#endif
#ifndef CVAE_FT
#define CVAE_FT 8
#endif
#ifndef CVAE_WAVES
#define CVAE_WAVES 4
#endif
// CVAE_SAVE_H: the encoder's and the decoder's hidden activations are stored by the forward (16 B per lane, hidden tile and row
// tile: 4 h bytes per row and net, 1 KB per row at C5 = 67 MB per 65 536-row step, which the 256 MB memory-side cache holds) and
// read back, one hidden tile ahead, by the two backward loops instead of recomputing GEMM1 + tanh there: 5 + 2 of the 19 + 14
// f32 MFMAs per unit and both tanh evaluations.  (The flow kernels' records are 8x larger per row -- eight layers -- and lose:
// rnvp_mfma_train_dev.h RNVP_SAVE_H.)
#ifndef CVAE_SAVE_H
#define CVAE_SAVE_H 1
#endif
constexpr bool kSaveH = CVAE_SAVE_H != 0;
constexpr int kWaves = CVAE_WAVES, kR = CVAE_R, kMaxGrid = 512, kFT = CVAE_FT;

// CVAE_STAMP: diagnostic build that accumulates cycle-counter deltas per phase and printf()s them for two workgroups
#ifdef CVAE_STAMP
#define CSTAMP(var) do { __builtin_amdgcn_sched_barrier(0); var = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define CSTAMP_ADD(acc, t0) do { unsigned long long t1__; CSTAMP(t1__); acc += t1__ - t0; t0 = t1__; } while (0)
#else
#define CSTAMP(var) do { } while (0)
#define CSTAMP_ADD(acc, t0) do { } while (0)
#endif

struct CG {                      // geometry + offsets (floats)
    int d, c, lat, h, HT;
    // flat (oracle-order) parameter offsets
    int fW1e, fb1e, fWmu, fWls, fbmu, fbls, fW1d, fb1d, fW2d, fb2d, P;
    // packed weights
    int oA1E, oB1E, oA2E, oA2ET, oA1D, oB1D, oA2D, oA2DT, oA1DX, oBH, oB2D, packed_floats;
    // per-workgroup partial gradient: [enc: HT x 3 blocks][dec: HT x 2 blocks][head biases 16][b2d 16]
    int gEnc, gDec, gBH, gB2D, gfloats;
};

CG make_cg(const cvae_shape *s) {
    CG g;
    g.d = s->d; g.c = s->c; g.lat = s->lat; g.h = s->hidden[0]; g.HT = (g.h + 15) / 16;
    const int ne = g.d + g.c, nd = g.lat + g.c;
    int o = 0;
    g.fW1e = o; o += g.h * ne; g.fb1e = o; o += g.h;
    g.fWmu = o; o += g.lat * g.h; g.fWls = o; o += g.lat * g.h; g.fbmu = o; o += g.lat; g.fbls = o; o += g.lat;
    g.fW1d = o; o += g.h * nd; g.fb1d = o; o += g.h; g.fW2d = o; o += g.d * g.h; g.fb2d = o; o += g.d;
    g.P = o;
    o = 0;
    g.oA1E = o; o += g.HT * 2 * 256; g.oB1E = o; o += g.HT * 16;
    g.oA2E = o; o += g.HT * 2 * 256; g.oA2ET = o; o += g.HT * 256;
    g.oA1D = o; o += g.HT * 256; g.oB1D = o; o += g.HT * 16;
    g.oA2D = o; o += g.HT * 256; g.oA2DT = o; o += g.HT * 256; g.oA1DX = o; o += g.HT * 256;
    g.oBH = o; o += 16; g.oB2D = o; o += 16;
    g.packed_floats = o;
    o = 0;
    g.gEnc = o; o += g.HT * 3 * 256; g.gDec = o; o += g.HT * 2 * 256; g.gBH = o; o += 16; g.gB2D = o; o += 16;
    g.gfloats = o;
    return g;
}

// ---- packing ----------------------------------------------------------------------------------------
// parameter source: the flat reference-order buffer (k_pack) ...
struct FlatCvae {
    const float *p;
    const CG &g;
    __device__ float w1e(int hid, int col) const { return p[g.fW1e + hid * (g.d + g.c) + col]; }
    __device__ float b1e(int hid) const { return p[g.fb1e + hid]; }
    __device__ float whead(int og, int i, int hid) const { return p[(og == 0 ? g.fWmu : g.fWls) + i * g.h + hid]; }
    __device__ float bhead(int og, int i) const { return p[(og == 0 ? g.fbmu : g.fbls) + i]; }
    __device__ float w1d(int hid, int col) const { return p[g.fW1d + hid * (g.lat + g.c) + col]; }
    __device__ float b1d(int hid) const { return p[g.fb1d + hid]; }
    __device__ float w2d(int out, int hid) const { return p[g.fW2d + out * g.h + hid]; }
    __device__ float b2d(int j) const { return p[g.fb2d + j]; }
};

template <class Src>
__device__ float pack_value(const CG &g, int idx, const Src &src) {
    const int h = g.h;
    auto W1e = [&](int hid, int col) { return (hid < h && col >= 0) ? src.w1e(hid, col) : 0.f; };
    auto W1d = [&](int hid, int col) { return (hid < h && col >= 0) ? src.w1d(hid, col) : 0.f; };
    auto Whead = [&](int og, int i, int hid) { return (hid < h && i < g.lat) ? src.whead(og, i, hid) : 0.f; };
    auto W2d = [&](int out, int hid) { return (hid < h && out < g.d) ? src.w2d(out, hid) : 0.f; };
    if (idx < g.oB1E) {                                        // A1E [t][2][lane][4]: k-steps 0-3 x, 4 c
        const int e = idx & 3, lane = (idx >> 2) & 63, rest = idx >> 8, k4 = rest & 1, t = rest >> 1;
        const int kk = 4 * k4 + e, q = lane >> 4, i = lane & 15, hid = 16 * t + i;
        int col = -1;
        if (kk < 4) { const int j = 4 * q + kk; col = j < g.d ? j : -1; }
        else if (kk == 4) col = q < g.c ? g.d + q : -1;
        return kTanhScale * W1e(hid, col);
    }
    if (idx < g.oA2E) {                                        // B1E [t][q][4]
        const int j = idx - g.oB1E, hid = 16 * (j >> 4) + (j & 15);
        return hid < h ? kTanhScale * src.b1e(hid) : 0.f;
    }
    if (idx < g.oA2ET) {                                       // A2E [t][og][lane][4 rho]  (4x4x1 heads)
        const int j = idx - g.oA2E, rho = j & 3, lane = (j >> 2) & 63, rest = j >> 8, og = rest & 1, t = rest >> 1;
        return Whead(og, lane & 3, 16 * t + 4 * (lane >> 4) + rho);
    }
    if (idx < g.oA1D) {                                        // A2ET [t][lane][4]: k-step s: head (s, q)
        const int j = idx - g.oA2ET, sidx = j & 3, lane = (j >> 2) & 63, t = j >> 8;
        return sidx < 2 ? Whead(sidx, lane >> 4, 16 * t + (lane & 15)) : 0.f;
    }
    if (idx < g.oB1D) {                                        // A1D [t][lane][4]: k-step 0 z[q], 1 c[q]
        const int j = idx - g.oA1D, kk = j & 3, lane = (j >> 2) & 63, t = j >> 8, q = lane >> 4, hid = 16 * t + (lane & 15);
        int col = -1;
        if (kk == 0) col = q < g.lat ? q : -1;
        else if (kk == 1) col = q < g.c ? g.lat + q : -1;
        return kTanhScale * W1d(hid, col);
    }
    if (idx < g.oA2D) {                                        // B1D [t][q][4]
        const int j = idx - g.oB1D, hid = 16 * (j >> 4) + (j & 15);
        return hid < h ? kTanhScale * src.b1d(hid) : 0.f;
    }
    if (idx < g.oA2DT) {                                       // A2D [t][lane][4 rho]: A[i = out][k = q <-> hid 16t+4q+rho]
        const int j = idx - g.oA2D, rho = j & 3, lane = (j >> 2) & 63, t = j >> 8;
        return W2d(lane & 15, 16 * t + 4 * (lane >> 4) + rho);
    }
    if (idx < g.oA1DX) {                                       // A2DT [t][lane][4 rho]: A[i = hid][k = q <-> out 4q+rho]
        const int j = idx - g.oA2DT, rho = j & 3, lane = (j >> 2) & 63, t = j >> 8;
        return W2d(4 * (lane >> 4) + rho, 16 * t + (lane & 15));
    }
    if (idx < g.oBH) {                                         // A1DX [t][lane][4 rho]: W1d[hid 16t+4q+rho][z col i]
        const int j = idx - g.oA1DX, rho = j & 3, lane = (j >> 2) & 63, t = j >> 8, i = lane & 3;
        return W1d(16 * t + 4 * (lane >> 4) + rho, i < g.lat ? i : -1);
    }
    if (idx < g.oB2D) {                                        // BH [q][4]: b_mu[q], b_ls[q]
        const int j = idx - g.oBH, e = j & 3, q = j >> 2;
        if (q >= g.lat || e > 1) return 0.f;
        return src.bhead(e, q);
    }
    {                                                          // B2D [q][4]
        const int j = idx - g.oB2D;
        return j < g.d ? src.b2d(j) : 0.f;
    }
}

__global__ void __launch_bounds__(256) k_pack(CG g, const float *__restrict__ params, float *__restrict__ packed) {
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < g.packed_floats; t += gridDim.x * blockDim.x)
        packed[t] = pack_value(g, t, FlatCvae{params, g});
}

// lane group q keeps element q of a 4-vector whose partial sums are spread over the 4 lane groups
__device__ __forceinline__ float reduce_scatter4(f4 v) {
    return swap_add16(swap_add32(v[0], v[2]), swap_add32(v[1], v[3]));
}

__device__ __forceinline__ float row16_sum(float v) {
    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
    return v;
}

// ---- the fused step --------------------------------------------------------------------------------------
// CVAE_TRAIN_WPE: the step kernel's 134 KB of LDS admit ONE workgroup of four waves per CU -- one wave per SIMD whatever the
// attribute says -- so it is compiled for that: the whole 512-entry register file (no spills; with the translation unit's
// -amdgpu-mfma-vgpr-form, csrc/Makefile, still VGPR-form MFMAs).  Step 55.5 -> 54.2 us (profiles/r06_sched_strategy_ab.txt, E).
#ifndef CVAE_TRAIN_WPE
#define CVAE_TRAIN_WPE 1
#endif
__global__ void __launch_bounds__(kWaves * 64) __attribute__((amdgpu_waves_per_eu(CVAE_TRAIN_WPE, CVAE_TRAIN_WPE)))
k_cvae_mfma(CG g, const float *__restrict__ wp, const float *__restrict__ x, const float *__restrict__ c,
            const int64_t *__restrict__ row_index, const float *__restrict__ eps, int64_t n, float inv_B, float klw,
            float *gpart, float *losspart, int do_grad, float *hsave) {
    constexpr int R = kR, RH = 2;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, tid = threadIdx.x;
    const int q = lane >> 4, r = lane & 15;
    const int HT = g.HT;
    const int SLOT = kFT * 3 * 256 + 32;                          // floats of one wave's slot (+ bias sums)
    float *slot = lds + wave * SLOT;
    float *tb = lds + kWaves * SLOT + wave * (7 * 16 * kTS);      // wave-private transposition tiles
    float *bufG = tb, *bufI = tb + 16 * kTS, *bufH = tb + 3 * 16 * kTS;   // g_out^T | [in|1]^T (2 tiles) | (h, g_pre) x RH
    const int64_t rows_per_wg = (int64_t)kWaves * R * 16;
    const int64_t ngroups = (n + rows_per_wg - 1) / rows_per_wg;
    const bool full = (g.d == 16) && (g.c == 4);
    float *gp = gpart + (size_t)blockIdx.x * g.gfloats;
    // CVAE_SAVE_H: this wave's records [net][tile][row tile][lane] f4
    float *hsE = hsave + ((size_t)blockIdx.x * kWaves + wave) * 2 * HT * R * 256 + lane * 4, *hsD = hsE + (size_t)HT * R * 256;
    const bool sv = kSaveH && do_grad;
    float wave_sum = 0.f;
    bool first = true;
    const f4 bh = *reinterpret_cast<const f4 *>(wp + g.oBH + q * 4);
    const f4 b2d = *reinterpret_cast<const f4 *>(wp + g.oB2D + q * 4);
    unsigned long long tk0 = 0, t0 = 0, s_ld = 0, s_ef = 0, s_df = 0, s_loss = 0, s_db = 0, s_dfl = 0, s_mid = 0, s_eb = 0, s_efl = 0;
    (void)tk0; (void)t0; (void)s_ld; (void)s_ef; (void)s_df; (void)s_loss; (void)s_db; (void)s_dfl; (void)s_mid; (void)s_eb; (void)s_efl;
    CSTAMP(tk0);
    for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int64_t base = grp * rows_per_wg + (int64_t)wave * R * 16;
        CSTAMP(t0);
        float xr[R][4], cr[R][1], er[R], mu[R], ls[R], el[R], z[R];
        bool valid[R];
#pragma unroll
        for (int rt = 0; rt < R; ++rt) {
            const int64_t row = base + rt * 16 + r;
            valid[rt] = row < n;
            const int64_t src = valid[rt] ? (row_index ? row_index[row] : row) : 0;
            mfma::load_row<2, 1>(x, c, src, g.d, g.c, full, q, xr[rt], cr[rt]);
            er[rt] = (valid[rt] && q < g.lat) ? eps[row * g.lat + q] : 0.f;
        }
        CSTAMP_ADD(s_ld, t0);
        // ---- encoder forward ---------------------------------------------------------------------
        {
            f4 outE[R][2];
#pragma unroll
            for (int rt = 0; rt < R; ++rt) { outE[rt][0] = f4{0.f, 0.f, 0.f, 0.f}; outE[rt][1] = f4{0.f, 0.f, 0.f, 0.f}; }
            const float *pA1 = wp + g.oA1E + lane * 4, *pB1 = wp + g.oB1E + q * 4, *pA2 = wp + g.oA2E + lane * 4;
            // the fragments of hidden tile t + 1 are requested before tile t is multiplied (one wave per SIMD: nothing
            // else would cover the L2 round trip)
            f4 a10 = *opaque(pA1), a11 = *opaque(pA1 + 256), b1 = *opaque(pB1), a20 = *opaque(pA2), a21 = *opaque(pA2 + 256);
            for (int t = 0; t < HT; ++t) {
                const int nx = t + 1 < HT ? t + 1 : t;
                const f4 na10 = *opaque(pA1 + (size_t)(2 * nx) * 256), na11 = *opaque(pA1 + (size_t)(2 * nx + 1) * 256);
                const f4 nb1 = *opaque(pB1 + nx * 16);
                const f4 na20 = *opaque(pA2 + (size_t)(2 * nx) * 256), na21 = *opaque(pA2 + (size_t)(2 * nx + 1) * 256);
                f4 hv[R];
#pragma unroll
                for (int rt = 0; rt < R; ++rt) {
                    f4 acc = b1;
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) acc = mfma16(a10[kk], xr[rt][kk], acc);
                    acc = mfma16(a11[0], cr[rt][0], acc);
                    hv[rt] = tanh4(acc);
                    if (sv) *reinterpret_cast<f4 *>(hsE + ((size_t)t * R + rt) * 256) = hv[rt];
                }
#pragma unroll
                for (int rho = 0; rho < 4; ++rho)
#pragma unroll
                    for (int rt = 0; rt < R; ++rt) {
                        outE[rt][0] = mfma4(a20[rho], hv[rt][rho], outE[rt][0]);
                        outE[rt][1] = mfma4(a21[rho], hv[rt][rho], outE[rt][1]);
                    }
                a10 = na10; a11 = na11; b1 = nb1; a20 = na20; a21 = na21;
            }
#pragma unroll
            for (int rt = 0; rt < R; ++rt) {
                mu[rt] = reduce_scatter4(outE[rt][0]) + bh[0];
                ls[rt] = reduce_scatter4(outE[rt][1]) + bh[1];
                el[rt] = expf(0.5f * ls[rt]);
                z[rt] = fmaf(el[rt], er[rt], mu[rt]);                                      // cvae.py:188
            }
        }
        CSTAMP_ADD(s_ef, t0);
        // ---- decoder forward -----------------------------------------------------------------------
        f4 xrec[R];
        {
#pragma unroll
            for (int rt = 0; rt < R; ++rt) xrec[rt] = b2d;
            const float *pA1 = wp + g.oA1D + lane * 4, *pB1 = wp + g.oB1D + q * 4, *pA2 = wp + g.oA2D + lane * 4;
            f4 a1 = *opaque(pA1), b1 = *opaque(pB1), a2 = *opaque(pA2);
            for (int t = 0; t < HT; ++t) {
                const int nx = t + 1 < HT ? t + 1 : t;
                const f4 na1 = *opaque(pA1 + (size_t)nx * 256), nb1 = *opaque(pB1 + nx * 16), na2 = *opaque(pA2 + (size_t)nx * 256);
                f4 hv[R];
#pragma unroll
                for (int rt = 0; rt < R; ++rt) {
                    f4 acc = mfma16(a1[0], z[rt], b1);
                    acc = mfma16(a1[1], cr[rt][0], acc);
                    hv[rt] = tanh4(acc);
                    if (sv) *reinterpret_cast<f4 *>(hsD + ((size_t)t * R + rt) * 256) = hv[rt];
                }
#pragma unroll
                for (int rho = 0; rho < 4; ++rho)
#pragma unroll
                    for (int rt = 0; rt < R; ++rt) xrec[rt] = mfma16(a2[rho], hv[rt][rho], xrec[rt]);
                a1 = na1; b1 = nb1; a2 = na2;
            }
        }
        CSTAMP_ADD(s_df, t0);
        // ---- loss: KL_weight * KL + MSE (cvae.py:190-193) ----------------------------------------
        f4 gx[R];
#pragma unroll
        for (int rt = 0; rt < R; ++rt) {
            float kl = (q < g.lat) ? (1.f + ls[rt] - mu[rt] * mu[rt] - expf(ls[rt])) : 0.f, se = 0.f;
            const float sc = valid[rt] ? inv_B : 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float df = xrec[rt][e] - xr[rt][e];
                se = fmaf(df, df, se);
                gx[rt][e] = (2.f * sc / (float)g.d) * df;                                    // d MSE / d x_rec
            }
            kl += __shfl_xor(kl, 16); kl += __shfl_xor(kl, 32);
            se += __shfl_xor(se, 16); se += __shfl_xor(se, 32);
            float v = (valid[rt] && q == 0) ? klw * (-0.5f * kl) + se / (float)g.d : 0.f;
            wave_sum += row16_sum(v);
        }
        if (!do_grad) continue;                                                              // uniform
        CSTAMP_ADD(s_loss, t0);

        // ---- decoder backward ------------------------------------------------------------------------
        float goT[R][4], inT[R][4];
        f4 gb2 = gx[0];
#pragma unroll
        for (int rt = 1; rt < R; ++rt) gb2 += gx[rt];
#pragma unroll
        for (int rt = 0; rt < R; ++rt) {
            transpose16(bufG, gx[rt], lane, goT[rt]);
            transpose16(bufI, f4{z[rt], cr[rt][0], q == 0 ? 1.f : 0.f, 0.f}, lane, inT[rt]);       // [z | c | 1]
        }
        f4 ginz[R];
#pragma unroll
        for (int rt = 0; rt < R; ++rt) ginz[rt] = f4{0.f, 0.f, 0.f, 0.f};
        {
            const float *pA1 = wp + g.oA1D + lane * 4, *pB1 = wp + g.oB1D + q * 4;
            const float *pA2T = wp + g.oA2DT + lane * 4, *pA1X = wp + g.oA1DX + lane * 4;
            f4 a1 = *opaque(pA1), b1 = *opaque(pB1), a2t = *opaque(pA2T), a1x = *opaque(pA1X);
            f4 hc[kSaveH ? R : 1], hn[kSaveH ? R : 1];
            if constexpr (kSaveH) {
#pragma unroll
                for (int rt = 0; rt < R; ++rt) hc[rt] = *reinterpret_cast<const f4 *>(hsD + (size_t)rt * 256);
            }
            for (int t = 0; t < HT; ++t) {
                const int nx = t + 1 < HT ? t + 1 : t;
                f4 na1 = a1, nb1 = b1;
                if constexpr (!kSaveH) { na1 = *opaque(pA1 + (size_t)nx * 256); nb1 = *opaque(pB1 + nx * 16); }
                else {
#pragma unroll
                    for (int rt = 0; rt < R; ++rt) hn[rt] = *opaque(hsD + ((size_t)nx * R + rt) * 256);
                }
                const f4 na2t = *opaque(pA2T + (size_t)nx * 256), na1x = *opaque(pA1X + (size_t)nx * 256);
                f4 gW1 = f4{0.f, 0.f, 0.f, 0.f}, gW2 = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r0 = 0; r0 < R; r0 += RH) {
                    f4 gpv[RH];
                    wave_lds_fence();
#pragma unroll
                    for (int u = 0; u < RH; ++u) {
                        const int rt = r0 + u;
                        f4 hv;
                        if constexpr (kSaveH) hv = hc[rt];
                        else {
                            f4 acc = mfma16(a1[0], z[rt], b1);
                            acc = mfma16(a1[1], cr[rt][0], acc);
                            hv = tanh4(acc);
                        }
                        f4 gh = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int rho = 0; rho < 4; ++rho) gh = mfma16(a2t[rho], gx[rt][rho], gh);
                        gpv[u] = gh * (1.0f - hv * hv);
                        *reinterpret_cast<f4 *>(bufH + (2 * u) * 16 * kTS + r * kTS + 4 * q) = hv;
                        *reinterpret_cast<f4 *>(bufH + (2 * u + 1) * 16 * kTS + r * kTS + 4 * q) = gpv[u];
                    }
                    wave_lds_fence();
#pragma unroll
                    for (int rho = 0; rho < 4; ++rho)
#pragma unroll
                        for (int u = 0; u < RH; ++u) ginz[r0 + u] = mfma4(a1x[rho], gpv[u][rho], ginz[r0 + u]);
#pragma unroll
                    for (int u = 0; u < RH; ++u)
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) {
                            const float hT = bufH[(2 * u) * 16 * kTS + (4 * ks + q) * kTS + r];
                            const float pT = bufH[(2 * u + 1) * 16 * kTS + (4 * ks + q) * kTS + r];
                            gW2 = mfma16(hT, goT[r0 + u][ks], gW2);
                            gW1 = mfma16(pT, inT[r0 + u][ks], gW1);
                        }
                }
                float *sb = slot + (size_t)(t % kFT) * 2 * 256 + lane * 4;
                *reinterpret_cast<f4 *>(sb) = gW1;
                *reinterpret_cast<f4 *>(sb + 256) = gW2;
                const bool last = (t + 1 == HT);
                if ((t + 1) % kFT == 0 || last) {
                    if (last) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { const float v = row16_sum(gb2[e]); if (r == 0) slot[kFT * 3 * 256 + q * 4 + e] = v; }
                    }
                    CSTAMP_ADD(s_db, t0);
                    __syncthreads();
                    const int tf = (t / kFT) * kFT, nfl4 = (t + 1 - tf) * 2 * 256 / 4;
                    f4 *dst = reinterpret_cast<f4 *>(gp + g.gDec + (size_t)tf * 2 * 256);
                    const f4 *s0 = reinterpret_cast<const f4 *>(lds);
                    for (int i = tid; i < nfl4; i += kWaves * 64) {
                        f4 v = s0[i];
#pragma unroll
                        for (int w = 1; w < kWaves; ++w) v += s0[w * (SLOT / 4) + i];
                        dst[i] = first ? v : dst[i] + v;
                    }
                    if (last && tid < 16) {
                        float v = 0.f;
#pragma unroll
                        for (int w = 0; w < kWaves; ++w) v += lds[w * SLOT + kFT * 3 * 256 + tid];
                        gp[g.gB2D + tid] = first ? v : gp[g.gB2D + tid] + v;
                    }
                    __syncthreads();
                    CSTAMP_ADD(s_dfl, t0);
                }
                a1 = na1; b1 = nb1; a2t = na2t; a1x = na1x;
                if constexpr (kSaveH) {
#pragma unroll
                    for (int rt = 0; rt < R; ++rt) hc[rt] = hn[rt];
                }
            }
        }
        CSTAMP_ADD(s_db, t0);
        // ---- gradient wrt mu / log_sigma ------------------------------------------------------------
        float gmu[R], gls[R];
        f4 gbh = f4{0.f, 0.f, 0.f, 0.f};
        float goTe[R][4], inTe[R][2][4];
#pragma unroll
        for (int rt = 0; rt < R; ++rt) {
            const float gz = reduce_scatter4(ginz[rt]);
            const float sc = valid[rt] ? inv_B : 0.f;
            gmu[rt] = fmaf(klw * sc, mu[rt], gz);
            gls[rt] = gz * er[rt] * 0.5f * el[rt] + klw * sc * (-0.5f) * (1.f - expf(ls[rt]));
            if (q >= g.lat) { gmu[rt] = 0.f; gls[rt] = 0.f; }
            gbh[0] += gmu[rt]; gbh[1] += gls[rt];
            transpose16(bufG, f4{gmu[rt], gls[rt], 0.f, 0.f}, lane, goTe[rt]);                    // out 4q: mu[q], 4q+1: ls[q]
            transpose16(bufI, f4{xr[rt][0], xr[rt][1], xr[rt][2], xr[rt][3]}, lane, inTe[rt][0]);   // x features
            transpose16(bufI + 16 * kTS, f4{cr[rt][0], q == 0 ? 1.f : 0.f, 0.f, 0.f}, lane, inTe[rt][1]);   // [c | 1]
        }
        CSTAMP_ADD(s_mid, t0);
        // ---- encoder backward ------------------------------------------------------------------------
        {
            const float *pA1 = wp + g.oA1E + lane * 4, *pB1 = wp + g.oB1E + q * 4, *pA2T = wp + g.oA2ET + lane * 4;
            f4 a10 = *opaque(pA1), a11 = *opaque(pA1 + 256), b1 = *opaque(pB1), a2t = *opaque(pA2T);
            f4 hc[kSaveH ? R : 1], hn[kSaveH ? R : 1];
            if constexpr (kSaveH) {
#pragma unroll
                for (int rt = 0; rt < R; ++rt) hc[rt] = *reinterpret_cast<const f4 *>(hsE + (size_t)rt * 256);
            }
            for (int t = 0; t < HT; ++t) {
                const int nx = t + 1 < HT ? t + 1 : t;
                f4 na10 = a10, na11 = a11, nb1 = b1;
                if constexpr (!kSaveH) {
                    na10 = *opaque(pA1 + (size_t)(2 * nx) * 256); na11 = *opaque(pA1 + (size_t)(2 * nx + 1) * 256); nb1 = *opaque(pB1 + nx * 16);
                } else {
#pragma unroll
                    for (int rt = 0; rt < R; ++rt) hn[rt] = *opaque(hsE + ((size_t)nx * R + rt) * 256);
                }
                const f4 na2t = *opaque(pA2T + (size_t)nx * 256);
                f4 gW1a = f4{0.f, 0.f, 0.f, 0.f}, gW1b = f4{0.f, 0.f, 0.f, 0.f}, gW2 = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r0 = 0; r0 < R; r0 += RH) {
                    wave_lds_fence();
#pragma unroll
                    for (int u = 0; u < RH; ++u) {
                        const int rt = r0 + u;
                        f4 hv;
                        if constexpr (kSaveH) hv = hc[rt];
                        else {
                            f4 acc = b1;
#pragma unroll
                            for (int kk = 0; kk < 4; ++kk) acc = mfma16(a10[kk], xr[rt][kk], acc);
                            acc = mfma16(a11[0], cr[rt][0], acc);
                            hv = tanh4(acc);
                        }
                        f4 gh = mfma16(a2t[0], gmu[rt], f4{0.f, 0.f, 0.f, 0.f});
                        gh = mfma16(a2t[1], gls[rt], gh);
                        const f4 gpv = gh * (1.0f - hv * hv);
                        *reinterpret_cast<f4 *>(bufH + (2 * u) * 16 * kTS + r * kTS + 4 * q) = hv;
                        *reinterpret_cast<f4 *>(bufH + (2 * u + 1) * 16 * kTS + r * kTS + 4 * q) = gpv;
                    }
                    wave_lds_fence();
#pragma unroll
                    for (int u = 0; u < RH; ++u)
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) {
                            const float hT = bufH[(2 * u) * 16 * kTS + (4 * ks + q) * kTS + r];
                            const float pT = bufH[(2 * u + 1) * 16 * kTS + (4 * ks + q) * kTS + r];
                            gW2 = mfma16(hT, goTe[r0 + u][ks], gW2);
                            gW1a = mfma16(pT, inTe[r0 + u][0][ks], gW1a);
                            gW1b = mfma16(pT, inTe[r0 + u][1][ks], gW1b);
                        }
                }
                float *sb = slot + (size_t)(t % kFT) * 3 * 256 + lane * 4;
                *reinterpret_cast<f4 *>(sb) = gW1a;
                *reinterpret_cast<f4 *>(sb + 256) = gW1b;
                *reinterpret_cast<f4 *>(sb + 512) = gW2;
                const bool last = (t + 1 == HT);
                if ((t + 1) % kFT == 0 || last) {
                    if (last) {
#pragma unroll
                        for (int e = 0; e < 2; ++e) { const float v = row16_sum(gbh[e]); if (r == 0) slot[kFT * 3 * 256 + 16 + q * 4 + e] = v; }
                    }
                    CSTAMP_ADD(s_eb, t0);
                    __syncthreads();
                    const int tf = (t / kFT) * kFT, nfl4 = (t + 1 - tf) * 3 * 256 / 4;
                    f4 *dst = reinterpret_cast<f4 *>(gp + g.gEnc + (size_t)tf * 3 * 256);
                    const f4 *s0 = reinterpret_cast<const f4 *>(lds);
                    for (int i = tid; i < nfl4; i += kWaves * 64) {
                        f4 v = s0[i];
#pragma unroll
                        for (int w = 1; w < kWaves; ++w) v += s0[w * (SLOT / 4) + i];
                        dst[i] = first ? v : dst[i] + v;
                    }
                    if (last && tid < 16) {
                        float v = 0.f;
#pragma unroll
                        for (int w = 0; w < kWaves; ++w) v += lds[w * SLOT + kFT * 3 * 256 + 16 + tid];
                        gp[g.gBH + tid] = first ? v : gp[g.gBH + tid] + v;
                    }
                    __syncthreads();
                    CSTAMP_ADD(s_efl, t0);
                }
                a10 = na10; a11 = na11; b1 = nb1; a2t = na2t;
                if constexpr (kSaveH) {
#pragma unroll
                    for (int rt = 0; rt < R; ++rt) hc[rt] = hn[rt];
                }
            }
        }
        CSTAMP_ADD(s_eb, t0);
        first = false;
    }
    if (lane == 0) losspart[blockIdx.x * kWaves + wave] = wave_sum;
#ifdef CVAE_STAMP
    {
        unsigned long long tk1; CSTAMP(tk1);
        if (lane == 0 && (blockIdx.x == 0 || blockIdx.x == 100))
            printf("CSTAMP wg %d wave %d total %llu load %llu encfwd %llu decfwd %llu loss %llu decbwd %llu decflush %llu mid %llu encbwd %llu encflush %llu\n",
                   (int)blockIdx.x, wave, tk1 - tk0, s_ld, s_ef, s_df, s_loss, s_db, s_dfl, s_mid, s_eb, s_efl);
    }
#endif
}

// ---- encoder / decoder alone (cvae_encode, cvae_decode): the forward blocks of the step, no LDS -------------
template <bool ENCODE>
__global__ void __launch_bounds__(kWaves * 64) __attribute__((amdgpu_waves_per_eu(RNVP_WPE, RNVP_WPE)))
k_cvae_mfma_mlp(CG g, const float *__restrict__ wp, const float *__restrict__ in, const float *__restrict__ c, int64_t n,
                float *__restrict__ out0, float *__restrict__ out1) {
    constexpr int R = kR;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, r = lane & 15;
    const int HT = g.HT;
    const int64_t rows_per_wg = (int64_t)kWaves * R * 16;
    const int64_t ngroups = (n + rows_per_wg - 1) / rows_per_wg;
    for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int64_t base = grp * rows_per_wg + (int64_t)wave * R * 16;
        if (base >= n) continue;
        float xr[R][4], cr[R][1];
        bool valid[R];
#pragma unroll
        for (int rt = 0; rt < R; ++rt) {
            const int64_t row = base + rt * 16 + r;
            valid[rt] = row < n;
            const int64_t src = valid[rt] ? row : 0;
            if (ENCODE) {
                mfma::load_row<2, 1>(in, c, src, g.d, g.c, (g.d == 16) && (g.c == 4), q, xr[rt], cr[rt]);
            } else {
                xr[rt][0] = (q < g.lat) ? in[src * g.lat + q] : 0.f;                          // z[q]
                cr[rt][0] = (q < g.c) ? c[src * g.c + q] : 0.f;
            }
        }
        if (ENCODE) {
            const f4 bh = *reinterpret_cast<const f4 *>(wp + g.oBH + q * 4);
            f4 outE[R][2];
#pragma unroll
            for (int rt = 0; rt < R; ++rt) { outE[rt][0] = f4{0.f, 0.f, 0.f, 0.f}; outE[rt][1] = f4{0.f, 0.f, 0.f, 0.f}; }
            const float *pA1 = wp + g.oA1E + lane * 4, *pB1 = wp + g.oB1E + q * 4, *pA2 = wp + g.oA2E + lane * 4;
            // the fragments of hidden tile t + 1 are requested before tile t is multiplied (one wave per SIMD: nothing
            // else would cover the L2 round trip)
            f4 a10 = *opaque(pA1), a11 = *opaque(pA1 + 256), b1 = *opaque(pB1), a20 = *opaque(pA2), a21 = *opaque(pA2 + 256);
            for (int t = 0; t < HT; ++t) {
                const int nx = t + 1 < HT ? t + 1 : t;
                const f4 na10 = *opaque(pA1 + (size_t)(2 * nx) * 256), na11 = *opaque(pA1 + (size_t)(2 * nx + 1) * 256);
                const f4 nb1 = *opaque(pB1 + nx * 16);
                const f4 na20 = *opaque(pA2 + (size_t)(2 * nx) * 256), na21 = *opaque(pA2 + (size_t)(2 * nx + 1) * 256);
                f4 hv[R];
#pragma unroll
                for (int rt = 0; rt < R; ++rt) {
                    f4 acc = b1;
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) acc = mfma16(a10[kk], xr[rt][kk], acc);
                    acc = mfma16(a11[0], cr[rt][0], acc);
                    hv[rt] = tanh4(acc);
                }
#pragma unroll
                for (int rho = 0; rho < 4; ++rho)
#pragma unroll
                    for (int rt = 0; rt < R; ++rt) {
                        outE[rt][0] = mfma4(a20[rho], hv[rt][rho], outE[rt][0]);
                        outE[rt][1] = mfma4(a21[rho], hv[rt][rho], outE[rt][1]);
                    }
                a10 = na10; a11 = na11; b1 = nb1; a20 = na20; a21 = na21;
            }
#pragma unroll
            for (int rt = 0; rt < R; ++rt) {
                const float mu = reduce_scatter4(outE[rt][0]) + bh[0], ls = reduce_scatter4(outE[rt][1]) + bh[1];
                const int64_t row = base + rt * 16 + r;
                if (valid[rt] && q < g.lat) { out0[row * g.lat + q] = mu; out1[row * g.lat + q] = ls; }
            }
        } else {
            const f4 b2d = *reinterpret_cast<const f4 *>(wp + g.oB2D + q * 4);
            f4 xrec[R];
#pragma unroll
            for (int rt = 0; rt < R; ++rt) xrec[rt] = b2d;
            const float *pA1 = wp + g.oA1D + lane * 4, *pB1 = wp + g.oB1D + q * 4, *pA2 = wp + g.oA2D + lane * 4;
            f4 a1 = *opaque(pA1), b1 = *opaque(pB1), a2 = *opaque(pA2);
            for (int t = 0; t < HT; ++t) {
                const int nx = t + 1 < HT ? t + 1 : t;
                const f4 na1 = *opaque(pA1 + (size_t)nx * 256), nb1 = *opaque(pB1 + nx * 16), na2 = *opaque(pA2 + (size_t)nx * 256);
                f4 hv[R];
#pragma unroll
                for (int rt = 0; rt < R; ++rt) {
                    f4 acc = mfma16(a1[0], xr[rt][0], b1);
                    acc = mfma16(a1[1], cr[rt][0], acc);
                    hv[rt] = tanh4(acc);
                }
#pragma unroll
                for (int rho = 0; rho < 4; ++rho)
#pragma unroll
                    for (int rt = 0; rt < R; ++rt) xrec[rt] = mfma16(a2[rho], hv[rt][rho], xrec[rt]);
                a1 = na1; b1 = nb1; a2 = na2;
            }
#pragma unroll
            for (int rt = 0; rt < R; ++rt) {
                const int64_t row = base + rt * 16 + r;
                const float v[4] = {xrec[rt][0], xrec[rt][1], xrec[rt][2], xrec[rt][3]};
                if (valid[rt]) mfma::store_row<2>(out0, row, g.d, g.d == 16, q, v);
            }
        }
    }
}

// D-layout location of (row i of the 16-row M tile = hid & 15, column j) inside a 256-float block
__device__ __forceinline__ int dloc(int hid, int col) { const int i = hid & 15; return (16 * (i >> 2) + col) * 4 + (i & 3); }

// ---- stage 2: ONE launch behind the step kernel (round 3 ran three: segment sums, scatter + Adam, the next step's pack) ----
// Workgroup t * 5 + kind owns one 256-float block of hidden tile t's gradient record -- kind 0: W1e, x columns; 1: W1e, condition
// columns + b1e; 2: the two heads' weights; 3: W1d + b1d; 4: W2d -- workgroup 5 HT the head and output biases, the last one the
// loss.  Stages as k_train_finish (rnvp_mfma_train.hip): sum the block over the G per-workgroup partials in a fixed order
// (kFinSum), scatter into the flat gradient, Adam on exactly those parameters (kFinAdam), and re-pack the fragment slots that
// hold them from the updated values kept in LDS (kFinPack), so that the next batch's step kernel needs no pack launch.
constexpr int kFinSum = 1, kFinAdam = 2, kFinPack = 4;
constexpr int kFinThreads = 512;

// pack_value's parameter source for one workgroup's LDS copy (layouts: the `e` order of k_cvae_finish)
struct BlockCvae {
    const float *pw;
    int t, d, c, lat;
    __device__ float w1e(int hid, int col) const { return col < d ? pw[(hid - 16 * t) * d + col] : pw[(hid - 16 * t) * (c + 1) + col - d]; }
    __device__ float b1e(int hid) const { return pw[(hid - 16 * t) * (c + 1) + c]; }
    __device__ float whead(int og, int i, int hid) const { return pw[(og * lat + i) * 16 + hid - 16 * t]; }
    __device__ float bhead(int og, int i) const { return pw[og * lat + i]; }
    __device__ float w1d(int hid, int col) const { return pw[(hid - 16 * t) * (lat + c + 1) + col]; }
    __device__ float b1d(int hid) const { return pw[(hid - 16 * t) * (lat + c + 1) + lat + c]; }
    __device__ float w2d(int out, int hid) const { return pw[out * 16 + hid - 16 * t]; }
    __device__ float b2d(int j) const { return pw[2 * lat + j]; }
};

__global__ void __launch_bounds__(kFinThreads)
k_cvae_finish(CG g, int mode, const float *__restrict__ gpart, int G, const float *__restrict__ losspart, int nloss, float inv_B,
              float *loss, float *grad, float *params, float *adam_m, float *adam_v, AdamK adam, float *packed) {
    __shared__ __attribute__((aligned(16))) float rec[256];
    __shared__ f4 red[kFinThreads];
    __shared__ float pw[16 * 21];                 // the widest block: 16 hidden units x (d <= 16 | lat + c + 1 <= 9 | ...)
    const int tix = threadIdx.x, b = blockIdx.x, HT = g.HT;
    if (b == 5 * HT + 1) {
        if (loss && tix < 64) {
            float a = 0.f;
            for (int i = tix; i < nloss; i += 64) a += losspart[i];
            for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off);
            if (tix == 0) loss[0] = a * inv_B;
        }
        return;
    }
    if (!(mode & kFinSum)) return;                // (cvae_loss_grad without gradients: only the loss block has work)
    const bool misc = b == 5 * HT;
    const int t = misc ? 0 : b / 5, kind = misc ? 5 : b - 5 * t;
    const int rec_off = misc ? g.gBH : (kind < 3 ? g.gEnc + (t * 3 + kind) * 256 : g.gDec + (t * 2 + kind - 3) * 256);
    const int rec_n = misc ? 32 : 256;
    const int d = g.d, c = g.c, lat = g.lat, h = g.h, ne = d + c, nd = lat + c;
    // this workgroup's parameters (at most one per thread): flat index, location inside the block; loads issued ahead of the sums
    const int npar = kind == 0 ? 16 * d : kind == 1 ? 16 * (c + 1) : kind == 2 ? 2 * lat * 16 : kind == 3 ? 16 * (nd + 1)
                     : kind == 4 ? d * 16 : 2 * lat + d;
    int p = -1, loc = 0;
    if (tix < npar) {
        const int e = tix;
        int hid = 0;
        if (kind == 0) { const int i = e / d, col = e - i * d; hid = 16 * t + i; p = g.fW1e + hid * ne + col; loc = dloc(i, col); }
        else if (kind == 1) {
            const int i = e / (c + 1), cc = e - i * (c + 1); hid = 16 * t + i;
            if (cc < c) { p = g.fW1e + hid * ne + d + cc; loc = dloc(i, 4 * cc); } else { p = g.fb1e + hid; loc = dloc(i, 1); }
        } else if (kind == 2) {
            const int og = e / (lat * 16), rem = e - og * lat * 16, li = rem >> 4, i = rem & 15; hid = 16 * t + i;
            p = (og ? g.fWls : g.fWmu) + li * h + hid; loc = dloc(i, 4 * li + og);
        } else if (kind == 3) {
            const int i = e / (nd + 1), col = e - i * (nd + 1); hid = 16 * t + i;
            if (col < nd) { p = g.fW1d + hid * nd + col; loc = dloc(i, col < lat ? 4 * col : 4 * (col - lat) + 1); }
            else { p = g.fb1d + hid; loc = dloc(i, 2); }
        } else if (kind == 4) {
            const int out = e >> 4, i = e & 15; hid = 16 * t + i;
            p = g.fW2d + out * h + hid; loc = dloc(i, out);
        } else if (e < 2 * lat) {
            const int og = e / lat, li = e - og * lat;
            p = (og ? g.fbls : g.fbmu) + li; loc = li * 4 + og;
        } else {
            p = g.fb2d + e - 2 * lat; loc = 16 + e - 2 * lat;
        }
        if (hid >= h) p = -1;
    }
    float pv = 0.f, mm = 0.f, vv = 0.f;
    if (p >= 0 && (mode & kFinAdam)) { pv = params[p]; mm = adam_m[p]; vv = adam_v[p]; }
    {
        const int nf4 = rec_n / 4, nsub = kFinThreads / nf4;
        const int col = tix % nf4, sub = tix / nf4;
        f4 acc = f4{0.f, 0.f, 0.f, 0.f};
        const f4 *src = reinterpret_cast<const f4 *>(gpart + rec_off) + col;
        const size_t stride4 = (size_t)g.gfloats / 4;
        int bb = sub;
        for (; bb + 7 * nsub < G; bb += 8 * nsub) {
            f4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(src + (size_t)(bb + u * nsub) * stride4);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; bb < G; bb += nsub) acc += __builtin_nontemporal_load(src + (size_t)bb * stride4);
        red[tix] = acc;
        __syncthreads();
        if (tix < nf4) {
            f4 a = red[tix];
            for (int s2 = 1; s2 < nsub; ++s2) a += red[s2 * nf4 + tix];
            *reinterpret_cast<f4 *>(rec + 4 * tix) = a;
        }
        __syncthreads();
    }
    if (tix < npar) {
        if (p >= 0) {
            const float a = rec[loc];
            grad[p] = a;
            if (mode & kFinAdam) {
                adam_one(pv, a, mm, vv, adam);
                params[p] = pv; adam_m[p] = mm; adam_v[p] = vv;
            }
        }
        pw[tix] = pv;
    }
    if (!(mode & kFinPack)) return;
    __syncthreads();
    const BlockCvae src{pw, t, d, c, lat};
    // the packed arrays whose slots of tile t hold exactly this workgroup's parameters: (offset, floats) pairs
    int off[3] = {0, 0, 0}, cnt[3] = {0, 0, 0};
    if (kind == 0) { off[0] = g.oA1E + (t * 2) * 256; cnt[0] = 256; }
    else if (kind == 1) { off[0] = g.oA1E + (t * 2 + 1) * 256; cnt[0] = 256; off[1] = g.oB1E + t * 16; cnt[1] = 16; }
    else if (kind == 2) { off[0] = g.oA2E + t * 512; cnt[0] = 512; off[1] = g.oA2ET + t * 256; cnt[1] = 256; }
    else if (kind == 3) { off[0] = g.oA1D + t * 256; cnt[0] = 256; off[1] = g.oB1D + t * 16; cnt[1] = 16; off[2] = g.oA1DX + t * 256; cnt[2] = 256; }
    else if (kind == 4) { off[0] = g.oA2D + t * 256; cnt[0] = 256; off[1] = g.oA2DT + t * 256; cnt[1] = 256; }
    else { off[0] = g.oBH; cnt[0] = 32; }         // BH and B2D are adjacent
#pragma unroll
    for (int a = 0; a < 3; ++a)
        for (int s2 = tix; s2 < cnt[a]; s2 += kFinThreads) packed[off[a] + s2] = pack_value(g, off[a] + s2, src);
}

size_t lds_bytes() { return ((size_t)kWaves * (kFT * 3 * 256 + 32) + (size_t)kWaves * 7 * 16 * kTS) * sizeof(float); }

}  // namespace

bool supported(const cvae_shape *s) {
    return s && s->n_hidden == 1 && s->act == 0 && s->d >= 1 && s->d <= 16 && s->c >= 0 && s->c <= 4 && s->lat >= 1 &&
           s->lat <= 4 && s->hidden[0] >= 1;
}

// CVAE_SAVE_H: per workgroup 2 nets x HT tiles x (waves x row tiles) KiB, for the workgroups max_rows rows need (at most kMaxGrid)
static size_t hsave_bytes(const CG &g, int64_t max_rows) {
    if (!kSaveH) return 0;
    const int64_t rows_per_wg = (int64_t)kWaves * kR * 16;
    int64_t wg = (max_rows + rows_per_wg - 1) / rows_per_wg;
    if (wg < 1) wg = 1;
    if (wg > kMaxGrid) wg = kMaxGrid;
    return align_up((size_t)wg * kWaves * 2 * g.HT * kR * 256 * sizeof(float), 256);
}

size_t workspace_bytes(const cvae_shape *s, int64_t max_rows) {
    const CG g = make_cg(s);
    return align_up((size_t)g.packed_floats * 4, 256) + align_up((size_t)kMaxGrid * g.gfloats * 4, 256) +
           align_up((size_t)kMaxGrid * kWaves * 4, 256) + hsave_bytes(g, max_rows);
}

// [pack] -> step kernel (per-workgroup partial gradients) -> k_cvae_finish (sum, scatter [, Adam [, re-pack]])
static int loss_grad_impl(hipStream_t st, const cvae_shape *s, const float *params, const float *x, const float *c,
                          const int64_t *row_index, const float *eps, int64_t n, float inv_B, float klw, float *grad_out,
                          float *loss_out, void *ws, size_t ws_bytes, float *adam_p, float *adam_m, float *adam_v, AdamK adam,
                          bool packed_valid, bool pack_next) {
    if (!ws || ws_bytes < workspace_bytes(s, n)) return RNVP_EWORKSPACE;
    const CG g = make_cg(s);
    char *w = static_cast<char *>(ws);
    float *packed = reinterpret_cast<float *>(w); w += align_up((size_t)g.packed_floats * 4, 256);
    float *gpart = reinterpret_cast<float *>(w); w += align_up((size_t)kMaxGrid * g.gfloats * 4, 256);
    float *losspart = reinterpret_cast<float *>(w); w += align_up((size_t)kMaxGrid * kWaves * 4, 256);
    float *hsave = reinterpret_cast<float *>(w);
    if (!packed_valid) {
        hipLaunchKernelGGL(k_pack, dim3((g.packed_floats + 255) / 256), dim3(256), 0, st, g, params, packed);
        RNVP_HIP_TRY(hipGetLastError());
    }
    static std::atomic<uint64_t> attr{0};
    {
        const int arc = allow_big_lds(reinterpret_cast<const void *>(k_cvae_mfma), 160 * 1024, attr);
        if (arc) return arc;
    }
    const int64_t rows_per_wg = (int64_t)kWaves * kR * 16, ngroups = (n + rows_per_wg - 1) / rows_per_wg;
    const int grid = (int)(ngroups < kMaxGrid ? ngroups : kMaxGrid);
    {
        KernelTimer timer(st, RNVP_PROFILE_TRAIN);
        hipLaunchKernelGGL(k_cvae_mfma, dim3(grid), dim3(kWaves * 64), lds_bytes(), st, g, packed, x, c, row_index, eps, n,
                           inv_B, klw, gpart, losspart, grad_out ? 1 : 0, hsave);
    }
    RNVP_HIP_TRY(hipGetLastError());
    const int mode = grad_out ? (kFinSum | (adam_p ? kFinAdam : 0) | (adam_p && pack_next ? kFinPack : 0)) : 0;
    hipLaunchKernelGGL(k_cvae_finish, dim3(5 * g.HT + 2), dim3(kFinThreads), 0, st, g, mode, gpart, grid, losspart, grid * kWaves, inv_B,
                       loss_out, grad_out, adam_p, adam_m, adam_v, adam, packed);
    RNVP_HIP_TRY(hipGetLastError());
    return RNVP_OK;
}

int loss_grad(hipStream_t st, const cvae_shape *s, const float *params, const float *x, const float *c,
              const int64_t *row_index, const float *eps, int64_t n, float inv_B, float klw, float *grad_out,
              float *loss_out, void *ws, size_t ws_bytes) {
    return loss_grad_impl(st, s, params, x, c, row_index, eps, n, inv_B, klw, grad_out, loss_out, ws, ws_bytes, nullptr, nullptr,
                          nullptr, AdamK{}, false, false);
}

// loss + gradient + Adam (+ the next step's re-pack) with the optimizer fused into the finish kernel: the same arithmetic as
// cvae_loss_grad + rnvp_adam_step, bit for bit.  packed_valid / pack_next: as rnvp::mfma::train_step (cvae_fit_epoch's loop)
int train_step(hipStream_t st, const cvae_shape *s, float *params, const float *x, const float *c, const int64_t *row_index,
               const float *eps, int64_t n, float inv_B, float klw, float *grad_buf, float *loss_out, float *exp_avg,
               float *exp_avg_sq, const AdamK &adam, void *ws, size_t ws_bytes, bool packed_valid, bool pack_next) {
    return loss_grad_impl(st, s, params, x, c, row_index, eps, n, inv_B, klw, grad_buf, loss_out, ws, ws_bytes, params, exp_avg,
                          exp_avg_sq, adam, packed_valid, pack_next);
}

// encoder (mu, log_sigma) or decoder (x_rec) alone on the MFMA blocks; the packed weights go to the caller's workspace
int forward(hipStream_t st, const cvae_shape *s, const float *params, bool encode, const float *in, const float *c,
            int64_t n, float *out0, float *out1, void *ws, size_t ws_bytes) {
    const CG g = make_cg(s);
    if (!ws || ws_bytes < align_up((size_t)g.packed_floats * 4, 256)) return RNVP_EWORKSPACE;
    float *packed = static_cast<float *>(ws);
    hipLaunchKernelGGL(k_pack, dim3((g.packed_floats + 255) / 256), dim3(256), 0, st, g, params, packed);
    RNVP_HIP_TRY(hipGetLastError());
    const int64_t rows_per_wg = (int64_t)kWaves * kR * 16, ngroups = (n + rows_per_wg - 1) / rows_per_wg;
    const int grid = (int)(ngroups < 2048 ? ngroups : 2048);
    if (encode)
        hipLaunchKernelGGL(k_cvae_mfma_mlp<true>, dim3(grid), dim3(kWaves * 64), 0, st, g, packed, in, c, n, out0, out1);
    else
        hipLaunchKernelGGL(k_cvae_mfma_mlp<false>, dim3(grid), dim3(kWaves * 64), 0, st, g, packed, in, c, n, out0, out1);
    RNVP_HIP_TRY(hipGetLastError());
    return RNVP_OK;
}

}  // namespace cvae_mfma
}  // namespace rnvp
