// rnvp_lmm.hip -- "lmm": any-shape RealNVP coupling stack on f32 MFMA with LDS-resident activations (gfx950).
//
// Serves what the register-chained MFMA path (rnvp_mfma*.hip, rnvp_bx3.hip) does not specialise -- several hidden
// layers (the reference's docstring example is hidden=(10, 20, 15): /root/reference/probaforms/models/realnvp.py:22-38),
// user-supplied masks (realnvp.py:65-68), d > 64, cdim > 16 -- which the one-thread-per-row VALU kernels of
// rnvp_generic.hip served at 0.3 M rows/s (fit, hidden=(128, 128)).  Design:
//   * one WORKGROUP of 4 waves owns one tile of 16 rows for the whole stack; its activations live in one LDS image
//     [feature][17]; the waves split every Linear's out tiles (and the elementwise passes' features) among themselves,
//     one workgroup barrier per Linear -- the image, not the registers, limits how many tiles a CU holds, so four
//     waves per image is what gives the SIMDs something to overlap the fragment loads with;
//   * every Linear is computed transposed, out^T[out x rows] = W[out x in] . act^T[in x rows], as
//     v_mfma_f32_16x16x4_f32: A = weight fragments pre-packed per call into lane order (the layer's mask folded into
//     the first Linear's x columns, so [x*mask || c] is never formed), B = the LDS image (ds_read_b32, conflict free),
//     the accumulator (bias added, activation applied) written back to the LDS image of the next Linear;
//   * forward / inverse: x, c read once, z / log-prob written once per row;
//   * training: the same kernel runs forward, then per layer recomputes both nets and chains the INPUT gradients through
//     W^T fragments (the hand-derived backward of SURVEY.md 3.3); the WEIGHT gradients contract over rows, so each
//     tile dumps, per Linear, its input activations (+ a ones column for the bias) and the pre-activation gradients as
//     [feature][16 rows] tiles, and k_lmm_wgrad forms dW = gP^T . act per 2 x 2 block of tile pairs over fixed row
//     splits; k_lmm_reduce adds the splits in order and scatters into the reference's flat parameter order.  A big
//     batch goes through in row chunks (the dumps stay under ~1 GiB).  No float atomics: bitwise reproducible.
#include "rnvp_common.h"
#include "rnvp_generic_net.h"
#include "rnvp_lmm.h"

namespace rnvp {
namespace lmm {
namespace {

using f4 = __attribute__((ext_vector_type(4))) float;
constexpr int RS = 17;                 // row stride of the LDS image [feature][RS]: 16 rows + 1 (bank spread for the row-contraction reads)
constexpr int kMaxGrid = 4096;
constexpr int kSplits = 64;            // row splits of the weight-gradient pass
constexpr int kW = 4;                  // waves per workgroup (all on the same 16-row tile)

__device__ __forceinline__ f4 mfma16(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
// ---- weight packing -----------------------------------------------------------------------------------------------
// Fragment value for out tile m, k-step ks, lane (q, i): forward W_k[16m + i][4ks + q]; transposed W_k[4ks + q][16m + i]
// (for k == 0 only the x columns: C gets no gradient).  First Linear: x column j is multiplied by mask[l][j].
// Layout of one Linear's fragments: out tiles in blocks of MB = 2 (then one of 1; MB = 1 throughout for fewer than 8 tiles), a block stored
// [group of 4 k-steps][tile in block][lane][k-step in group] with the k-steps padded to a multiple of 4 (zeros), so that the
// kernel's group of 4 k-steps x MB tiles is MB 16-byte loads at constant offsets from one base (round 4: one dwordx4 per tile and
// group instead of four dword loads -- a quarter of the vector-memory instructions, whole 1-KiB lines through the L1).  Blocks go
// round-robin to the waves.
// tiles per block of a Linear with MT out tiles: 2 where that still gives every wave a block, else 1
__host__ __device__ __forceinline__ int block_tiles(int MT) { return MT >= 2 * kW ? 2 : 1; }

__device__ __forceinline__ void block_decode(int o, int MT, int KSp, int *m, int *ks, int *lane) {
    const int per = KSp * 64;
    int MB = block_tiles(MT), m0, oo;
    const int nb = MT / MB;                       // full blocks; a last single tile follows when MB == 2 and MT is odd
    if (o < nb * MB * per) { m0 = MB * (o / (MB * per)); oo = o % (MB * per); }
    else { m0 = nb * MB; oo = o - nb * MB * per; MB = 1; }
    // inside a block: [group of 4 k-steps][tile][lane][k-step in group] -- a lane's four k-steps of one tile are ONE 16-byte load
    *lane = (oo >> 2) & 63;
    const int rest = oo >> 8;
    *m = m0 + rest % MB;
    *ks = 4 * (rest / MB) + (oo & 3);
}

__global__ void __launch_bounds__(256)
k_lmm_pack(KShape s, LGeo g, const float *__restrict__ params, const uint8_t *__restrict__ masks, float *__restrict__ packed) {
    const int64_t total = (int64_t)g.nnets * g.net_floats;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int ln = (int)(t / g.net_floats), o = (int)(t - (int64_t)ln * g.net_floats);
        const int l = ln >> 1;
        const float *pn = params + (size_t)ln * s.npn;
        float v = 0.f;
        for (int k = 0; k < g.nlin; ++k) {
            const int nin = g.nin[k], nout = g.nout[k];
            const float *W = pn + s.woff[k];
            int m, ks, lane;
            if (o >= g.offF[k] && o < g.offF[k] + g.MT[k] * g.KS[k] * 64) {
                block_decode(o - g.offF[k], g.MT[k], g.KS[k], &m, &ks, &lane);
                const int row = 16 * m + (lane & 15), col = 4 * ks + (lane >> 4);
                if (row < nout && col < nin) v = W[row * nin + col] * ((masks && k == 0 && col < s.d) ? (float)masks[l * s.d + col] : 1.f);
                break;
            }
            if (o >= g.offT[k] && o < g.offT[k] + g.MTt[k] * g.KSt[k] * 64) {
                block_decode(o - g.offT[k], g.MTt[k], g.KSt[k], &m, &ks, &lane);
                const int out = 4 * ks + (lane >> 4), in = 16 * m + (lane & 15);
                const int nin_eff = k == 0 ? s.d + (s.gcw ? s.c : 0) : nin;
                if (out < nout && in < nin_eff) v = W[out * nin + in] * ((masks && k == 0 && in < s.d) ? (float)masks[l * s.d + in] : 1.f);
                break;
            }
        }
        packed[t] = v;
    }
}

// out^T[nout x 16 rows] = act(W . in^T + b) for the MB out tiles of one fragment block; `in` / `out` are LDS images
// [feature][RS].  ACT: -1 none, RNVP_ACT_TANH, RNVP_ACT_RELU.  ACCUM: add into `out` (input gradients of the two nets).
// The instruction budget matters more than anything here -- an f32 MFMA and VALU work do not overlap on a SIMD, and
// a wave's LDS image leaves room for one or two waves per SIMD only -- so the group loop is 4 k-steps x MB tiles of
// loads at constant offsets from two bases (no clamps, no guards: the fragments are zero-padded, and an LDS row past
// the real inputs holds finite stale data that those zeros cancel), MB independent accumulator chains, and the next
// group's operands in flight while the current one multiplies.
template <bool ACCUM, int MB, int ACT>
__device__ __forceinline__ void linear_mb(const float *__restrict__ fblk, int m0, int KSp, int nout, const float *in,
                                          float *out, const float *__restrict__ bias, int lane) {
    const int q = lane >> 4, r = lane & 15;
    f4 acc[MB];
#pragma unroll
    for (int t = 0; t < MB; ++t) {
        acc[t] = f4{0.f, 0.f, 0.f, 0.f};
        if (bias) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const int o = 16 * (m0 + t) + 4 * q + e; acc[t][e] = o < nout ? bias[o] : 0.f; }
        }
    }
    const float *fa = fblk + lane * 4;             // + ((ks / 4) * MB + t) * 256: a lane's four k-steps of tile t
    const float *fb = in + q * RS + r;             // + (ks + u) * 4 * RS
    f4 a0[MB], a1[MB], a2[MB];
    float b0[4], b1[4], b2[4];
    auto fetch = [&](f4 (&a)[MB], float (&b)[4], int ks) {
        const float *pa = fa + ks * MB * 64, *pb = fb + ks * 4 * RS;
#pragma unroll
        for (int t = 0; t < MB; ++t) a[t] = *reinterpret_cast<const f4 *>(pa + t * 256);
#pragma unroll
        for (int u = 0; u < 4; ++u) b[u] = pb[u * 4 * RS];
    };
    auto mul = [&](const f4 (&a)[MB], const float (&b)[4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < MB; ++t) acc[t] = mfma16(a[t][u], b[u], acc[t]);
    };
    // three named operand sets rotate: the fragments of two groups (2 x 16 MFMAs ~ 1000 cycles) are in flight while
    // one multiplies -- an L2 hit under load takes about that long, and the wave's LDS image, not its registers,
    // limits the occupancy, so nothing else would hide it
    // The look-ahead requests are UNCONDITIONAL (past the end they re-request the last group): a load under a branch makes the
    // compiler's s_waitcnt insertion wait for ALL outstanding loads at the join -- i.e. for the group just requested -- and the
    // sched_barriers keep the machine scheduler from sinking the requests down to their uses (both seen in the ISA, rnvp_lmm64.hip)
    const int last = KSp - 4;
    auto ahead = [&](int ks) { return ks < last ? ks : last; };
    fetch(a0, b0, 0);
    fetch(a1, b1, ahead(4));
    for (int ks = 0; ks < KSp; ks += 12) {
        fetch(a2, b2, ahead(ks + 8));
        __builtin_amdgcn_sched_barrier(0);
        mul(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        if (ks + 4 >= KSp) break;
        fetch(a0, b0, ahead(ks + 12));
        __builtin_amdgcn_sched_barrier(0);
        mul(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        if (ks + 8 >= KSp) break;
        fetch(a1, b1, ahead(ks + 16));
        __builtin_amdgcn_sched_barrier(0);
        mul(a2, b2);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int t = 0; t < MB; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int o = 16 * (m0 + t) + 4 * q + e;
            if (o < nout) {
                float v = acc[t][e];
                if (ACT >= 0) v = act_fwd(v, ACT);
                if (ACCUM) out[o * RS + r] += v; else out[o * RS + r] = v;
            }
        }
}

template <bool ACCUM, int ACT>
__device__ __forceinline__ void linear_act(const float *__restrict__ frag, int MT, int KSp, int nout, const float *in, float *out,
                                           const float *__restrict__ bias, int lane, int wave) {
    if (block_tiles(MT) == 2) {
        const int n2 = MT >> 1;
        for (int b = wave; b < n2; b += kW) linear_mb<ACCUM, 2, ACT>(frag + (size_t)b * 2 * KSp * 64, 2 * b, KSp, nout, in, out, bias, lane);
        if ((MT & 1) && wave == n2 % kW) linear_mb<ACCUM, 1, ACT>(frag + (size_t)n2 * 2 * KSp * 64, 2 * n2, KSp, nout, in, out, bias, lane);
    } else {
        for (int b = wave; b < MT; b += kW) linear_mb<ACCUM, 1, ACT>(frag + (size_t)b * KSp * 64, b, KSp, nout, in, out, bias, lane);
    }
}

template <bool ACCUM>
__device__ __forceinline__ void linear(const float *__restrict__ frag, int MT, int KSp, int nin, int nout, const float *in,
                                       float *out, const float *__restrict__ bias, int act, int lane, int wave) {
    (void)nin;
    if (act < 0) linear_act<ACCUM, -1>(frag, MT, KSp, nout, in, out, bias, lane, wave);
    else if (act == RNVP_ACT_TANH) linear_act<ACCUM, RNVP_ACT_TANH>(frag, MT, KSp, nout, in, out, bias, lane, wave);
    else linear_act<ACCUM, RNVP_ACT_RELU>(frag, MT, KSp, nout, in, out, bias, lane, wave);
}

// one s or t net, forward: Linear k reads buffer k-1's output; hidden activations go to hbuf + (KEEP ? running
// offset : ping-pong between hbuf and hbuf2); the last Linear writes `out` [d]
template <bool KEEP>
__device__ __forceinline__ void net_fwd(const float *__restrict__ pk, const float *__restrict__ pn, const KShape &s,
                                        const LGeo &g, const float *xc, float *hbuf, float *hbuf2, float *out, int lane, int wave) {
    const float *cur = xc;
    float *dst = hbuf;
    for (int k = 0; k < g.nlin; ++k) {
        const bool last = k == g.nlin - 1;
        float *ob = last ? out : dst;
        linear<false>(pk + g.offF[k], g.MT[k], g.KS[k], g.nin[k], g.nout[k], cur, ob, pn + s.boff[k], last ? -1 : s.act, lane, wave);
        __syncthreads();
        cur = ob;
        if (!last) {
            if (KEEP) dst += g.nout[k] * RS;
            else dst = (dst == hbuf) ? hbuf2 : hbuf;
        }
    }
}

// rows of one tile <-> LDS image; element e = lane + 64 t of the 16 x w block (coalesced when the rows are contiguous)
__device__ __forceinline__ void load_tile(const float *__restrict__ src, const int64_t *__restrict__ row_index, int64_t base,
                                          int64_t n, int w, float *img, int tid) {
    for (int e = tid; e < 16 * w; e += 64 * kW) {
        const int rr = e / w, j = e - rr * w;
        const int64_t row = base + rr;
        float v = 0.f;
        if (row < n) v = src[(row_index ? row_index[row] : row) * w + j];
        img[j * RS + rr] = v;
    }
}

// cross-wave sum of a per-row value: every lane holds a partial for row r (already reduced over the lane groups q);
// `red` is a kW x 16 scratch in LDS; the result is returned to the lanes of wave 0 (other waves get garbage)
__device__ __forceinline__ float wg_row_sum(float v, float *red, int lane, int wave) {
    const int q = lane >> 4, r = lane & 15;
    if (q == 0) red[wave * 16 + r] = v;
    __syncthreads();
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < kW; ++w) a += red[w * 16 + r];
    __syncthreads();
    return a;
}

// ---- forward (+ log-det + prior) / inverse ----------------------------------------------------------------------------
template <bool INVERSE>
__global__ void __launch_bounds__(64 * kW)
k_lmm_flow(KShape s, LGeo g, const float *__restrict__ packed, const float *__restrict__ params,
           const uint8_t *__restrict__ masks, const float *x, const float *__restrict__ c,
           const int64_t *__restrict__ row_index, int64_t n, float *out_x, float *logdet_out, float *logp_out, float *part) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // (the wave number through readfirstlane: known uniform, so the per-wave row bases and LDS offsets are scalar arithmetic --
    // round 6, same box: loss + gradient call of hidden=(128,128) 4.42 -> 4.37 ms, (10,20,15) 1.607 -> 1.589: profiles/r06_wave_sgpr_sweep.txt)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), q = lane >> 4, r = lane & 15, d = s.d, cd = s.c;
    float *XC = lds, *H0 = XC + (d + cd) * RS, *H1 = H0 + g.hmax * RS, *T = H1 + g.hmax * RS, *S = T + d * RS;
    float *RED = lds + g.lds_flow / sizeof(float) - 2 * kW * 16;
    for (int e = tid; e < (int)(g.lds_flow / sizeof(float)); e += 64 * kW) lds[e] = 0.f;      // see linear_mb: stale rows must be finite
    __syncthreads();
    const int64_t ntiles = (n + 15) / 16;
    const float prior_c = 0.5f * (float)d * kLog2Pi;
    float wave_sum = 0.f;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t base = tile * 16;
        load_tile(x, row_index, base, n, d, XC, tid);
        if (cd) load_tile(c, row_index, base, n, cd, XC + d * RS, tid);
        __syncthreads();
        float ld = 0.f;
        for (int lp = 0; lp < s.L; ++lp) {
            const int l = INVERSE ? s.L - 1 - lp : lp;
            const float *pk = packed + (size_t)l * 2 * g.net_floats, *pn = params + (size_t)l * 2 * s.npn;
            net_fwd<false>(pk, pn, s, g, XC, H0, H1, T, lane, wave);
            net_fwd<false>(pk + g.net_floats, pn + s.npn, s, g, XC, H0, H1, S, lane, wave);
            const uint8_t *m = masks + l * d;
            for (int j = q + 4 * wave; j < d; j += 4 * kW) {
                if (!m[j]) {
                    const float sv = S[j * RS + r], tv = T[j * RS + r], xv = XC[j * RS + r];
                    if (INVERSE) XC[j * RS + r] = (xv - tv) * expf(-sv);
                    else { XC[j * RS + r] = fmaf(xv, expf(sv), tv); ld += sv; }
                }
            }
            __syncthreads();
        }
        const int64_t row = base + r;
        const bool valid = row < n;
        if (out_x) {
            for (int e = tid; e < 16 * d; e += 64 * kW) {
                const int rr = e / d, j = e - rr * d;
                if (base + rr < n) out_x[(base + rr) * d + j] = XC[j * RS + rr];
            }
        }
        if (!INVERSE) {
            float ss = 0.f;
            for (int j = q + 4 * wave; j < d; j += 4 * kW) { const float zv = XC[j * RS + r]; ss = fmaf(zv, zv, ss); }
            ld += __shfl_xor(ld, 16); ld += __shfl_xor(ld, 32);
            ss += __shfl_xor(ss, 16); ss += __shfl_xor(ss, 32);
            ld = wg_row_sum(ld, RED, lane, wave);
            ss = wg_row_sum(ss, RED + kW * 16, lane, wave);
            if (wave == 0) {
                const float lpv = ld + (-0.5f * ss - prior_c);
                if (valid && q == 0) {
                    if (logdet_out) logdet_out[row] = ld;
                    if (logp_out) logp_out[row] = lpv;
                }
                float v = (valid && q == 0) ? lpv : 0.f;
                v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
                wave_sum += v;
            }
        }
        __syncthreads();
    }
    if (!INVERSE && part && tid == 0) part[blockIdx.x] = wave_sum;
}

// [feature][RS] image -> [feature][16] tile in global memory, nfeat real features padded with `pad` zero features up
// to ntot; feature `ones_at` (if >= 0) is written as 1 (the bias column of the weight-gradient product)
__device__ __forceinline__ void dump_tile(const float *img, int nfeat, int ntot, int ones_at, float *__restrict__ dst, int lane, int wave) {
    const int q = lane >> 4, r = lane & 15;
    for (int f0 = q + 16 * wave; f0 < ntot; f0 += 16 * kW) {        // four features per lane and pass: the LDS reads overlap
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int f = f0 + 4 * u;
            v[u] = f < nfeat ? img[f * RS + r] : (f == ones_at ? 1.f : 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int f = f0 + 4 * u;
            if (f < ntot) dst[f * 16 + r] = v[u];
        }
    }
}

// one net, backward, for the tile: on entry GA holds d loss / d (net output) and ACT the net's hidden activations
// (net_fwd<true>).  Last Linear to first: activation derivative, dump of the weight-gradient operands (pre-activation
// gradient + the Linear's input with a ones column) into dn, input gradient through the W^T fragments; Linear 0's goes,
// for the first s.d inputs only, ADDED into gin (the condition columns get a gradient -- rows s.d .. s.d + s.c - 1 of gin -- only
// when the shape was packed with gcw: rnvp_backward_cond / rnvp_inverse_backward).  Ends on a barrier.
__device__ __forceinline__ void net_bwd(const float *__restrict__ pkn, const KShape &s, const LGeo &g, const float *in0, float *ACT,
                                        float *GA, float *GB, float *gin, float *__restrict__ dn, int lane, int wave) {
    const int q = lane >> 4, r = lane & 15, nh = s.nh;
    float *gcur = GA, *gprev = GB;
    int aoff = g.hs;                                       // running feature offset of Linear k's own activation block
    for (int k = nh; k >= 0; --k) {
        const int nin = g.nin[k], nout = g.nout[k];
        if (k < nh) {
            aoff -= nout;
            const float *ak = ACT + aoff * RS;
            for (int f0 = q + 16 * wave; f0 < nout; f0 += 16 * kW) {          // four features per pass: overlapped LDS reads
                float a[4], gv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int f = f0 + 4 * u < nout ? f0 + 4 * u : f0;
                    a[u] = ak[f * RS + r]; gv[u] = gcur[f * RS + r];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (f0 + 4 * u < nout)
                        gcur[(f0 + 4 * u) * RS + r] = (s.act == RNVP_ACT_TANH) ? gv[u] * (1.f - a[u] * a[u]) : (a[u] > 0.f ? gv[u] : 0.f);
            }
            __syncthreads();
        }
        const float *inp = (k == 0) ? in0 : ACT + (aoff - nin) * RS;
        dump_tile(gcur, nout, 16 * g.MT[k], -1, dn + g.offP[k], lane, wave);
        dump_tile(inp, nin, 16 * g.PT[k], nin, dn + g.offA[k], lane, wave);
        if (k == 0) linear<true>(pkn + g.offT[0], g.MTt[0], g.KSt[0], nout, s.d + (s.gcw ? s.c : 0), gcur, gin, nullptr, -1, lane, wave);
        else linear<false>(pkn + g.offT[k], g.MTt[k], g.KSt[k], nout, nin, gcur, gprev, nullptr, -1, lane, wave);
        __syncthreads();
        float *tmp = gcur; gcur = gprev; gprev = tmp;
    }
}

// ---- loss + input-gradient chain; dumps the operands of the weight gradients -------------------------------------------
__global__ void __launch_bounds__(64 * kW)
k_lmm_train(KShape s, LGeo g, const float *__restrict__ packed, const float *__restrict__ params,
            const uint8_t *__restrict__ masks, const float *__restrict__ x, const float *__restrict__ c,
            const int64_t *__restrict__ row_index, int64_t n, float inv_B, Seeds sd,
            float *__restrict__ dump, float *__restrict__ xsave, float *losspart, int first_chunk) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), q = lane >> 4, r = lane & 15, d = s.d, cd = s.c;
    const int jq = q + 4 * wave, jstep = 4 * kW;           // this lane's features in the elementwise passes
    const float *__restrict__ gz = sd.gz;
    float *XC = lds, *ACT = XC + (d + cd) * RS, *T = ACT + g.hs * RS, *S = T + d * RS, *GY = S + d * RS, *GIN = GY + d * RS;
    const int gcr = s.gcw ? cd : 0;                        // rows d .. d + gcr - 1 of GIN: d loss / d c, summed over layers and nets
    const bool inv = sd.inv != 0;                          // backward through the inverse (Seeds)
    float *GA = GIN + (d + gcr) * RS, *GB = GA + g.wmax * RS;
    float *RED = lds + g.lds_train / sizeof(float) - 2 * kW * 16;
    for (int e = tid; e < (int)(g.lds_train / sizeof(float)); e += 64 * kW) lds[e] = 0.f;     // see linear_mb: stale rows must be finite
    __syncthreads();
    const int64_t ntiles = (n + 15) / 16;
    const float prior_c = 0.5f * (float)d * kLog2Pi;
    float wave_sum = 0.f;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t base = tile * 16, row = base + r;
        const bool valid = row < n;
        load_tile(x, row_index, base, n, d, XC, tid);
        if (cd) load_tile(c, row_index, base, n, cd, XC + d * RS, tid);
        __syncthreads();
        float *xs = xsave + (size_t)tile * s.L * d * 16;
        float ld = 0.f;
        for (int lp = 0; lp < s.L; ++lp) {
            // inv: the rows are z; the inverse pass (layers L-1 .. 0, realnvp.py:120-128) leaves X_l, the input of f's layer l,
            // behind it -- the same record the forward saves in front of the layer
            const int l = inv ? s.L - 1 - lp : lp;
            const float *pk = packed + (size_t)l * 2 * g.net_floats, *pn = params + (size_t)l * 2 * s.npn;
            if (!inv)
                for (int j = jq; j < d; j += jstep) xs[(l * d + j) * 16 + r] = XC[j * RS + r];    // layer input, for the backward
            net_fwd<false>(pk, pn, s, g, XC, GA, GB, T, lane, wave);
            net_fwd<false>(pk + g.net_floats, pn + s.npn, s, g, XC, GA, GB, S, lane, wave);
            const uint8_t *m = masks + l * d;
            for (int j = jq; j < d; j += jstep) {
                if (!m[j]) {
                    const float sv = S[j * RS + r];
                    if (inv) XC[j * RS + r] = (XC[j * RS + r] - T[j * RS + r]) * expf(-sv);
                    else { XC[j * RS + r] = fmaf(XC[j * RS + r], expf(sv), T[j * RS + r]); ld += sv; }
                }
                if (inv) xs[(l * d + j) * 16 + r] = XC[j * RS + r];
            }
            __syncthreads();
        }
        {   // loss terms and the seed of the backward
            float ss = 0.f;
            for (int j = jq; j < d; j += jstep) { const float zv = XC[j * RS + r]; ss = fmaf(zv, zv, ss); }
            ld += __shfl_xor(ld, 16); ld += __shfl_xor(ld, 32);
            ss += __shfl_xor(ss, 16); ss += __shfl_xor(ss, 32);
            ld = wg_row_sum(ld, RED, lane, wave);
            ss = wg_row_sum(ss, RED + kW * 16, lane, wave);
            if (wave == 0) {
                const float lpv = gz ? ld : ld + (-0.5f * ss - prior_c);
                float v = (valid && q == 0) ? lpv : 0.f;
                v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
                wave_sum += v;
            }
            for (int j = jq; j < d; j += jstep)
                GY[j * RS + r] = valid ? (gz ? gz[row * d + j] : XC[j * RS + r] * inv_B) : 0.f;
            for (int j = jq; j < gcr; j += jstep) GIN[(d + j) * RS + r] = 0.f;
        }
        const float gld = inv ? 0.f : (valid ? (sd.gld ? sd.gld[row] : -inv_B) : 0.f);      // rnvp_backward: the caller's d loss / d logdet
        __syncthreads();
        for (int lp = s.L - 1; lp >= 0; --lp) {
            const int l = inv ? s.L - 1 - lp : lp;          // inv: the inverse applied layer 0 last, so its backward starts there
            const float *pk = packed + (size_t)l * 2 * g.net_floats, *pn = params + (size_t)l * 2 * s.npn;
            const uint8_t *m = masks + l * d;
            float *dl = dump + ((size_t)tile * s.L + l) * 2 * g.dump_floats;
            for (int j = jq; j < d; j += jstep) { XC[j * RS + r] = xs[(l * d + j) * 16 + r]; GIN[j * RS + r] = 0.f; }
            __syncthreads();
            for (int net = 1; net >= 0; --net) {                     // s first (its output is needed for exp(s)), then t
                const float *pkn = pk + net * g.net_floats, *pnn = pn + (size_t)net * s.npn;
                float *dn = dl + (size_t)net * g.dump_floats;
                net_fwd<true>(pkn, pnn, s, g, XC, ACT, nullptr, net ? S : T, lane, wave);
                // d loss / d (net output): s: (1-m)(gy x e^s + gld), t: (1-m) gy
                for (int j = jq; j < d; j += jstep) {
                    float v = 0.f;
                    if (!m[j]) {
                        const float gy = GY[j * RS + r];
                        // inv: x_u = (y_u - t) e^{-s}  =>  d / d t = -gy e^{-s},  d / d s = -gy x_u   (XC holds x_u = X_l)
                        if (inv) v = net ? -gy * XC[j * RS + r] : -gy * expf(-S[j * RS + r]);
                        else v = net ? fmaf(gy * XC[j * RS + r], expf(S[j * RS + r]), gld) : gy;
                    }
                    GA[j * RS + r] = v;
                }
                __syncthreads();
                net_bwd(pkn, s, g, XC, ACT, GA, GB, GIN, dn, lane, wave);
            }
            for (int j = jq; j < d; j += jstep) {
                const float gy = GY[j * RS + r];
                GY[j * RS + r] = (m[j] ? gy : gy * expf(inv ? -S[j * RS + r] : S[j * RS + r])) + GIN[j * RS + r];
            }
            __syncthreads();
        }
        if (sd.gx && valid)                                  // rnvp_backward: d loss / d x of the batch rows (inv: d loss / d z)
            for (int j = jq; j < d; j += jstep) sd.gx[row * d + j] = GY[j * RS + r];
        if (sd.gc && valid)                                  // d loss / d c (shape packed with gcw)
            for (int j = jq; j < gcr; j += jstep) sd.gc[row * cd + j] = GIN[(d + j) * RS + r];
    }
    if (tid == 0) losspart[blockIdx.x] = first_chunk ? wave_sum : losspart[blockIdx.x] + wave_sum;      // chunks run in order on one stream
}

// ---- CVAE: encoder -> reparameterize -> decoder -> KL + MSE -> backward, one 16-row tile per workgroup -----------------
// (/root/reference/probaforms/models/cvae.py:186-199 compute_loss; the hand-derived backward of cvae_generic.hip.)
// The encoder and the decoder are single nets (LGeo::nnets = 1) with their own geometry; the tile's LDS image is
//   EIN [x | c]  EACT [sum hidden]  EO [mu | log_sigma]  DIN [z | c]  DACT [sum hidden]  XR [d]  EPS [lat]  GZ [lat]
//   GA, GB [widest layer]  (+ slack, + reduction scratch)
// and the weight gradients go through the same dump -> k_lmm_wgrad -> k_lmm_reduce pipeline, once per net.
struct CvaeL {
    KShape enc, dec;
    LGeo ge, gd;
    int d, c, lat, pe, wm;
    size_t lds_train, lds_enc, lds_dec;
};

__global__ void __launch_bounds__(64 * kW)
k_lmm_cvae_train(CvaeL s, const float *__restrict__ packed_e, const float *__restrict__ packed_d, const float *__restrict__ params,
                 const float *__restrict__ x, const float *__restrict__ c, const int64_t *__restrict__ row_index,
                 const float *__restrict__ eps, int64_t n, float inv_B, float klw, float *__restrict__ dump_e,
                 float *__restrict__ dump_d, float *losspart, int first_chunk, int do_grad) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), q = lane >> 4, r = lane & 15;
    const int d = s.d, cd = s.c, lat = s.lat;
    const int jq = q + 4 * wave, jstep = 4 * kW;
    float *EIN = lds, *EACT = EIN + (d + cd) * RS, *EO = EACT + s.enc.hs * RS, *DIN = EO + 2 * lat * RS;
    float *DACT = DIN + (lat + cd) * RS, *XR = DACT + s.dec.hs * RS, *EPS = XR + d * RS, *GZ = EPS + lat * RS;
    float *GA = GZ + lat * RS, *GB = GA + s.wm * RS;
    float *RED = lds + s.lds_train / sizeof(float) - 2 * kW * 16;
    for (int e = tid; e < (int)(s.lds_train / sizeof(float)); e += 64 * kW) lds[e] = 0.f;     // see linear_mb: stale rows must be finite
    __syncthreads();
    const int64_t ntiles = (n + 15) / 16;
    const float inv_d = 1.f / (float)d;
    float wave_sum = 0.f;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t base = tile * 16, row = base + r;
        const bool valid = row < n;
        load_tile(x, row_index, base, n, d, EIN, tid);                                            // cat(X, C), cvae.py:58
        if (cd) { load_tile(c, row_index, base, n, cd, EIN + d * RS, tid); load_tile(c, row_index, base, n, cd, DIN + lat * RS, tid); }
        load_tile(eps, nullptr, base, n, lat, EPS, tid);                                          // eps is already in batch order
        __syncthreads();
        net_fwd<true>(packed_e, params, s.enc, s.ge, EIN, EACT, nullptr, EO, lane, wave);         // mu | log_sigma
        for (int j = jq; j < lat; j += jstep)                                                     // sample_z, cvae.py:188
            DIN[j * RS + r] = fmaf(expf(0.5f * EO[(lat + j) * RS + r]), EPS[j * RS + r], EO[j * RS + r]);
        __syncthreads();
        net_fwd<true>(packed_d, params + s.pe, s.dec, s.gd, DIN, DACT, nullptr, XR, lane, wave);
        {
            float kl = 0.f, se = 0.f;
            for (int j = jq; j < lat; j += jstep) {
                const float mu = EO[j * RS + r], ls = EO[(lat + j) * RS + r];
                kl += 1.f + ls - mu * mu - expf(ls);                                              // cvae.py:191
            }
            for (int j = jq; j < d; j += jstep) { const float df = EIN[j * RS + r] - XR[j * RS + r]; se = fmaf(df, df, se); }
            float lrow = klw * (-0.5f * kl) + se * inv_d;
            lrow += __shfl_xor(lrow, 16); lrow += __shfl_xor(lrow, 32);
            lrow = wg_row_sum(lrow, RED, lane, wave);
            if (wave == 0) {
                float v = (valid && q == 0) ? lrow : 0.f;
                v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
                wave_sum -= v;                                    // k_lmm_reduce reports -(sum of partials) * inv_B
            }
        }
        if (do_grad) {
            const float sc = valid ? inv_B : 0.f;
            for (int j = jq; j < lat; j += jstep) GZ[j * RS + r] = 0.f;
            for (int j = jq; j < d; j += jstep)                                                   // d MSE / d x_rec
                GA[j * RS + r] = (2.f * sc * inv_d) * (XR[j * RS + r] - EIN[j * RS + r]);
            __syncthreads();
            net_bwd(packed_d, s.dec, s.gd, DIN, DACT, GA, GB, GZ, dump_d + (size_t)tile * s.gd.dump_floats, lane, wave);
            for (int j = jq; j < lat; j += jstep) {
                const float mu = EO[j * RS + r], ls = EO[(lat + j) * RS + r], gv = GZ[j * RS + r];
                GA[j * RS + r] = fmaf(klw * sc, mu, gv);                                          // d / d mu
                GA[(lat + j) * RS + r] = gv * EPS[j * RS + r] * 0.5f * expf(0.5f * ls)            // d / d log_sigma
                                         + klw * sc * (-0.5f) * (1.f - expf(ls));
            }
            __syncthreads();
            net_bwd(packed_e, s.enc, s.ge, EIN, EACT, GA, GB, GZ, dump_e + (size_t)tile * s.ge.dump_floats, lane, wave);
        }
        __syncthreads();
    }
    if (tid == 0) losspart[blockIdx.x] = first_chunk ? wave_sum : losspart[blockIdx.x] + wave_sum;
}

// encoder (mu, log_sigma) or decoder (x_rec) alone: cvae.py:58-62 / 81-84
template <bool ENCODE>
__global__ void __launch_bounds__(64 * kW)
k_lmm_cvae_mlp(CvaeL s, const float *__restrict__ packed, const float *__restrict__ params, const float *__restrict__ a,
               const float *__restrict__ c, int64_t n, float *__restrict__ out0, float *__restrict__ out1) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const KShape &m = ENCODE ? s.enc : s.dec;
    const LGeo &g = ENCODE ? s.ge : s.gd;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int na = ENCODE ? s.d : s.lat, cd = s.c, no = ENCODE ? 2 * s.lat : s.d;
    const size_t bytes = ENCODE ? s.lds_enc : s.lds_dec;
    float *IN = lds, *H0 = IN + (na + cd) * RS, *H1 = H0 + m.hmax * RS, *O = H1 + m.hmax * RS;
    for (int e = tid; e < (int)(bytes / sizeof(float)); e += 64 * kW) lds[e] = 0.f;
    __syncthreads();
    const float *pn = ENCODE ? params : params + s.pe;
    const int64_t ntiles = (n + 15) / 16;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t base = tile * 16;
        load_tile(a, nullptr, base, n, na, IN, tid);
        if (cd) load_tile(c, nullptr, base, n, cd, IN + na * RS, tid);
        __syncthreads();
        net_fwd<false>(packed, pn, m, g, IN, H0, H1, O, lane, wave);
        for (int e = tid; e < 16 * no; e += 64 * kW) {
            const int rr = e / no, j = e - rr * no;
            if (base + rr < n) {
                const float v = O[j * RS + rr];
                if (!ENCODE) out0[(base + rr) * no + j] = v;
                else if (j < s.lat) out0[(base + rr) * s.lat + j] = v;
                else out1[(base + rr) * s.lat + (j - s.lat)] = v;
            }
        }
        __syncthreads();
    }
}

// ---- weight gradients: dW[k] tile (m, p) = sum over rows gP[row][16m + i] * act[row][16p + j] ---------------------------
// one wave per (layer, net, Linear, 2 x 2 block of tile pairs) and row split; operands are the [feature][16 rows] tiles
// dumped above, read as one float4 per lane (rows 4q .. 4q+3 of feature i): k-step ks of lane group q stands for row
// 4q + ks on both sides.  The 2 x 2 block takes 4 loads of 1 KiB per 16 MFMAs instead of 8.
__global__ void __launch_bounds__(64)
k_lmm_wgrad(KShape s, LGeo g, const float *__restrict__ dump, int64_t ntiles, float *__restrict__ gpart, int accumulate) {
    const int lane = threadIdx.x, q = lane >> 4, i = lane & 15;
    int quad = blockIdx.x;
    const int ln = quad / g.quads_per_net;
    quad -= ln * g.quads_per_net;
    int k = 0;
    while (k + 1 < g.nlin && quad >= ((g.MT[k] + 1) >> 1) * ((g.PT[k] + 1) >> 1)) { quad -= ((g.MT[k] + 1) >> 1) * ((g.PT[k] + 1) >> 1); ++k; }
    const int pb2 = (g.PT[k] + 1) >> 1;
    const int m0 = 2 * (quad / pb2), p0 = 2 * (quad % pb2);
    const bool m1 = m0 + 1 < g.MT[k], p1 = p0 + 1 < g.PT[k];
    const size_t tstride = (size_t)g.nnets * g.dump_floats;
    const float *pa = dump + (size_t)ln * g.dump_floats + g.offP[k] + m0 * 256 + i * 16 + 4 * q;
    const float *pb = dump + (size_t)ln * g.dump_floats + g.offA[k] + p0 * 256 + i * 16 + 4 * q;
    f4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f4{0.f, 0.f, 0.f, 0.f};
    const f4 zero = f4{0.f, 0.f, 0.f, 0.f};
    for (int64_t t = blockIdx.y; t < ntiles; t += gridDim.y) {
        const f4 a0 = *reinterpret_cast<const f4 *>(pa + (size_t)t * tstride);
        const f4 a1 = m1 ? *reinterpret_cast<const f4 *>(pa + (size_t)t * tstride + 256) : zero;
        const f4 b0 = *reinterpret_cast<const f4 *>(pb + (size_t)t * tstride);
        const f4 b1 = p1 ? *reinterpret_cast<const f4 *>(pb + (size_t)t * tstride + 256) : zero;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            acc[0][0] = mfma16(a0[ks], b0[ks], acc[0][0]);
            acc[0][1] = mfma16(a0[ks], b1[ks], acc[0][1]);
            acc[1][0] = mfma16(a1[ks], b0[ks], acc[1][0]);
            acc[1][1] = mfma16(a1[ks], b1[ks], acc[1][1]);
        }
    }
    float *dst = gpart + ((size_t)blockIdx.y * g.nnets + ln) * g.gnet_floats + g.offG[k] + lane * 4;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
            if ((a == 0 || m1) && (b == 0 || p1)) {
                f4 *o = reinterpret_cast<f4 *>(dst + ((m0 + a) * g.PT[k] + p0 + b) * 256);
                *o = accumulate ? *o + acc[a][b] : acc[a][b];          // row chunks of one call, in order: deterministic
            }
}

// flat reference-order gradient: sum of the row splits in index order; loss = -(sum of wave partials) * inv_B
__global__ void __launch_bounds__(256)
k_lmm_reduce(KShape s, LGeo g, const float *__restrict__ gpart, int S, const uint8_t *__restrict__ masks,
             const float *__restrict__ losspart, int G, float inv_B, float *__restrict__ grad, float *loss) {
    const size_t P = (size_t)g.nnets * s.npn;
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) {
        if (loss && blockIdx.x == gridDim.x - 1 && threadIdx.x >= 192) {
            const int lane = threadIdx.x - 192;
            float a = 0.f;
            for (int t = lane; t < G; t += 64) a += losspart[t];
            for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off);
            if (lane == 0) loss[0] = -a * inv_B;
        }
        return;
    }
    const int ln = (int)(p / s.npn), l = ln >> 1;
    const int idx = (int)(p - (size_t)ln * s.npn);
    int k = 0;
    while (k + 1 < g.nlin && idx >= s.woff[k + 1]) ++k;
    const int nin = g.nin[k];
    int o, in;
    float scale = 1.f;
    if (idx < s.boff[k]) {
        o = (idx - s.woff[k]) / nin; in = (idx - s.woff[k]) - o * nin;
        if (masks && k == 0 && in < s.d) scale = (float)masks[l * s.d + in];          // W'[:, j] = W[:, j] * mask_j
    } else { o = idx - s.boff[k]; in = nin; }                                // bias: the ones column
    const int m = o >> 4, pt = in >> 4;
    const int loc = g.offG[k] + (m * g.PT[k] + pt) * 256 + ((((o & 15) >> 2) * 16) + (in & 15)) * 4 + (o & 3);
    const float *src = gpart + (size_t)ln * g.gnet_floats + loc;
    const size_t stride = (size_t)g.nnets * g.gnet_floats;
    float a = 0.f;
    for (int b = 0; b < S; ++b) a += src[(size_t)b * stride];
    grad[p] = a * scale;
}

std::atomic<uint64_t> g_attr_fwd{0}, g_attr_inv{0}, g_attr_train{0};

int grid_for(int64_t ntiles) { return (int)(ntiles < kMaxGrid ? ntiles : kMaxGrid); }

}  // namespace

LGeo make_lgeo(const KShape &k) {
    LGeo g;
    std::memset(&g, 0, sizeof(g));
    g.nlin = k.nh + 1;
    g.nnets = 2 * k.L;
    int oW = 0, oG = 0, oD = 0, pairs = 0, quads = 0;
    for (int i = 0; i < g.nlin; ++i) {
        g.nin[i] = k.nin[i]; g.nout[i] = k.nout[i];
        g.MT[i] = (k.nout[i] + 15) / 16; g.KS[i] = ((k.nin[i] + 3) / 4 + 3) / 4 * 4;          // k-steps padded to groups of 4
        g.MTt[i] = ((i == 0 ? k.d + (k.gcw ? k.c : 0) : k.nin[i]) + 15) / 16; g.KSt[i] = ((k.nout[i] + 3) / 4 + 3) / 4 * 4;
        g.PT[i] = (k.nin[i] + 1 + 15) / 16;
        g.offF[i] = oW; oW += g.MT[i] * g.KS[i] * 64;
        g.offT[i] = oW; oW += g.MTt[i] * g.KSt[i] * 64;
        g.offG[i] = oG; oG += g.MT[i] * g.PT[i] * 256;
        g.offA[i] = oD; oD += g.PT[i] * 256;
        g.offP[i] = oD; oD += g.MT[i] * 256;
        pairs += g.MT[i] * g.PT[i];
        quads += ((g.MT[i] + 1) / 2) * ((g.PT[i] + 1) / 2);
    }
    g.quads_per_net = quads;
    g.net_floats = oW; g.gnet_floats = oG; g.dump_floats = oD; g.pairs_per_net = pairs;
    g.hs = k.hs; g.hmax = k.hmax;
    g.wmax = k.hmax > k.d ? k.hmax : k.d;
    // + 16 slack rows: a padded k-group of the last buffer may read (and multiply by zero weights) up to 15 rows past it
    // ... then 2 x kW x 16 floats of cross-wave reduction scratch at the very end
    g.lds_flow = ((size_t)(3 * k.d + k.c + 2 * k.hmax + 16) * RS + 2 * kW * 16) * sizeof(float);
    g.lds_train = ((size_t)(k.d + k.c + k.hs + 4 * k.d + (k.gcw ? k.c : 0) + 2 * g.wmax + 16) * RS + 2 * kW * 16) * sizeof(float);
    return g;
}

// auto: whenever a tile's LDS image leaves room for at least two workgroups per CU.  Measured against the VALU kernels
// (rnvp_generic.hip) the MFMA form wins at every width and batch size tried -- hidden=(10,20,15), d=2, batch 32:
// 245 vs 1368 us per training step; hidden=(128,128), 65536 rows: 3.9 vs 217 ms -- so those remain only for shapes whose
// image does not fit (and as the second implementation the tests pin this one against).
bool use_lmm(const KShape &k, int op) {
    if (k.family == RNVP_FAMILY_VALU) return false;
    const LGeo g = make_lgeo(k);
    const size_t need = op == RNVP_OP_TRAIN ? g.lds_train : g.lds_flow;
    return need <= 76 * 1024;
}

// rows per pass of the training kernels: the dumped weight-gradient operands take dump_floats * L * 2 / 16 floats per
// row (38 KB for hidden = (128,128), L = 8), so a big batch is processed in row chunks that keep them under ~1 GiB
static int64_t chunk_rows(const KShape &k, const LGeo &g) {
    const double per_row = (double)k.L * 2 * g.dump_floats * sizeof(float) / 16.0;
    int64_t r = (int64_t)((1u << 30) / per_row) / 16 * 16;
    if (r < 4096) r = 4096;
    if (r > 65536) r = 65536;
    return r;
}

size_t workspace_bytes(const KShape &k, int op, int64_t max_rows) {
    const LGeo g = make_lgeo(k);
    size_t b = align_up((size_t)k.L * 2 * g.net_floats * sizeof(float), 256);                       // packed weights
    b += align_up((size_t)kMaxGrid * sizeof(float), 256);                                           // per-wave loss / log-prob partials
    if (op == RNVP_OP_TRAIN) {
        const int64_t cr = chunk_rows(k, g);
        const int64_t ntiles = ((max_rows < cr ? max_rows : cr) + 15) / 16;
        b += align_up((size_t)ntiles * k.L * 2 * g.dump_floats * sizeof(float), 256);               // wgrad operands of one chunk
        b += align_up((size_t)ntiles * k.L * k.d * 16 * sizeof(float), 256);                        // layer inputs of one chunk
        b += align_up((size_t)kSplits * k.L * 2 * g.gnet_floats * sizeof(float), 256);              // split partials
        if (!k.gcw && use_train64(k, max_rows)) { const size_t b64 = train64_workspace_bytes(k, max_rows); if (b64 > b) b = b64; }
    }
    return b;
}

static int pack(hipStream_t st, const KShape &k, const LGeo &g, const float *params, const uint8_t *masks, float *packed) {
    const int64_t total = (int64_t)k.L * 2 * g.net_floats;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_lmm_pack, dim3(blocks), dim3(256), 0, st, k, g, params, masks, packed);
    RNVP_HIP_TRY(hipGetLastError());
    return RNVP_OK;
}

int forward(hipStream_t st, const KShape &k, const float *params, const uint8_t *masks, const float *x, const float *c,
            const int64_t *row_index, int64_t n, float *z_out, float *logdet_out, float *logp_out, float *logp_sum,
            void *ws, size_t ws_bytes) {
    if (!ws || ws_bytes < workspace_bytes(k, RNVP_OP_FORWARD, n)) return RNVP_EWORKSPACE;
    const LGeo g = make_lgeo(k);
    float *packed = static_cast<float *>(ws);
    float *part = reinterpret_cast<float *>(static_cast<char *>(ws) + align_up((size_t)k.L * 2 * g.net_floats * sizeof(float), 256));
    int rc = pack(st, k, g, params, masks, packed);
    if (rc) return rc;
    rc = allow_big_lds(reinterpret_cast<const void *>(k_lmm_flow<false>), 160 * 1024, g_attr_fwd);
    if (rc) return rc;
    const int G = grid_for((n + 15) / 16);
    note_dispatch(RNVP_PROFILE_FORWARD, "k_lmm_flow", RNVP_VARIANT_LMM, 1, kW, G, RNVP_PREC_F32, n);
    hipLaunchKernelGGL(k_lmm_flow<false>, dim3(G), dim3(64 * kW), g.lds_flow, st, k, g, packed, params, masks, x, c, row_index, n,
                       z_out, logdet_out, logp_out, part);
    RNVP_HIP_TRY(hipGetLastError());
    if (logp_sum) return generic_reduce_partials(st, nullptr, part, G, 0, 1.0f, nullptr, logp_sum);
    return RNVP_OK;
}

int inverse(hipStream_t st, const KShape &k, const float *params, const uint8_t *masks, const float *z, const float *c,
            int64_t n, float *x_out, void *ws, size_t ws_bytes) {
    if (!ws || ws_bytes < workspace_bytes(k, RNVP_OP_INVERSE, n)) return RNVP_EWORKSPACE;
    const LGeo g = make_lgeo(k);
    float *packed = static_cast<float *>(ws);
    int rc = pack(st, k, g, params, masks, packed);
    if (rc) return rc;
    rc = allow_big_lds(reinterpret_cast<const void *>(k_lmm_flow<true>), 160 * 1024, g_attr_inv);
    if (rc) return rc;
    note_dispatch(RNVP_PROFILE_INVERSE, "k_lmm_flow", RNVP_VARIANT_LMM, 1, kW, grid_for((n + 15) / 16), RNVP_PREC_F32, n);
    hipLaunchKernelGGL(k_lmm_flow<true>, dim3(grid_for((n + 15) / 16)), dim3(64 * kW), g.lds_flow, st, k, g, packed, params, masks, z,
                       c, nullptr, n, x_out, nullptr, nullptr, nullptr);
    RNVP_HIP_TRY(hipGetLastError());
    return RNVP_OK;
}

int loss_grad(hipStream_t st, const KShape &k, const float *params, const uint8_t *masks, const float *x, const float *c,
              const int64_t *row_index, int64_t n, float inv_B, float *grad_out, float *loss_out, void *ws, size_t ws_bytes,
              Seeds sd) {
    if (sd.gc && !k.gcw) return RNVP_EINVAL;
    // d loss / d c and the backward through the inverse live in the 16-row kernel only
    if (!sd.gc && !sd.inv && !k.gcw && use_train64(k, n))
        return loss_grad64(st, k, params, masks, x, c, row_index, n, inv_B, grad_out, loss_out, ws, ws_bytes, sd);
    if (!ws || ws_bytes < workspace_bytes(k, RNVP_OP_TRAIN, n)) return RNVP_EWORKSPACE;
    const LGeo g = make_lgeo(k);
    const int64_t cr = chunk_rows(k, g);
    const int64_t ctiles = ((n < cr ? n : cr) + 15) / 16;
    char *w = static_cast<char *>(ws);
    float *packed = reinterpret_cast<float *>(w); w += align_up((size_t)k.L * 2 * g.net_floats * sizeof(float), 256);
    float *losspart = reinterpret_cast<float *>(w); w += align_up((size_t)kMaxGrid * sizeof(float), 256);
    float *dump = reinterpret_cast<float *>(w); w += align_up((size_t)ctiles * k.L * 2 * g.dump_floats * sizeof(float), 256);
    float *xsave = reinterpret_cast<float *>(w); w += align_up((size_t)ctiles * k.L * k.d * 16 * sizeof(float), 256);
    float *gpart = reinterpret_cast<float *>(w);
    int rc = pack(st, k, g, params, masks, packed);
    if (rc) return rc;
    rc = allow_big_lds(reinterpret_cast<const void *>(k_lmm_train), 160 * 1024, g_attr_train);
    if (rc) return rc;
    // every chunk uses the same grid and the same number of row splits, so partial b always holds the same rows' sums
    const int G = grid_for(ctiles);
    const int S = (int)(ctiles < kSplits ? ctiles : kSplits);
    for (int64_t r0 = 0; r0 < n; r0 += cr) {
        const int64_t rows = n - r0 < cr ? n - r0 : cr;
        const int64_t ntiles = (rows + 15) / 16;
        const bool first = r0 == 0;
        const float *xc = row_index ? x : x + r0 * k.d;
        const float *cc = (row_index || !c) ? c : c + r0 * k.c;
        note_dispatch(RNVP_PROFILE_TRAIN, "k_lmm_train", RNVP_VARIANT_LMM, 1, kW, G, RNVP_PREC_F32, rows);
        {
            KernelTimer timer(st, RNVP_PROFILE_TRAIN);
            hipLaunchKernelGGL(k_lmm_train, dim3(G), dim3(64 * kW), g.lds_train, st, k, g, packed, params, masks, xc, cc,
                               row_index ? row_index + r0 : nullptr, rows, inv_B,
                               Seeds{sd.gz ? sd.gz + r0 * k.d : nullptr, sd.gld ? sd.gld + r0 : nullptr, sd.gx ? sd.gx + r0 * k.d : nullptr,
                                     sd.gc ? sd.gc + r0 * k.c : nullptr, sd.inv},
                               dump, xsave, losspart, first ? 1 : 0);
        }
        RNVP_HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(k_lmm_wgrad, dim3((unsigned)(k.L * 2 * g.quads_per_net), (unsigned)S), dim3(64), 0, st, k, g, dump,
                           ntiles, gpart, first ? 0 : 1);
        RNVP_HIP_TRY(hipGetLastError());
    }
    const size_t P = (size_t)2 * k.npn * k.L;
    hipLaunchKernelGGL(k_lmm_reduce, dim3((unsigned)(P / 256 + 2)), dim3(256), 0, st, k, g, gpart, S, masks, losspart, G, inv_B,
                       grad_out, loss_out);
    RNVP_HIP_TRY(hipGetLastError());
    return RNVP_OK;
}

// ---- CVAE host side ------------------------------------------------------------------------------------------------------
namespace {

constexpr size_t kCvaeLdsMax = 159 * 1024;          // one workgroup per CU at worst: still far ahead of one thread per row

LGeo single_net_geo(const KShape &k) { LGeo g = make_lgeo(k); g.nnets = 1; return g; }

CvaeL make_cvae_l(const CvaeK &k) {
    CvaeL s;
    std::memset(&s, 0, sizeof(s));
    s.enc = k.enc; s.dec = k.dec; s.d = k.d; s.c = k.c; s.lat = k.lat; s.pe = k.pe;
    s.ge = single_net_geo(k.enc); s.gd = single_net_geo(k.dec);
    int wm = k.enc.hmax > k.dec.hmax ? k.enc.hmax : k.dec.hmax;
    if (2 * k.lat > wm) wm = 2 * k.lat;
    if (k.d > wm) wm = k.d;
    s.wm = wm;
    const size_t rows_train = (size_t)(k.d + k.c) + k.enc.hs + 2 * k.lat + (k.lat + k.c) + k.dec.hs + k.d + 2 * k.lat + 2 * wm + 16;
    s.lds_train = (rows_train * RS + 2 * kW * 16) * sizeof(float);
    s.lds_enc = ((size_t)(k.d + k.c + 2 * k.enc.hmax + 2 * k.lat + 16) * RS) * sizeof(float);
    s.lds_dec = ((size_t)(k.lat + k.c + 2 * k.dec.hmax + k.d + 16) * RS) * sizeof(float);
    return s;
}

int64_t cvae_chunk_rows(const CvaeL &s) {
    const double per_row = (double)(s.ge.dump_floats + s.gd.dump_floats) * sizeof(float) / 16.0;
    int64_t r = (int64_t)((1u << 30) / per_row) / 16 * 16;
    if (r < 4096) r = 4096;
    if (r > 65536) r = 65536;
    return r;
}

int pack_net(hipStream_t st, const KShape &k, const LGeo &g, const float *params, float *packed) {
    int blocks = (g.net_floats + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_lmm_pack, dim3(blocks), dim3(256), 0, st, k, g, params, (const uint8_t *)nullptr, packed);
    RNVP_HIP_TRY(hipGetLastError());
    return RNVP_OK;
}

std::atomic<uint64_t> g_attr_cvae_train{0}, g_attr_cvae_enc{0}, g_attr_cvae_dec{0};

}  // namespace

bool cvae_fits(const CvaeK &k, int op) {
    const CvaeL s = make_cvae_l(k);
    const size_t need = op == RNVP_OP_TRAIN ? s.lds_train : (op == RNVP_OP_FORWARD ? s.lds_enc : s.lds_dec);
    return need <= kCvaeLdsMax;
}

size_t cvae_workspace_bytes(const CvaeK &k, int64_t max_rows) {
    const CvaeL s = make_cvae_l(k);
    if (max_rows < 1) max_rows = 1;
    const int64_t cr = cvae_chunk_rows(s);
    const int64_t ntiles = ((max_rows < cr ? max_rows : cr) + 15) / 16;
    size_t b = align_up((size_t)s.ge.net_floats * sizeof(float), 256) + align_up((size_t)s.gd.net_floats * sizeof(float), 256);
    b += align_up((size_t)kMaxGrid * sizeof(float), 256);
    b += align_up((size_t)ntiles * s.ge.dump_floats * sizeof(float), 256) + align_up((size_t)ntiles * s.gd.dump_floats * sizeof(float), 256);
    b += align_up((size_t)kSplits * s.ge.gnet_floats * sizeof(float), 256) + align_up((size_t)kSplits * s.gd.gnet_floats * sizeof(float), 256);
    return b;
}

int cvae_loss_grad(hipStream_t st, const CvaeK &k, const float *params, const float *x, const float *c, const int64_t *row_index,
                   const float *eps, int64_t n, float inv_B, float klw, float *grad_out, float *loss_out, void *ws, size_t ws_bytes) {
    if (!ws || ws_bytes < cvae_workspace_bytes(k, n)) return RNVP_EWORKSPACE;
    const CvaeL s = make_cvae_l(k);
    const int64_t cr = cvae_chunk_rows(s);
    const int64_t ctiles = ((n < cr ? n : cr) + 15) / 16;
    char *w = static_cast<char *>(ws);
    float *packed_e = reinterpret_cast<float *>(w); w += align_up((size_t)s.ge.net_floats * sizeof(float), 256);
    float *packed_d = reinterpret_cast<float *>(w); w += align_up((size_t)s.gd.net_floats * sizeof(float), 256);
    float *losspart = reinterpret_cast<float *>(w); w += align_up((size_t)kMaxGrid * sizeof(float), 256);
    float *dump_e = reinterpret_cast<float *>(w); w += align_up((size_t)ctiles * s.ge.dump_floats * sizeof(float), 256);
    float *dump_d = reinterpret_cast<float *>(w); w += align_up((size_t)ctiles * s.gd.dump_floats * sizeof(float), 256);
    float *gpart_e = reinterpret_cast<float *>(w); w += align_up((size_t)kSplits * s.ge.gnet_floats * sizeof(float), 256);
    float *gpart_d = reinterpret_cast<float *>(w);
    int rc = pack_net(st, s.enc, s.ge, params, packed_e);
    if (rc) return rc;
    rc = pack_net(st, s.dec, s.gd, params + s.pe, packed_d);
    if (rc) return rc;
    rc = allow_big_lds(reinterpret_cast<const void *>(k_lmm_cvae_train), 160 * 1024, g_attr_cvae_train);
    if (rc) return rc;
    const int G = grid_for(ctiles);
    const int S = (int)(ctiles < kSplits ? ctiles : kSplits);
    for (int64_t r0 = 0; r0 < n; r0 += cr) {
        const int64_t rows = n - r0 < cr ? n - r0 : cr;
        const int64_t ntiles = (rows + 15) / 16;
        const bool first = r0 == 0;
        const float *xc = row_index ? x : x + r0 * k.d;
        const float *cc = (row_index || !c) ? c : c + r0 * k.c;
        {
            KernelTimer timer(st, RNVP_PROFILE_TRAIN);
            hipLaunchKernelGGL(k_lmm_cvae_train, dim3(G), dim3(64 * kW), s.lds_train, st, s, packed_e, packed_d, params, xc, cc,
                               row_index ? row_index + r0 : nullptr, eps + r0 * k.lat, rows, inv_B, klw, dump_e, dump_d, losspart,
                               first ? 1 : 0, grad_out ? 1 : 0);
        }
        RNVP_HIP_TRY(hipGetLastError());
        if (grad_out) {
            hipLaunchKernelGGL(k_lmm_wgrad, dim3((unsigned)s.ge.quads_per_net, (unsigned)S), dim3(64), 0, st, s.enc, s.ge, dump_e, ntiles,
                               gpart_e, first ? 0 : 1);
            hipLaunchKernelGGL(k_lmm_wgrad, dim3((unsigned)s.gd.quads_per_net, (unsigned)S), dim3(64), 0, st, s.dec, s.gd, dump_d, ntiles,
                               gpart_d, first ? 0 : 1);
            RNVP_HIP_TRY(hipGetLastError());
        }
    }
    if (!grad_out) {                                   // loss only: the same summation as with gradients (bit-identical loss)
        if (!loss_out) return RNVP_OK;
        LGeo none = s.ge;
        none.nnets = 0;
        hipLaunchKernelGGL(k_lmm_reduce, dim3(1), dim3(256), 0, st, s.enc, none, gpart_e, 0, (const uint8_t *)nullptr, losspart, G, inv_B,
                           (float *)nullptr, loss_out);
        RNVP_HIP_TRY(hipGetLastError());
        return RNVP_OK;
    }
    hipLaunchKernelGGL(k_lmm_reduce, dim3((unsigned)(s.enc.npn / 256 + 2)), dim3(256), 0, st, s.enc, s.ge, gpart_e, S,
                       (const uint8_t *)nullptr, losspart, G, inv_B, grad_out, loss_out);
    hipLaunchKernelGGL(k_lmm_reduce, dim3((unsigned)(s.dec.npn / 256 + 2)), dim3(256), 0, st, s.dec, s.gd, gpart_d, S,
                       (const uint8_t *)nullptr, losspart, G, inv_B, grad_out + s.pe, (float *)nullptr);
    RNVP_HIP_TRY(hipGetLastError());
    return RNVP_OK;
}

int cvae_forward(hipStream_t st, const CvaeK &k, const float *params, bool encode, const float *in, const float *c, int64_t n,
                 float *out0, float *out1, void *ws, size_t ws_bytes) {
    if (!ws || ws_bytes < cvae_workspace_bytes(k, 1)) return RNVP_EWORKSPACE;
    const CvaeL s = make_cvae_l(k);
    char *w = static_cast<char *>(ws);
    float *packed_e = reinterpret_cast<float *>(w); w += align_up((size_t)s.ge.net_floats * sizeof(float), 256);
    float *packed_d = reinterpret_cast<float *>(w);
    const int G = grid_for((n + 15) / 16);
    if (encode) {
        int rc = pack_net(st, s.enc, s.ge, params, packed_e);
        if (rc) return rc;
        rc = allow_big_lds(reinterpret_cast<const void *>(k_lmm_cvae_mlp<true>), 160 * 1024, g_attr_cvae_enc);
        if (rc) return rc;
        KernelTimer timer(st, RNVP_PROFILE_FORWARD);
        hipLaunchKernelGGL(k_lmm_cvae_mlp<true>, dim3(G), dim3(64 * kW), s.lds_enc, st, s, packed_e, params, in, c, n, out0, out1);
    } else {
        int rc = pack_net(st, s.dec, s.gd, params + s.pe, packed_d);
        if (rc) return rc;
        rc = allow_big_lds(reinterpret_cast<const void *>(k_lmm_cvae_mlp<false>), 160 * 1024, g_attr_cvae_dec);
        if (rc) return rc;
        KernelTimer timer(st, RNVP_PROFILE_INVERSE);
        hipLaunchKernelGGL(k_lmm_cvae_mlp<false>, dim3(G), dim3(64 * kW), s.lds_dec, st, s, packed_d, params, in, c, n, out0, out1);
    }
    RNVP_HIP_TRY(hipGetLastError());
    return RNVP_OK;
}

}  // namespace lmm
}  // namespace rnvp
