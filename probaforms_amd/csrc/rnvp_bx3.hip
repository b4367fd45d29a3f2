// rnvp_bx3.hip -- forward (+ log-det + prior) / inverse / sampling kernels of the coupling stack with the first
// Linear of the s/t nets on split-bf16 MFMA and LDS-staged weights (gfx950).  Geometry and rationale: rnvp_bx3.h.
// Replaces the same reference code as rnvp_mfma.hip: RealNVPLayer.f / .g for every layer and the loops of
// NormalizingFlow.log_prob / .sample (/root/reference/probaforms/models/realnvp.py:91-101,120-129;
// nflow.py:107-117,141-145).
#include "rnvp_bx3.h"
#include "rnvp_mfma_layer.h"
#include "rnvp_prior.h"

#ifndef RNVP_WPE
#define RNVP_WPE 2
#endif

namespace rnvp {
namespace bx3 {
namespace {

using mfma::f4;
using mfma::Geo;
using bf8 = __attribute__((ext_vector_type(8))) __bf16;
using u4 = __attribute__((ext_vector_type(4))) unsigned;

__device__ __forceinline__ f4 mfma32(f4 a, f4 b, f4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c, 0, 0, 0);
}

// x = t1 + t2 + t3 with three bf16 terms (returned as the upper halves' bit patterns); t1, t2 by truncation,
// so the residuals are exact; t3 rounded to nearest-even (weights only; the kernel truncates its own third term)
__device__ __forceinline__ void split3_rne(float w, uint32_t &a1, uint32_t &a2, uint32_t &a3) {
    const uint32_t u = __float_as_uint(w);
    a1 = u >> 16;
    const float r1 = w - __uint_as_float(u & 0xffff0000u);
    const uint32_t u1 = __float_as_uint(r1);
    a2 = u1 >> 16;
    const float r2 = r1 - __uint_as_float(u1 & 0xffff0000u);
    const uint32_t u2 = __float_as_uint(r2);
    a3 = (u2 + 0x7fffu + ((u2 >> 16) & 1u)) >> 16;
}

// ---- weight packing: one thread per dword of the staged image (runs at the head of every call) -----------------
__global__ void __launch_bounds__(256)
k_pack_bx3(KShape k, Geo3 g, const float *__restrict__ params, uint32_t *__restrict__ packed) {
    const int64_t stage_total = (int64_t)k.L * 2 * g.NCH * g.SD;
    const int64_t total = stage_total + (int64_t)k.L * g.b2_floats;
    const int nin = k.d + k.c, h = k.nout[0], NF = g.NF, CQ = g.CQ;
    const float sc = k.act == RNVP_ACT_TANH ? mfma::kTanhScale : 1.0f;      // tanh: pre-scaled, see tanh4
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        if (t >= stage_total) {                              // bias2 [layer][ot][q][4]
            const int j = (int)(t - stage_total);
            const int l = j / g.b2_floats, jj = j % g.b2_floats;
            const int pc = (l + k.alt) & 1;
            const int ro = jj & 3, qo = (jj >> 2) & 3, ot = jj >> 4;
            const int OTL = NF >= 4 ? NF / 4 : 1;
            int f, net;
            if (NF >= 4) { net = ot / OTL; f = 4 * (ot % OTL) + ro; }
            else { net = ro >> 1; f = ro & 1; }
            const int feat = mfma::feat_trans(NF, qo, f, pc);
            const float v = feat < k.d ? params[(size_t)l * 2 * k.npn + net * k.npn + k.boff[1] + feat] : 0.f;
            packed[t] = __float_as_uint(v);
            continue;
        }
        const int64_t s = t / g.SD;
        const int o = (int)(t - s * g.SD);
        const int l = (int)(s / (2 * g.NCH)), rem = (int)(s % (2 * g.NCH));
        const int net = rem / g.NCH, ch = rem % g.NCH;
        const int pc = (l + k.alt) & 1;
        const float *pn = params + (size_t)l * 2 * k.npn + (size_t)net * k.npn;
        uint32_t out = 0u;
        if (o < g.TC * g.tile_dw) {
            const int tt = o / g.tile_dw, o2 = o % g.tile_dw;
            const int tile = ch * g.TC + tt;
            if (o2 < g.NI * 256) {                           // GEMM1 fragments: two bf16 slots per dword
                const int ni = o2 >> 8, lane = (o2 & 255) >> 2, e = o2 & 3;
                const int D = 4 * ni + e, kk = D / 3, p = D % 3;
                const int q = lane >> 4, hid = 16 * tile + (lane & 15);
                int col = -1;
                if (kk < NF) { const int feat = mfma::feat_cond(NF, q, kk, pc); col = feat < k.d ? feat : -1; }
                else if (kk < g.KS1) { const int ci = q * CQ + (kk - NF); col = ci < k.c ? k.d + ci : -1; }
                if (tile < g.HT && hid < h && col >= 0) {
                    uint32_t a1, a2, a3;
                    split3_rne(sc * pn[k.woff[0] + hid * nin + col], a1, a2, a3);
                    out = p == 0 ? (a1 | (a1 << 16)) : (p == 1 ? (a2 | (a1 << 16)) : (a2 | (a3 << 16)));
                }
            } else if (g.g2b) {                              // GEMM2 fragments, split bf16: [otl][3][lane][4 dwords]
                const int j = o2 - g.NI * 256;
                const int fi = j >> 8, lane = (j & 255) >> 2, e = j & 3;
                const int otl = fi / 3, D = 4 * (fi % 3) + e, v = D / 3, p = D % 3;
                const int q = lane >> 4, i = lane & 15, hid = 16 * tile + 4 * q + v;
                const int feat = mfma::feat_trans(NF, i >> 2, 4 * otl + (i & 3), pc);
                if (tile < g.HT && hid < h && feat < k.d) {
                    uint32_t a1, a2, a3;
                    split3_rne(pn[k.woff[1] + feat * h + hid], a1, a2, a3);
                    out = p == 0 ? (a1 | (a1 << 16)) : (p == 1 ? (a2 | (a1 << 16)) : (a2 | (a3 << 16)));
                }
            } else {                                         // GEMM2 fragments (f32), as rnvp_mfma.hip packs them
                const int j = o2 - g.NI * 256;
                const int ai = j >> 8, lane = (j & 255) >> 2, rho = j & 3;
                const int q = lane >> 4, hid = 16 * tile + 4 * q + rho;
                int feat;
                if (NF == 2) { const int i = lane & 3; feat = mfma::feat_trans(NF, 2 * ai + (i >> 1), i & 1, pc); }
                else { const int i = lane & 15; feat = mfma::feat_trans(NF, i >> 2, 4 * ai + (i & 3), pc); }
                if (tile < g.HT && hid < h && feat < k.d) out = __float_as_uint(pn[k.woff[1] + feat * h + hid]);
            }
        } else {                                             // bias1 [tile][q][4] (f32, pre-scaled)
            const int j = o - g.TC * g.tile_dw;
            if (j < g.TC * 16) {
                const int tt = j >> 4, q = (j >> 2) & 3, e = j & 3;
                const int tile = ch * g.TC + tt, hid = 16 * tile + 4 * q + e;
                if (tile < g.HT && hid < h) out = __float_as_uint(sc * pn[k.boff[0] + hid]);
            }
        }
        packed[t] = out;
    }
}

#ifndef RNVP_BX3_PIPE_MAX_NI
#define RNVP_BX3_PIPE_MAX_NI 4
#endif
constexpr int kPipeMaxNI = RNVP_BX3_PIPE_MAX_NI;

template <int NF, int CQ> struct D3 {
    static constexpr int KS1 = NF + CQ;
    static constexpr int NI = (3 * KS1 + 3) / 4;
    static constexpr int OTL = NF >= 4 ? NF / 4 : 1;
    static constexpr bool G2B = g2b_for(NF);
    static constexpr int NA2 = NF == 2 ? 2 : (G2B ? 3 * OTL : OTL);
    static constexpr int NT2 = NF >= 4 ? 2 * OTL : 1;
};

// the three bf16 terms of an accumulator (4 values) as the B fragments of the next MFMA: value v owns dwords
// 3v .. 3v+2 = (b1, b2), (b1, b3), (b2, b1), the slot order of rnvp_bx3.h
__device__ __forceinline__ void split_acc(f4 hv, f4 (&fr)[3]) {
    uint32_t dw[12];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const uint32_t u = __float_as_uint(hv[v]);
        const float r1 = hv[v] - __uint_as_float(u & 0xffff0000u);
        const uint32_t u1 = __float_as_uint(r1);
        const float r2 = r1 - __uint_as_float(u1 & 0xffff0000u);
        const uint32_t u2 = __float_as_uint(r2);
        dw[3 * v + 0] = __builtin_amdgcn_perm(u1, u, 0x07060302u);
        dw[3 * v + 1] = __builtin_amdgcn_perm(u2, u, 0x07060302u);
        dw[3 * v + 2] = __builtin_amdgcn_perm(u, u1, 0x07060302u);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
        fr[i] = f4{__uint_as_float(dw[4 * i]), __uint_as_float(dw[4 * i + 1]), __uint_as_float(dw[4 * i + 2]),
                   __uint_as_float(dw[4 * i + 3])};
}

// B operand of GEMM1 for the whole layer: the three bf16 terms of every input value, in the slot order of rnvp_bx3.h
template <int NF, int CQ, int PC, int R>
__device__ __forceinline__ void build_bin(const float (&xr)[R][2 * NF], const float (&cr)[R][CQ > 0 ? CQ : 1],
                                          f4 (&bin)[R][D3<NF, CQ>::NI]) {
    constexpr int NI = D3<NF, CQ>::NI, KS1 = D3<NF, CQ>::KS1;
#pragma unroll
    for (int rt = 0; rt < R; ++rt) {
        uint32_t dw[4 * NI];
#pragma unroll
        for (int i = 0; i < 4 * NI; ++i) dw[i] = 0u;
#pragma unroll
        for (int kk = 0; kk < KS1; ++kk) {
            const float v = mfma::in_op<NF, CQ, PC, R>(xr, cr, rt, kk);
            const uint32_t u = __float_as_uint(v);
            const float r1 = v - __uint_as_float(u & 0xffff0000u);
            const uint32_t u1 = __float_as_uint(r1);
            const float r2 = r1 - __uint_as_float(u1 & 0xffff0000u);
            uint32_t u2 = __float_as_uint(r2);
            u2 += 0x7fffu + ((u2 >> 16) & 1u);                                // third term rounded to nearest-even
            dw[3 * kk + 0] = __builtin_amdgcn_perm(u1, u, 0x07060302u);      // (b1, b2)
            dw[3 * kk + 1] = __builtin_amdgcn_perm(u2, u, 0x07060302u);      // (b1, b3)
            dw[3 * kk + 2] = __builtin_amdgcn_perm(u, u1, 0x07060302u);      // (b2, b1)
        }
#pragma unroll
        for (int i = 0; i < NI; ++i)
            bin[rt][i] = f4{__uint_as_float(dw[4 * i]), __uint_as_float(dw[4 * i + 1]), __uint_as_float(dw[4 * i + 2]),
                            __uint_as_float(dw[4 * i + 3])};
    }
}

__device__ __forceinline__ f4 lds_f4(const uint32_t *p) { return *reinterpret_cast<const f4 *>(p); }

// One stage (ntiles hidden tiles of net NET) from the LDS image `sb`.  Software pipeline as rnvp_mfma_layer.h:
// GEMM1 of tile t+1 issues while the activation of tile t runs on the VALU; fragments come from LDS one tile ahead.
// X4 (NF == 2): GEMM2 as 16 independent 4x4x1 blocks into outx[rt][2*NET + o]; otherwise 16x16x4 into out[rt][NET*OTL + o].
template <int NF, int CQ, int R, int NET, int ACT, int NOUT>
__device__ __forceinline__ void run_stage(const uint32_t *sb, const Geo3 &g, int lane, int ntiles,
                                          const f4 (&bin)[R][D3<NF, CQ>::NI], f4 (&out)[R][NOUT]) {
    using D = D3<NF, CQ>;
    constexpr int NI = D::NI, NA2 = D::NA2, OTL = D::OTL;
    constexpr bool X4 = NF == 2;
    const int q = lane >> 4;
    const uint32_t *pT = sb + lane * 4;                                  // + t * tile_dw + frag * 256
    const uint32_t *pB1 = sb + g.TC * g.tile_dw + q * 4;                 // + t * 16
    const int tdw = g.tile_dw;
    struct St { f4 a1n[NI], b1n, a2c[NA2], acc[R]; };
    const int last = ntiles - 1;
    auto gemm1 = [&](const f4 (&a)[NI], f4 b1, int rt) {
        f4 acc = mfma32(a[0], bin[rt][0], b1);
#pragma unroll
        for (int i = 1; i < NI; ++i) acc = mfma32(a[i], bin[rt][i], acc);
        return acc;
    };
    static_assert(!D::G2B || NI > kPipeMaxNI, "the split-bf16 GEMM2 lives in the read-before-use loop");
    if constexpr (NI > kPipeMaxNI) {
        // wide inputs (many GEMM1 fragments per tile): holding two pipeline states in registers would spill; read each
        // tile's fragments from LDS right before use and let the SIMD's other wave cover the latency
        for (int t = 0; t < ntiles; ++t) {
            f4 hv[R];
            {
                f4 a1c[NI];
#pragma unroll
                for (int i = 0; i < NI; ++i) a1c[i] = lds_f4(pT + t * tdw + i * 256);
                const f4 b1c = lds_f4(pB1 + t * 16);
#pragma unroll
                for (int rt = 0; rt < R; ++rt) hv[rt] = mfma::act4<ACT>(gemm1(a1c, b1c, rt));
            }
            f4 a2c[NA2];
#pragma unroll
            for (int o = 0; o < NA2; ++o) a2c[o] = lds_f4(pT + t * tdw + (NI + o) * 256);
            if constexpr (D::G2B) {
#pragma unroll
                for (int rt = 0; rt < R; ++rt) {
                    f4 hb[3];
                    split_acc(hv[rt], hb);
#pragma unroll
                    for (int o = 0; o < OTL; ++o)
#pragma unroll
                        for (int i3 = 0; i3 < 3; ++i3)
                            out[rt][NET * OTL + o] = mfma32(a2c[3 * o + i3], hb[i3], out[rt][NET * OTL + o]);
                }
            } else {
#pragma unroll
                for (int o = 0; o < OTL; ++o)
#pragma unroll
                    for (int rho = 0; rho < 4; ++rho)
#pragma unroll
                        for (int rt = 0; rt < R; ++rt)
                            out[rt][NET * OTL + o] = mfma::mfma16(a2c[o][rho], hv[rt][rho], out[rt][NET * OTL + o]);
            }
        }
        return;
    }
    St s0, s1;
    {
        f4 a1c[NI], b1c;
#pragma unroll
        for (int i = 0; i < NI; ++i) a1c[i] = lds_f4(pT + i * 256);
        b1c = lds_f4(pB1);
#pragma unroll
        for (int o = 0; o < NA2; ++o) s0.a2c[o] = lds_f4(pT + (NI + o) * 256);
        const int t1 = last < 1 ? last : 1;
#pragma unroll
        for (int i = 0; i < NI; ++i) s0.a1n[i] = lds_f4(pT + t1 * tdw + i * 256);
        s0.b1n = lds_f4(pB1 + t1 * 16);
#pragma unroll
        for (int rt = 0; rt < R; ++rt) s0.acc[rt] = gemm1(a1c, b1c, rt);
    }
    auto gemm2 = [&](const St &c, const f4 (&hv)[R]) {
        if constexpr (X4) {
#pragma unroll
            for (int rho = 0; rho < 4; ++rho)
#pragma unroll
                for (int o = 0; o < 2; ++o)
#pragma unroll
                    for (int rt = 0; rt < R; ++rt) out[rt][2 * NET + o] = mfma::mfma4(c.a2c[o][rho], hv[rt][rho], out[rt][2 * NET + o]);
        } else {
#pragma unroll
            for (int o = 0; o < OTL; ++o)
#pragma unroll
                for (int rho = 0; rho < 4; ++rho)
#pragma unroll
                    for (int rt = 0; rt < R; ++rt)
                        out[rt][NET * OTL + o] = mfma::mfma16(c.a2c[o][rho], hv[rt][rho], out[rt][NET * OTL + o]);
        }
    };
    auto step = [&](const St &c, St &nx, int t) {
        const int t2 = (t + 2 < last) ? t + 2 : last;
#pragma unroll
        for (int i = 0; i < NI; ++i) nx.a1n[i] = lds_f4(pT + t2 * tdw + i * 256);
        nx.b1n = lds_f4(pB1 + t2 * 16);
#pragma unroll
        for (int o = 0; o < NA2; ++o) nx.a2c[o] = lds_f4(pT + (t + 1) * tdw + (NI + o) * 256);
        __builtin_amdgcn_sched_barrier(0);
        f4 hv[R];
#pragma unroll
        for (int rt = 0; rt < R; ++rt) nx.acc[rt] = gemm1(c.a1n, c.b1n, rt);      // GEMM1 of tile t+1 (matrix pipe) ...
#pragma unroll
        for (int rt = 0; rt < R; ++rt) hv[rt] = mfma::act4<ACT>(c.acc[rt]);       // ... under the activation of tile t
        __builtin_amdgcn_sched_barrier(0);
        gemm2(c, hv);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto finish = [&](const St &c) {
        f4 hv[R];
#pragma unroll
        for (int rt = 0; rt < R; ++rt) hv[rt] = mfma::act4<ACT>(c.acc[rt]);
        gemm2(c, hv);
    };
    int t = 0;
    for (; t + 1 < last; t += 2) { step(s0, s1, t); step(s1, s0, t + 1); }
    if (t < last) { step(s0, s1, t); finish(s1); } else finish(s0);
}

// LDS-DMA of one stage: the 8 waves copy its NP 1-KiB pieces (piece = one wave-instruction: wave-uniform LDS base +
// lane * 16, exactly the fragment layout)
template <int WAVES>
__device__ __forceinline__ void issue_stage(const uint32_t *gsrc, uint32_t *ldst, int NP, int wave, int lane) {
    for (int p = wave; p < NP; p += WAVES) {
        const uint32_t *gp = gsrc + (size_t)p * 256 + lane * 4;
        uint32_t *lp = ldst + p * 256;        // wave-uniform; the hardware adds lane * 16
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) uint32_t *)gp,
                                         (__attribute__((address_space(3))) uint32_t *)lp, 16, 0, 0);
    }
}

// the prior draw of a fused sampling call, out of line: its constants and Philox state stay out of the flow kernel's register
// allocation (inlined, the compiler hoists them across the stage loop and spills them)
__device__ __attribute__((noinline)) f4 prior_normal4_cold(uint64_t seed, int64_t row, int blk) {
    float z[4];
    prior_normal4(seed, row, blk, z);
    return f4{z[0], z[1], z[2], z[3]};
}

// STAGED: the workgroup's WAVES waves share each stage through LDS (above).  !STAGED (d <= 16, where a stage is small and
// the barrier-paced schedule loses more than the shared copy saves): every wave reads the same stage images straight from
// global memory, one hidden tile ahead, like the f32 kernels of rnvp_mfma_layer.h -- no LDS, no barriers.
template <int NF, int CQ, int R, bool INVERSE, int ACT, int WAVES, bool STAGED>
__global__ void __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(RNVP_WPE, RNVP_WPE)))
k_flow_bx3(const uint32_t *__restrict__ wp, Geo3 g, int L, int alt, const float *x, const float *__restrict__ c,
           const int64_t *__restrict__ row_index, int64_t n, float *out_x, float *logdet_out, float *logp_out,
           float *part, uint64_t seed, int64_t row0) {
    using D = D3<NF, CQ>;
    constexpr int DD = 8 * NF, CD = 4 * CQ, NI = D::NI, OTL = D::OTL, NT2 = D::NT2;
    constexpr bool X4 = NF == 2;
    constexpr int NOUT = X4 ? 4 : NT2;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];       // two stage buffers of g.SD dwords
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // known uniform: row bases in SGPRs
    const int q = lane >> 4, r = lane & 15;
    const int64_t rows_per_wg = (int64_t)WAVES * R * 16;
    const int64_t ngroups = (n + rows_per_wg - 1) / rows_per_wg;
    const float prior_c = 0.5f * (float)g.d * kLog2Pi;
    const bool full = (g.d == DD) && (g.c == CD) && (((uintptr_t)x | (uintptr_t)out_x) & 15) == 0;
    const bool gen = INVERSE && x == nullptr;
    const int nstages = L * 2 * g.NCH;
    const float *b2base = reinterpret_cast<const float *>(wp + (size_t)nstages * g.SD);
    // stage si of a pass: layer position lp = si / (2 NCH) (inverse walks the layers backwards), net, chunk
    auto stage_src = [&](int si) {
        const int lp = si / (2 * g.NCH), rem = si % (2 * g.NCH);
        const int l = INVERSE ? L - 1 - lp : lp;
        return wp + ((size_t)l * 2 * g.NCH + rem) * g.SD;
    };
    float wave_sum = 0.f;
    if constexpr (STAGED) {
        if ((int64_t)blockIdx.x < ngroups) issue_stage<WAVES>(stage_src(0), lds, g.NP, wave, lane);
    }
    for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int64_t base = grp * rows_per_wg + (int64_t)wave * R * 16;
        float xr[R][2 * NF], cr[R][CQ > 0 ? CQ : 1], ld[R];
#pragma unroll
        for (int rt = 0; rt < R; ++rt) {
            const int64_t row = base + rt * 16 + r;
            const bool valid = row < n;
            const int64_t src = valid ? (row_index ? row_index[row] : row) : 0;
            if (gen) {
                mfma::load_row<NF, CQ, false>(x, c, src, g.d, g.c, full, q, xr[rt], cr[rt]);
#pragma unroll
                for (int b = 0; b < 2 * NF / 4; ++b) {
                    const f4 z4 = prior_normal4_cold(seed, row0 + row, (q * 2 * NF) / 4 + b);
#pragma unroll
                    for (int e = 0; e < 4; ++e) xr[rt][4 * b + e] = (q * 2 * NF + 4 * b + e < g.d) ? z4[e] : 0.f;
                }
            } else {
                mfma::load_row<NF, CQ>(x, c, src, g.d, g.c, full, q, xr[rt], cr[rt]);
            }
            ld[rt] = 0.f;
        }
        const bool more = grp + gridDim.x < ngroups; (void)more;
        f4 bin[R][NI];
        f4 out[R][NOUT];
        for (int si = 0; si < nstages; ++si) {
            // stage si has landed (this wave's pieces: vmcnt; every wave's: the barrier), and every wave is done
            // reading the other buffer, which the next stage may now overwrite
            if constexpr (STAGED) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (si + 1 < nstages) issue_stage<WAVES>(stage_src(si + 1), lds + ((si + 1) & 1) * g.SD, g.NP, wave, lane);
                else if (more) issue_stage<WAVES>(stage_src(0), lds + ((si + 1) & 1) * g.SD, g.NP, wave, lane);
            }
            const int lp = si / (2 * g.NCH), rem = si % (2 * g.NCH);
            const int l = INVERSE ? L - 1 - lp : lp;
            const int net = rem / g.NCH, ch = rem % g.NCH;
            const int pc = (l + alt) & 1;
            const int nt = (g.HT - ch * g.TC) < g.TC ? (g.HT - ch * g.TC) : g.TC;
            if (rem == 0) {                                  // layer start: GEMM1's B operand, accumulators
                if constexpr (!STAGED && CQ > 0) {
                    // the conditions' bf16 terms are the same in every layer: left alone, the compiler computes them once per row
                    // group, finds no register for them and parks them in scratch (written through to HBM, re-read per layer);
                    // 7 VALU operations per value and layer instead
#pragma unroll
                    for (int rt = 0; rt < R; ++rt)
#pragma unroll
                        for (int u = 0; u < CQ; ++u) asm volatile("" : "+v"(cr[rt][u]));
                }
                if (pc) build_bin<NF, CQ, 1, R>(xr, cr, bin); else build_bin<NF, CQ, 0, R>(xr, cr, bin);
#pragma unroll
                for (int rt = 0; rt < R; ++rt)
#pragma unroll
                    for (int u = 0; u < NOUT; ++u) {
                        if constexpr (X4) out[rt][u] = f4{0.f, 0.f, 0.f, 0.f};
                        else out[rt][u] = *reinterpret_cast<const f4 *>(b2base + (size_t)l * g.b2_floats + (u * 4 + q) * 4);
                    }
            }
            if constexpr (STAGED) {
                const uint32_t *sb = lds + (si & 1) * g.SD;
                if (net == 0) run_stage<NF, CQ, R, 0, ACT, NOUT>(sb, g, lane, nt, bin, out);
                else run_stage<NF, CQ, R, 1, ACT, NOUT>(sb, g, lane, nt, bin, out);
            } else {
                const uint32_t *__restrict__ sb = stage_src(si);
                if (net == 0) run_stage<NF, CQ, R, 0, ACT, NOUT>(sb, g, lane, nt, bin, out);
                else run_stage<NF, CQ, R, 1, ACT, NOUT>(sb, g, lane, nt, bin, out);
            }
            if (rem == 2 * g.NCH - 1) {                      // layer end: affine update of the transformed features
                f4 bias2 = f4{0.f, 0.f, 0.f, 0.f};
                if constexpr (X4) bias2 = *reinterpret_cast<const f4 *>(b2base + (size_t)l * g.b2_floats + q * 4);
#pragma unroll
                for (int rt = 0; rt < R; ++rt) {
                    float tvv[NF], svv[NF];
                    if constexpr (X4) {
                        float tv[2], sv[2];
                        mfma::reduce_scatter_x4(out[rt], tv, sv);
                        tvv[0] = tv[0] + bias2[0]; tvv[1] = tv[1] + bias2[1];
                        svv[0] = sv[0] + bias2[2]; svv[1] = sv[1] + bias2[3];
                    } else {
#pragma unroll
                        for (int f = 0; f < NF; ++f) { tvv[f] = out[rt][f >> 2][f & 3]; svv[f] = out[rt][OTL + (f >> 2)][f & 3]; }
                    }
#pragma unroll
                    for (int f = 0; f < NF; ++f) {
                        // pc is wave-uniform: select the register of the transformed feature without dynamic indexing
                        const float xe0 = xr[rt][2 * f], xe1 = xr[rt][2 * f + 1];
                        const float xv = pc ? xe0 : xe1;
                        float nv;
                        if (INVERSE) nv = (xv - tvv[f]) * expf(-svv[f]);
                        else { nv = fmaf(xv, expf(svv[f]), tvv[f]); ld[rt] += svv[f]; }
                        xr[rt][2 * f] = pc ? nv : xe0;
                        xr[rt][2 * f + 1] = pc ? xe1 : nv;
                    }
                }
            }
        }
        // (the epilogue derives its rows from an opaque copy of the lane's row number: nothing row-related stays live -- or gets
        // spilled -- across the stage loop)
        int r_out = r, q_out = q;
        asm volatile("" : "+v"(r_out), "+v"(q_out));
#pragma unroll
        for (int rt = 0; rt < R; ++rt) {
            const int64_t row = base + rt * 16 + r_out;
            const bool valid = row < n;
            if (out_x && valid) mfma::store_row<NF>(out_x, row, g.d, full, q_out, xr[rt]);
            if (!INVERSE) {
                float ss = 0.f;
#pragma unroll
                for (int v = 0; v < 2 * NF; ++v) ss = fmaf(xr[rt][v], xr[rt][v], ss);
                float l1 = ld[rt];
                l1 += __shfl_xor(l1, 16); l1 += __shfl_xor(l1, 32);
                ss += __shfl_xor(ss, 16); ss += __shfl_xor(ss, 32);
                const float lpv = l1 + (-0.5f * ss - prior_c);          // nflow.py:115
                if (valid && q_out == 0) {
                    if (logdet_out) logdet_out[row] = l1;
                    if (logp_out) logp_out[row] = lpv;
                }
                if (part) {
                    float v = (valid && q_out == 0) ? lpv : 0.f;
                    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
                    wave_sum += v;
                }
            }
        }
    }
    if (!INVERSE && part && lane == 0) part[blockIdx.x * WAVES + wave] = wave_sum;
}

#ifndef RNVP_BX3_R8
#define RNVP_BX3_R8 2
#endif
#ifndef RNVP_BX3_R4
#define RNVP_BX3_R4 4
#endif
#ifndef RNVP_BX3_R2
#define RNVP_BX3_R2 2
#endif
// Row tiles per wave, measured on MI355X at 1M rows (forward / inverse ms): C4 (NF 8) R = 1 / 2 / 3 / 4: 3.0 / 2.51 /
// 2.75 / 3.35 (spills); C3 (NF 4) read-before-use loop R = 2 / 3 / 4: 4.24 / 4.03 / 3.80, pipelined R = 2: 3.93.
template <int NF, int CQ> struct RowsBx3 { static constexpr int value = NF == 8 ? RNVP_BX3_R8 : (NF == 4 ? RNVP_BX3_R4 : RNVP_BX3_R2); };

#ifndef RNVP_BX3_DIRECT_R2
#define RNVP_BX3_DIRECT_R2 3
#endif
// d <= 16: the barrier-free form (4 waves, stages read from global memory); wider rows: LDS-staged, 8 waves
template <int NF> struct DirectBx3 { static constexpr bool value = NF == 2 && RNVP_BX3_DIRECT_R2 > 0; };

template <int NF, int CQ, bool INVERSE, int ACT>
int launch(hipStream_t st, const KShape &k, const Geo3 &g, const uint32_t *packed, const float *x, const float *c,
           const int64_t *row_index, int64_t n, float *out_x, float *logdet_out, float *logp_out, float *part,
           int *grid_out, int *waves_out, uint64_t seed, int64_t row0) {
    constexpr bool DIRECT = DirectBx3<NF>::value;
    constexpr int R = DIRECT ? (RNVP_BX3_DIRECT_R2 > 0 ? RNVP_BX3_DIRECT_R2 : 1) : RowsBx3<NF, CQ>::value;
    constexpr int WAVES = DIRECT ? 4 : kWavesBx3;
    auto kern = k_flow_bx3<NF, CQ, R, INVERSE, ACT, WAVES, !DIRECT>;
    static std::atomic<uint64_t> attr_done{0};
    const size_t lds_bytes = DIRECT ? 0 : (size_t)2 * g.SD * sizeof(uint32_t);
    if (!DIRECT) {
        const int arc = allow_big_lds(reinterpret_cast<const void *>(kern), 160 * 1024, attr_done);
        if (arc) return arc;
    }
    const int64_t rows_per_wg = (int64_t)WAVES * R * 16;
    const int64_t ngroups = (n + rows_per_wg - 1) / rows_per_wg;
    const int maxgrid = DIRECT ? kMaxGridBx3 * (kWavesBx3 / 4) : kMaxGridBx3;          // same bound on grid * waves
    const int grid = (int)(ngroups < maxgrid ? ngroups : maxgrid);
    if (grid_out) *grid_out = grid;
    if (waves_out) *waves_out = WAVES;
    note_dispatch(INVERSE ? RNVP_PROFILE_INVERSE : RNVP_PROFILE_FORWARD, "k_flow_bx3", DIRECT ? RNVP_VARIANT_BX3_DIRECT : RNVP_VARIANT_BX3_STAGED,
                  R, WAVES, grid, RNVP_PREC_BX3, n);
    {
        const KernelEvents ev(INVERSE ? RNVP_PROFILE_INVERSE : RNVP_PROFILE_FORWARD);
        hipExtLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds_bytes, st, ev.start, ev.stop, 0, packed, g, k.L, k.alt,
                              x, c, row_index, n, out_x, logdet_out, logp_out, part, seed, row0);
    }
    RNVP_HIP_TRY(hipGetLastError());
    return RNVP_OK;
}

template <bool INVERSE>
int dispatch(hipStream_t st, const KShape &k, const Geo3 &g, const uint32_t *packed, const float *x, const float *c,
             const int64_t *row_index, int64_t n, float *out_x, float *logdet_out, float *logp_out, float *part,
             int *grid_out, int *waves_out, uint64_t seed, int64_t row0) {
#define RNVP_CASE(nf, cq)                                                                                          \
    if (g.NF == nf && g.CQ == cq) {                                                                                \
        if (k.act == RNVP_ACT_TANH)                                                                                \
            return launch<nf, cq, INVERSE, 0>(st, k, g, packed, x, c, row_index, n, out_x, logdet_out, logp_out, part, \
                                              grid_out, waves_out, seed, row0);                                    \
        return launch<nf, cq, INVERSE, 1>(st, k, g, packed, x, c, row_index, n, out_x, logdet_out, logp_out, part,    \
                                          grid_out, waves_out, seed, row0);                                        \
    }
    RNVP_CASE(2, 1) RNVP_CASE(2, 0) RNVP_CASE(4, 2) RNVP_CASE(8, 4)
#undef RNVP_CASE
    return RNVP_EUNSUPPORTED;
}

int pack(hipStream_t st, const KShape &k, const Geo3 &g, const float *params, uint32_t *packed) {
    const int64_t total = (int64_t)k.L * 2 * g.NCH * g.SD + (int64_t)k.L * g.b2_floats;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_pack_bx3, dim3(blocks), dim3(256), 0, st, k, g, params, packed);
    RNVP_HIP_TRY(hipGetLastError());
    return RNVP_OK;
}

}  // namespace

bool supported(const KShape &k) { return mfma::supported(k); }

size_t packed_bytes(const KShape &k) {
    const Geo3 g = make_geo3(k.d, k.c, k.nout[0]);
    return align_up(((size_t)k.L * 2 * g.NCH * g.SD + (size_t)k.L * g.b2_floats) * sizeof(uint32_t), 256);
}

int forward(hipStream_t st, const KShape &k, const float *params, const float *x, const float *c,
            const int64_t *row_index, int64_t n, float *z_out, float *logdet_out, float *logp_out, float *part,
            int *grid_out, int *waves_out, void *packed) {
    const Geo3 g = make_geo3(k.d, k.c, k.nout[0]);
    int rc = pack(st, k, g, params, static_cast<uint32_t *>(packed));
    if (rc) return rc;
    return dispatch<false>(st, k, g, static_cast<const uint32_t *>(packed), x, c, row_index, n, z_out, logdet_out, logp_out,
                           part, grid_out, waves_out, 0, 0);
}

int inverse(hipStream_t st, const KShape &k, const float *params, const float *z, const float *c, int64_t n,
            float *x_out, uint64_t seed, int64_t row0, void *packed) {
    const Geo3 g = make_geo3(k.d, k.c, k.nout[0]);
    int rc = pack(st, k, g, params, static_cast<uint32_t *>(packed));
    if (rc) return rc;
    return dispatch<true>(st, k, g, static_cast<const uint32_t *>(packed), z, c, nullptr, n, x_out, nullptr, nullptr, nullptr,
                          nullptr, nullptr, seed, row0);
}

}  // namespace bx3
}  // namespace rnvp
