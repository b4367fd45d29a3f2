// rnvp_mfma_layer.h -- one coupling layer of the register-chained MFMA path, forward direction
// (shared by the forward / inverse kernels and by the forward phase of the training kernel).
//
// The hidden-tile loop is software-pipelined BY HAND, because hipcc (ROCm 7.2) otherwise
// serialises it: left to itself it computes each tanh just in time for the MFMA that consumes it
// (exp -> add -> rcp -> fma -> mfma on ONE register, ~45 dependent cycles per MFMA) and re-loads
// the weight fragments at the top of every iteration behind an s_waitcnt vmcnt(0).  Structure of
// one iteration for hidden tile t (R row tiles per wave):
//     top    : issue the fragment loads for tile t+2 (GEMM1) and t+1 (GEMM2); their addresses are
//              made opaque so the optimiser cannot sink the loads into the next iteration
//     phase A: for each row tile: 3..12 GEMM1 MFMAs of tile t+1  ||  the 4 tanh of tile t (VALU)
//     phase B: for each row tile: the GEMM2 MFMAs of tile t
// __builtin_amdgcn_sched_barrier(0) between the blocks pins that order; inside a block the MFMAs
// issue first and the VALU work runs in their shadow.
#pragma once
#include "rnvp_mfma.h"

namespace rnvp {
namespace mfma {

using f4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ f4 mfma16(float a, float b, f4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// 16 independent 4x4x1 outer products: D_b[i][j] (VGPR i of lane 4b+j) += A_b[i] (lane 4b+i) * B_b[j]
__device__ __forceinline__ f4 mfma4(float a, float b, f4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
}

// reduce-scatter steps across the 4 lane groups q = lane >> 4 (v_permlane32_swap / v_permlane16_swap):
// lanes of the lower half (q < 2) end with x(own) + x(partner q ^ 2), the upper half with y + y(partner)
__device__ __forceinline__ float swap_add32(float x, float y) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// even 16-lane rows (q & 1 == 0) end with x(own) + x(partner q ^ 1), odd rows with y + y(partner)
__device__ __forceinline__ float swap_add16(float x, float y) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

#ifndef RNVP_ABLATE
#define RNVP_ABLATE 0
#endif
// RNVP_NO_X4: developer switch to fall back to the shared-tile 16x16x4 GEMM2 for d == 16 (A/B timing)
#ifdef RNVP_NO_X4
constexpr bool kUseX4 = false;
#else
constexpr bool kUseX4 = true;
#endif

// tanh(v) = 1 - 2 / (1 + e^{2v}) with u = 2*log2(e)*v arriving PRE-SCALED: k_pack_weights folds
// kTanhScale into the packed W1 and b1, so GEMM1's accumulator already is u.  Per value: one
// v_exp_f32 + one v_rcp_f32 (~1 ulp each) and half a v_pk_add_f32 / v_pk_fma_f32.  f32 MFMA and
// VALU do not overlap on a SIMD (measured: MFMA busy + VALU busy = 99 % of the forward kernel),
// so every VALU instruction saved here is wall time.  Saturates correctly through e = +inf / 0;
// absolute error ~1e-7, the rounding level of values near 1.
constexpr float kTanhScale = 2.8853900817779268f;          // 2 * log2(e)
using f2 = __attribute__((ext_vector_type(2))) float;

__device__ __forceinline__ f4 tanh4(f4 u) {
    if (RNVP_ABLATE & 8) return u * 0.5f;
    f2 e0, e1;
    e0[0] = __builtin_amdgcn_exp2f(u[0]); e0[1] = __builtin_amdgcn_exp2f(u[1]);
    e1[0] = __builtin_amdgcn_exp2f(u[2]); e1[1] = __builtin_amdgcn_exp2f(u[3]);
    e0 = e0 + 1.0f; e1 = e1 + 1.0f;                                           // v_pk_add_f32
    f2 r0, r1;
    r0[0] = __builtin_amdgcn_rcpf(e0[0]); r0[1] = __builtin_amdgcn_rcpf(e0[1]);
    r1[0] = __builtin_amdgcn_rcpf(e1[0]); r1[1] = __builtin_amdgcn_rcpf(e1[1]);
    r0 = __builtin_elementwise_fma(r0, f2{-2.0f, -2.0f}, f2{1.0f, 1.0f});     // v_pk_fma_f32
    r1 = __builtin_elementwise_fma(r1, f2{-2.0f, -2.0f}, f2{1.0f, 1.0f});
    return f4{r0[0], r0[1], r1[0], r1[1]};
}

// Hidden activation of gen_network (realnvp.py:26-37) on a GEMM1 accumulator and its derivative from the
// activation's OUTPUT.  ACT 0: tanh (accumulator pre-scaled, above); ACT 1: ReLU (no pre-scale: k_pack_weights
// leaves W1, b1 as they are), derivative 1 where the output is positive, 0 at and below zero like torch.
template <int ACT> __device__ __forceinline__ f4 act4(f4 u) {
    if constexpr (ACT == 0) return tanh4(u);
    return f4{fmaxf(u[0], 0.f), fmaxf(u[1], 0.f), fmaxf(u[2], 0.f), fmaxf(u[3], 0.f)};
}
template <int ACT> __device__ __forceinline__ f4 dact4(f4 hv) {
    if constexpr (ACT == 0) return 1.0f - hv * hv;
    return f4{hv[0] > 0.f ? 1.f : 0.f, hv[1] > 0.f ? 1.f : 0.f, hv[2] > 0.f ? 1.f : 0.f, hv[3] > 0.f ? 1.f : 0.f};
}

// hide a fragment address from the optimiser (it would otherwise prove that next iteration's
// "current" fragment equals this iteration's prefetch and replace the prefetch by a late re-load)
typedef const __attribute__((address_space(1))) f4 *gf4_ptr;      // keeps the loads global_load (not flat)
__device__ __forceinline__ gf4_ptr opaque(const float *p) {
    asm volatile("" : "+v"(p));
    return (gf4_ptr)p;
}

// Row I/O of lane group q: features q*2NF .. q*2NF + 2NF - 1 and conditions q*CQ .. q*CQ + CQ - 1.
// `full` (sizes equal to the padded tile sizes): float4 loads, 1 KiB per wave; otherwise guarded scalar
// loads that fill the padded slots with zeros.
// WITH_X = false loads the conditions only (the caller fills xr: prior draws made in the kernel).
template <int NF, int CQ, bool WITH_X = true>
__device__ __forceinline__ void load_row(const float *x, const float *__restrict__ c, int64_t src,
                                         int d, int cd, bool full, int q, float (&xr)[2 * NF],
                                         float (&cr)[CQ > 0 ? CQ : 1]) {
    if (full) {
        if constexpr (WITH_X) {
            const float *xp = x + src * (8 * NF) + q * 2 * NF;
#pragma unroll
            for (int v = 0; v < 2 * NF; v += 4) {
                const f4 t = *reinterpret_cast<const f4 *>(xp + v);
                xr[v] = t[0]; xr[v + 1] = t[1]; xr[v + 2] = t[2]; xr[v + 3] = t[3];
            }
        }
        if (CQ > 0) {
            const float *cp = c + src * (4 * CQ) + q * CQ;
#pragma unroll
            for (int v = 0; v < CQ; ++v) cr[v] = cp[v];
        } else {
            cr[0] = 0.f;
        }
    } else {
        if constexpr (WITH_X) {
#pragma unroll
            for (int v = 0; v < 2 * NF; ++v) {
                const int j = q * 2 * NF + v;
                xr[v] = (j < d) ? x[src * d + j] : 0.f;
            }
        }
#pragma unroll
        for (int v = 0; v < (CQ > 0 ? CQ : 1); ++v) {
            const int j = q * CQ + v;
            cr[v] = (CQ > 0 && j < cd) ? c[src * cd + j] : 0.f;
        }
    }
}

template <int NF>
__device__ __forceinline__ void store_row(float *out, int64_t row, int d, bool full, int q, const float (&xr)[2 * NF]) {
    if (full) {
        float *op = out + row * (8 * NF) + q * 2 * NF;
#pragma unroll
        for (int v = 0; v < 2 * NF; v += 4) {
            f4 t;
            t[0] = xr[v]; t[1] = xr[v + 1]; t[2] = xr[v + 2]; t[3] = xr[v + 3];
            *reinterpret_cast<f4 *>(op + v) = t;
        }
    } else {
#pragma unroll
        for (int v = 0; v < 2 * NF; ++v) {
            const int j = q * 2 * NF + v;
            if (j < d) out[row * d + j] = xr[v];
        }
    }
}

constexpr int kTS = 20;                  // row stride (floats) of a 16-wide LDS transposition tile

// Every spin on an LDS counter is bounded: a wait that outlives 2^22 naps (seconds; the longest legitimate one is a few
// microseconds) is a protocol error.  The wave then raises the step's ERROR WORD in the workspace (global memory, behind the loss
// partials; cleared by the pack kernel at the head of every call) and poisons the workgroup's loss partial with a NaN, and ENDS
// instead of hanging the device.  The workgroup's gradient partial is then incomplete: k_train_finish sees the error word,
// applies NO Adam step and NO re-pack -- the parameters keep their last good values -- and reports the loss kProtocolNaN, a NaN
// with a payload no arithmetic produces, which RealNVP.fit turns into a RuntimeError (_engine.check_losses).  (`__builtin_trap()` there cost the C2 kernel 2.5 % and
// the wide C3 kernel 6.5 % of its time through register allocation alone; `s_endpgm` costs nothing: profiles/r05_spin_bound_ab.txt.)
constexpr unsigned kProtocolNaN = RNVP_PROTOCOL_NAN_BITS;       // include/rnvp_hip.h
struct Poison { float *loss; int *flag; int *err; };          // this workgroup's loss partials (global), its LDS error flag, the step's error word (global)
__device__ __forceinline__ void spin_nap(int &spins, const Poison &po) {
    __builtin_amdgcn_s_sleep(1);
    if (++spins > (1 << 22)) {
        if (po.flag) __hip_atomic_store(po.flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // waves that still finish write NaN too
        if (po.err) __hip_atomic_store(po.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (po.loss) *po.loss = __builtin_nanf("");
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_endpgm" ::: "memory");
    }
}
// LDS traffic between lanes of ONE wave: DS operations of a wave execute in order, so only the
// compiler has to be kept from reordering the accesses.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- partial sums over workgroups, stage 1 (shared by the RealNVP and CVAE steps) ---------------------
// G per-workgroup partials of n4 float4 -> kSeg segment sums (segment s adds partials s, s+kSeg, ... in
// index order: deterministic); coalesced float4 streaming.
constexpr int kSeg = 16;

static __global__ void __launch_bounds__(256)
k_sum_segments(const float *__restrict__ gpart, int G, size_t n4, float *__restrict__ seg) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int sgi = blockIdx.y;
    if (i >= n4) return;
    f4 a = f4{0.f, 0.f, 0.f, 0.f};
    const f4 *src = reinterpret_cast<const f4 *>(gpart) + i;
    const int S = gridDim.y;
    int b = sgi;
    // eight loads in flight per thread (the partials are read exactly once: a pure HBM stream), added in index order
    for (; b + 7 * S < G; b += 8 * S) {
        f4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(src + (size_t)(b + u * S) * n4);
#pragma unroll
        for (int u = 0; u < 8; ++u) a += v[u];
    }
    for (; b < G; b += S) a += __builtin_nontemporal_load(src + (size_t)b * n4);
    reinterpret_cast<f4 *>(seg)[(size_t)sgi * n4 + i] = a;
}

// accumulator-layout tile (lane (q, r): features 4q..4q+3 of row r)  ->  k-step operands for a
// contraction over rows (lane (qk, j): feature j of rows 4*ks + qk, ks = 0..3)
__device__ __forceinline__ void transpose16(float *buf, f4 v, int lane, float (&o)[4]) {
    if (RNVP_ABLATE & 1) { o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3]; return; }
    const int q = lane >> 4, r = lane & 15;
    wave_lds_fence();
    *reinterpret_cast<f4 *>(buf + r * kTS + 4 * q) = v;
    wave_lds_fence();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) o[ks] = buf[(4 * ks + q) * kTS + r];
}

template <int NF, int CQ> struct FwdDims {
    static constexpr int KS1 = NF + CQ;
    static constexpr int K4 = (KS1 + 3) / 4;
    static constexpr int OTL = NF >= 4 ? NF / 4 : 1;
    static constexpr int NT2 = NF >= 4 ? 2 * OTL : 1;
};

// B operand of GEMM1 for k-step kk (kk is a compile-time constant after unrolling)
template <int NF, int CQ, int PC, int R>
__device__ __forceinline__ float in_op(const float (&xr)[R][2 * NF], const float (&cr)[R][CQ > 0 ? CQ : 1],
                                       int rt, int kk) {
    return (kk < NF) ? xr[rt][2 * (kk < NF ? kk : 0) + PC] : cr[rt][kk >= NF ? kk - NF : 0];
}

template <int NF, int CQ, int PC, int R>
__device__ __forceinline__ f4 gemm1(const f4 (&a1)[FwdDims<NF, CQ>::K4], f4 b1, const float (&xr)[R][2 * NF],
                                    const float (&cr)[R][CQ > 0 ? CQ : 1], int rt) {
    f4 acc = b1;
#pragma unroll
    for (int kk = 0; kk < NF + CQ; ++kk)
        acc = mfma16(a1[kk >> 2][kk & 3], in_op<NF, CQ, PC, R>(xr, cr, rt, kk), acc);
    return acc;
}

// BX (training kernel, split-GEMM1 form): GEMM1 runs on v_mfma_f32_16x16x32_bf16 with three-term bf16 operands
// (rnvp_split.h) -- the A fragments come from the oA1S section (NI1 per tile instead of K4), the B operand is `bin`,
// the layer's inputs split once per layer and row tile (build_bin).
using split::mfma32;
template <int NF, int CQ> struct SplitDims {
    static constexpr int NI1 = split::n_mfma(NF + CQ);     // bf16 MFMAs per GEMM1 tile
    static constexpr int NI2 = split::n_mfma(NF);          // bf16 MFMAs of g_h = W2^T g_out per tile
};
template <int NF, int CQ, bool BX> struct G1Dims {
    static constexpr int NA = BX ? SplitDims<NF, CQ>::NI1 : FwdDims<NF, CQ>::K4;      // f4 fragments per GEMM1 tile
};
template <int NF, int CQ, int PC, int R>
__device__ __forceinline__ void build_bin(const float (&xr)[R][2 * NF], const float (&cr)[R][CQ > 0 ? CQ : 1],
                                          f4 (&bin)[R][SplitDims<NF, CQ>::NI1]) {
#pragma unroll
    for (int rt = 0; rt < R; ++rt) {
        float v[NF + CQ];
#pragma unroll
        for (int kk = 0; kk < NF + CQ; ++kk) v[kk] = in_op<NF, CQ, PC, R>(xr, cr, rt, kk);
        split::build_b<NF + CQ>(v, bin[rt]);
    }
}
// one GEMM1 step (fragment i of NA) of row tile rt onto acc
template <int NF, int CQ, int PC, int R, bool BX>
__device__ __forceinline__ f4 gemm1_step(const f4 (&a1)[G1Dims<NF, CQ, BX>::NA], int i, const float (&xr)[R][2 * NF],
                                         const float (&cr)[R][CQ > 0 ? CQ : 1],
                                         const f4 (*bin)[SplitDims<NF, CQ>::NI1], int rt, f4 acc) {
    if constexpr (BX) {
        return mfma32(a1[i], bin[rt][i], acc);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (4 * i + e < NF + CQ) acc = mfma16(a1[i][e], in_op<NF, CQ, PC, R>(xr, cr, rt, 4 * i + e), acc);
        return acc;
    }
}
template <int NF, int CQ, int PC, int R, bool BX>
__device__ __forceinline__ f4 gemm1_any(const f4 (&a1)[G1Dims<NF, CQ, BX>::NA], f4 b1, const float (&xr)[R][2 * NF],
                                        const float (&cr)[R][CQ > 0 ? CQ : 1],
                                        const f4 (*bin)[SplitDims<NF, CQ>::NI1], int rt) {
    f4 acc = b1;
#pragma unroll
    for (int i = 0; i < G1Dims<NF, CQ, BX>::NA; ++i) acc = gemm1_step<NF, CQ, PC, R, BX>(a1, i, xr, cr, bin, rt, acc);
    return acc;
}

// The fragments a tile loop starts with (the prologue of run_tiles / run_tiles_x4, f32 forms): the tile-split kernels load
// them for the NEXT layer before the current layer's LDS rendezvous, so that a layer does not open with a dependent
// round trip to memory (those launches are latency chains; the packed block was written by another kernel moments ago and
// the first touch of a line comes from beyond this XCD's L2).
template <int NF, int CQ> struct TilePre {
    static constexpr int K4 = FwdDims<NF, CQ>::K4, NA2 = (NF == 2 && kUseX4) ? 2 : FwdDims<NF, CQ>::OTL;
    f4 a1c[K4], b1c, a2c[NA2], a1n[K4], b1n;
};
template <int NF, int CQ>
__device__ __forceinline__ void load_tile_pre(const float *__restrict__ W, const Geo &g, int lane, int tile0, int ntiles,
                                              TilePre<NF, CQ> &p) {
    constexpr int K4 = TilePre<NF, CQ>::K4, NA2 = TilePre<NF, CQ>::NA2;
    const int q = lane >> 4;
    const float *pA1 = W + g.oA1 + ((size_t)tile0 * K4 * 64 + lane) * 4;
    const float *pB1 = W + g.oB1 + ((size_t)tile0 * 4 + q) * 4;
    const float *pA2 = W + ((NF == 2 && kUseX4) ? g.oA2X : g.oA2) + ((size_t)tile0 * NA2 * 64 + lane) * 4;
    const int t1 = ntiles > 1 ? 1 : 0;
#pragma unroll
    for (int k4 = 0; k4 < K4; ++k4) p.a1c[k4] = *opaque(pA1 + k4 * 256);
    p.b1c = *opaque(pB1);
#pragma unroll
    for (int o = 0; o < NA2; ++o) p.a2c[o] = *opaque(pA2 + o * 256);
#pragma unroll
    for (int k4 = 0; k4 < K4; ++k4) p.a1n[k4] = *opaque(pA1 + ((size_t)t1 * K4 + k4) * 256);
    p.b1n = *opaque(pB1 + t1 * 16);
}

// Tiles [tile0, tile0 + ntiles) of the packed layer feed out tiles OT0 .. OT0 + OTL - 1.
template <int NF, int CQ, int R, int PC, int OT0, int ACT, bool BX = false>
__device__ __forceinline__ void run_tiles(const float *__restrict__ W, const Geo &g, int lane, int tile0,
                                          int ntiles, const float (&xr)[R][2 * NF],
                                          const float (&cr)[R][CQ > 0 ? CQ : 1],
                                          f4 (&out)[R][FwdDims<NF, CQ>::NT2],
                                          const f4 (*bin)[SplitDims<NF, CQ>::NI1] = nullptr,
                                          const TilePre<NF, CQ> *pre = nullptr) {
    using D = FwdDims<NF, CQ>;
    constexpr int K4 = G1Dims<NF, CQ, BX>::NA, OTL = D::OTL;
    const int q = lane >> 4;
    const float *pA1 = W + (BX ? g.oA1S : g.oA1) + ((size_t)tile0 * K4 * 64 + lane) * 4;
    const float *pB1 = W + g.oB1 + ((size_t)tile0 * 4 + q) * 4;
    const float *pA2 = W + g.oA2 + ((size_t)tile0 * OTL * 64 + lane) * 4;
    const int last = ntiles - 1;
    // The pipeline state {pre-activations of tile t, GEMM1 fragments of tile t+1, GEMM2 fragments of tile t}
    // alternates between two named register sets, so an iteration ends without the ~30 v_mov that rotating one
    // set costs (f32 MFMA and VALU share the SIMD: every v_mov is wall time).
    struct St { f4 a1n[K4], b1n, a2c[OTL], acc[R]; };
    St s0, s1;
    {   // prologue: GEMM1 of the first tile; fragments of the second
        f4 a1c[K4], b1c;
        bool have = false;
        if constexpr (!BX && (NF >= 4 || !kUseX4)) {
            if (pre) {          // loaded ahead by the caller (TilePre)
                have = true;
#pragma unroll
                for (int k4 = 0; k4 < K4; ++k4) { a1c[k4] = pre->a1c[k4]; s0.a1n[k4] = pre->a1n[k4]; }
                b1c = pre->b1c; s0.b1n = pre->b1n;
#pragma unroll
                for (int o = 0; o < OTL; ++o) s0.a2c[o] = pre->a2c[o];
            }
        }
        if (!have) {
#pragma unroll
        for (int k4 = 0; k4 < K4; ++k4) a1c[k4] = *reinterpret_cast<const f4 *>(pA1 + k4 * 256);
        b1c = *reinterpret_cast<const f4 *>(pB1);
#pragma unroll
        for (int o = 0; o < OTL; ++o) s0.a2c[o] = *reinterpret_cast<const f4 *>(pA2 + o * 256);
        const int t1 = last < 1 ? last : 1;
#pragma unroll
        for (int k4 = 0; k4 < K4; ++k4) s0.a1n[k4] = *opaque(pA1 + ((size_t)t1 * K4 + k4) * 256);
        s0.b1n = *opaque(pB1 + t1 * 16);
        }
#pragma unroll
        for (int rt = 0; rt < R; ++rt) s0.acc[rt] = gemm1_any<NF, CQ, PC, R, BX>(a1c, b1c, xr, cr, bin, rt);
    }
    auto gemm2 = [&](const St &c, const f4 (&hv)[R]) {
#pragma unroll
        for (int o = 0; o < OTL; ++o)                  // independent accumulators back to back
#pragma unroll
            for (int rho = 0; rho < 4; ++rho)
#pragma unroll
                for (int rt = 0; rt < R; ++rt) out[rt][OT0 + o] = mfma16(c.a2c[o][rho], hv[rt][rho], out[rt][OT0 + o]);
    };
    auto step = [&](const St &c, St &nx, int t) {
        // fragments for the NEXT iteration: GEMM1 of tile t+2, GEMM2 of tile t+1
        const int t2 = (t + 2 < last) ? t + 2 : last;
#pragma unroll
        for (int k4 = 0; k4 < K4; ++k4) nx.a1n[k4] = *opaque(pA1 + ((size_t)t2 * K4 + k4) * 256);
        nx.b1n = *opaque(pB1 + t2 * 16);
#pragma unroll
        for (int o = 0; o < OTL; ++o) nx.a2c[o] = *opaque(pA2 + ((size_t)(t + 1) * OTL + o) * 256);
        __builtin_amdgcn_sched_barrier(0);
        f4 hv[R];
        constexpr int RB = (R % 2 == 0) ? 2 : 1;       // row tiles per phase-A block (two chains interleave)
#pragma unroll
        for (int r0 = 0; r0 < R; r0 += RB) {           // phase A: GEMM1 of tile t+1 || activation of tile t
#pragma unroll
            for (int u = 0; u < RB; ++u) nx.acc[r0 + u] = c.b1n;
            if constexpr (BX) {
#pragma unroll
                for (int i = 0; i < K4; ++i)
#pragma unroll
                    for (int u = 0; u < RB; ++u) nx.acc[r0 + u] = mfma32(c.a1n[i], bin[r0 + u][i], nx.acc[r0 + u]);
            } else {
#pragma unroll
                for (int kk = 0; kk < NF + CQ; ++kk)
#pragma unroll
                    for (int u = 0; u < RB; ++u)
                        nx.acc[r0 + u] = mfma16(c.a1n[kk >> 2][kk & 3], in_op<NF, CQ, PC, R>(xr, cr, r0 + u, kk), nx.acc[r0 + u]);
            }
#pragma unroll
            for (int u = 0; u < RB; ++u) hv[r0 + u] = act4<ACT>(c.acc[r0 + u]);
            __builtin_amdgcn_sched_barrier(0);
        }
        gemm2(c, hv);                                  // phase B
        __builtin_amdgcn_sched_barrier(0);
    };
    auto finish = [&](const St &c) {                   // the last tile has no successor to overlap with
        f4 hv[R];
#pragma unroll
        for (int rt = 0; rt < R; ++rt) hv[rt] = act4<ACT>(c.acc[rt]);
        gemm2(c, hv);
    };
    int t = 0;
    for (; t + 1 < last; t += 2) { step(s0, s1, t); step(s1, s0, t + 1); }
    if (t < last) { step(s0, s1, t); finish(s1); } else finish(s0);
}

// d == 16 (NF == 2): the t and s nets would share ONE 16-row out tile in which every hidden tile
// multiplies 8 rows of structural zeros.  Here GEMM2 runs instead as 16 independent 4x4x1 blocks per
// instruction: block (q, rg = r >> 2) multiplies the lane group's own hidden unit 16t+4q+rho (B = the
// GEMM1 accumulator register, as before) by 4 outputs (A, pre-packed per lane) for rows 4rg..4rg+3.
// Partial sums over the lane groups q are combined once per layer by a two-step reduce-scatter.
// Measured on MI355X: 8 x 4x4x1 take 37 ns against 58 ns for the 4 x 16x16x4 they replace.
template <int CQ, int R, int PC, int NET, int ACT, bool BX = false>
__device__ __forceinline__ void run_tiles_x4(const float *__restrict__ W, const Geo &g, int lane, int tile0,
                                             int ntiles, const float (&xr)[R][4],
                                             const float (&cr)[R][CQ > 0 ? CQ : 1], f4 (&outx)[R][4],
                                             const f4 (*bin)[SplitDims<2, CQ>::NI1] = nullptr,
                                             const TilePre<2, CQ> *pre = nullptr) {
    constexpr int NF = 2;
    constexpr int K4 = G1Dims<NF, CQ, BX>::NA;
    const int q = lane >> 4;
    const float *pA1 = W + (BX ? g.oA1S : g.oA1) + ((size_t)tile0 * K4 * 64 + lane) * 4;
    const float *pB1 = W + g.oB1 + ((size_t)tile0 * 4 + q) * 4;
    const float *pA2 = W + g.oA2X + ((size_t)tile0 * 2 * 64 + lane) * 4;
    const int last = ntiles - 1;
    struct St { f4 a1n[K4], b1n, a2c[2], acc[R]; };        // two alternating pipeline states: see run_tiles
    St s0, s1;
    {
        f4 a1c[K4], b1c;
        bool have = false;
        if constexpr (!BX) {
            if (pre) {          // loaded ahead by the caller (TilePre)
                have = true;
#pragma unroll
                for (int k4 = 0; k4 < K4; ++k4) { a1c[k4] = pre->a1c[k4]; s0.a1n[k4] = pre->a1n[k4]; }
                b1c = pre->b1c; s0.b1n = pre->b1n;
                s0.a2c[0] = pre->a2c[0]; s0.a2c[1] = pre->a2c[1];
            }
        }
        if (!have) {
#pragma unroll
        for (int k4 = 0; k4 < K4; ++k4) a1c[k4] = *reinterpret_cast<const f4 *>(pA1 + k4 * 256);
        b1c = *reinterpret_cast<const f4 *>(pB1);
#pragma unroll
        for (int o = 0; o < 2; ++o) s0.a2c[o] = *reinterpret_cast<const f4 *>(pA2 + o * 256);
        const int t1 = last < 1 ? last : 1;
#pragma unroll
        for (int k4 = 0; k4 < K4; ++k4) s0.a1n[k4] = *opaque(pA1 + ((size_t)t1 * K4 + k4) * 256);
        s0.b1n = *opaque(pB1 + t1 * 16);
        }
#pragma unroll
        for (int rt = 0; rt < R; ++rt) s0.acc[rt] = gemm1_any<NF, CQ, PC, R, BX>(a1c, b1c, xr, cr, bin, rt);
    }
    auto gemm2 = [&](const St &c, const f4 (&hv)[R]) {     // 4x4x1 GEMM2, 2R independent chains
#pragma unroll
        for (int rho = 0; rho < 4; ++rho)
#pragma unroll
            for (int o = 0; o < 2; ++o)
#pragma unroll
                for (int rt = 0; rt < R; ++rt)
                    outx[rt][2 * NET + o] = mfma4(c.a2c[o][rho], hv[rt][rho], outx[rt][2 * NET + o]);
    };
    auto step = [&](const St &c, St &nx, int t) {
        const int t2 = (t + 2 < last) ? t + 2 : last;
#pragma unroll
        for (int k4 = 0; k4 < K4; ++k4) nx.a1n[k4] = *opaque(pA1 + ((size_t)t2 * K4 + k4) * 256);
        nx.b1n = *opaque(pB1 + t2 * 16);
#pragma unroll
        for (int o = 0; o < 2; ++o) nx.a2c[o] = *opaque(pA2 + ((size_t)(t + 1) * 2 + o) * 256);
        __builtin_amdgcn_sched_barrier(0);
        f4 hv[R];
        constexpr int RB = (R % 2 == 0) ? 2 : 1;
#pragma unroll
        for (int r0 = 0; r0 < R; r0 += RB) {           // phase A: GEMM1 of tile t+1 || activation of tile t
#pragma unroll
            for (int u = 0; u < RB; ++u) nx.acc[r0 + u] = c.b1n;
            if constexpr (BX) {
#pragma unroll
                for (int i = 0; i < K4; ++i)
#pragma unroll
                    for (int u = 0; u < RB; ++u) nx.acc[r0 + u] = mfma32(c.a1n[i], bin[r0 + u][i], nx.acc[r0 + u]);
            } else {
#pragma unroll
                for (int kk = 0; kk < NF + CQ; ++kk)
#pragma unroll
                    for (int u = 0; u < RB; ++u)
                        nx.acc[r0 + u] = mfma16(c.a1n[kk >> 2][kk & 3], in_op<NF, CQ, PC, R>(xr, cr, r0 + u, kk), nx.acc[r0 + u]);
            }
#pragma unroll
            for (int u = 0; u < RB; ++u) hv[r0 + u] = act4<ACT>(c.acc[r0 + u]);
            __builtin_amdgcn_sched_barrier(0);
        }
        gemm2(c, hv);                                  // phase B
        __builtin_amdgcn_sched_barrier(0);
    };
    auto finish = [&](const St &c) {
        f4 hv[R];
#pragma unroll
        for (int rt = 0; rt < R; ++rt) hv[rt] = act4<ACT>(c.acc[rt]);
        gemm2(c, hv);
    };
    int t = 0;
    for (; t + 1 < last; t += 2) { step(s0, s1, t); step(s1, s0, t + 1); }
    if (t < last) { step(s0, s1, t); finish(s1); } else finish(s0);
}

// partial sums over lane groups -> each lane keeps (net, slot f) of the features it owns:
// out (og, i) belongs to lane group 2*og + (i >> 1), slot i & 1
__device__ __forceinline__ void reduce_scatter_x4(const f4 (&part)[4], float (&tv)[2], float (&sv)[2]) {
#pragma unroll
    for (int net = 0; net < 2; ++net) {
        float s4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) s4[i] = swap_add32(part[2 * net][i], part[2 * net + 1][i]);
        const float f0 = swap_add16(s4[0], s4[2]), f1 = swap_add16(s4[1], s4[3]);
        if (net == 0) { tv[0] = f0; tv[1] = f1; } else { sv[0] = f0; sv[1] = f1; }
    }
}

// MODE 0: forward (x*exp(s)+t, log-det)   realnvp.py:99-100
// MODE 1: inverse ((x-t)*exp(-s))          realnvp.py:128
// MODE 2: forward, also writing the layer input of the transformed features and exp(s) to scr
template <int NF, int CQ, int R, int PC, int MODE, int ACT, bool BX = false>
__device__ __forceinline__ void layer_forward(const float *__restrict__ W, const Geo &g, int lane,
                                              float (&xr)[R][2 * NF], const float (&cr)[R][CQ > 0 ? CQ : 1],
                                              float (&ld)[R], float *__restrict__ scr) {
    using D = FwdDims<NF, CQ>;
    constexpr int OTL = D::OTL, NT2 = D::NT2;
    const int q = lane >> 4;
    f4 bin[BX ? R : 1][SplitDims<NF, CQ>::NI1];
    if constexpr (BX) build_bin<NF, CQ, PC, R>(xr, cr, bin);
    f4 out[R][NT2];
    const f4 bias2 = *reinterpret_cast<const f4 *>(W + g.oB2 + q * 4);      // out tile 0 (all of it when NF == 2)
    if constexpr (NF == 2 && kUseX4) {
        f4 outx[R][4];
#pragma unroll
        for (int rt = 0; rt < R; ++rt)
#pragma unroll
            for (int u = 0; u < 4; ++u) outx[rt][u] = f4{0.f, 0.f, 0.f, 0.f};
        run_tiles_x4<CQ, R, PC, 0, ACT, BX>(W, g, lane, 0, g.HT, xr, cr, outx, bin);
        run_tiles_x4<CQ, R, PC, 1, ACT, BX>(W, g, lane, g.HT, g.HT, xr, cr, outx, bin);
#pragma unroll
        for (int rt = 0; rt < R; ++rt) {
            float tv[2], sv[2];
            reduce_scatter_x4(outx[rt], tv, sv);
            out[rt][0] = f4{tv[0], tv[1], sv[0], sv[1]} + bias2;              // same slots as the shared tile
        }
    } else {
#pragma unroll
        for (int ot = 0; ot < NT2; ++ot) {
            const f4 b = (ot == 0) ? bias2 : *reinterpret_cast<const f4 *>(W + g.oB2 + (ot * 4 + q) * 4);
#pragma unroll
            for (int rt = 0; rt < R; ++rt) out[rt][ot] = b;
        }
        if (NF >= 4) {      // each net feeds its own out tiles
            run_tiles<NF, CQ, R, PC, 0, ACT, BX>(W, g, lane, 0, g.HT, xr, cr, out, bin);
            run_tiles<NF, CQ, R, PC, (NF >= 4 ? OTL : 0), ACT, BX>(W, g, lane, g.HT, g.HT, xr, cr, out, bin);
        } else {            // t and s share one out tile: one pipelined pass over all 2*HT tiles
            run_tiles<NF, CQ, R, PC, 0, ACT, BX>(W, g, lane, 0, 2 * g.HT, xr, cr, out, bin);
        }
    }
#pragma unroll
    for (int rt = 0; rt < R; ++rt) {
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            float tv, sv;
            if (NF >= 4) { tv = out[rt][f >> 2][f & 3]; sv = out[rt][(NF >= 4 ? OTL : 0) + (f >> 2)][f & 3]; }
            else { tv = out[rt][0][f & 1]; sv = out[rt][0][2 + (f & 1)]; }
            const int e = 2 * f + 1 - PC;
            if (MODE == 1) {
                xr[rt][e] = (xr[rt][e] - tv) * expf(-sv);
            } else {
                const float es = expf(sv), xv = xr[rt][e];
                if (MODE == 2) {
                    scr[((rt * 2 * NF) + f) * 64 + lane] = xv;
                    scr[((rt * 2 * NF) + NF + f) * 64 + lane] = es;
                }
                xr[rt][e] = fmaf(xv, es, tv);
                ld[rt] += sv;
            }
        }
    }
}

// One layer in net-split mode: the workgroup has 8 waves; waves w and w + 4 hold the SAME row tiles and take one
// net each (role 0: t, role 1: s).  This wave runs the hidden tiles of its own net, reduces them to its net's
// output for the features each lane owns, and swaps it with the partner wave through xown / xother in LDS
// (one __syncthreads per layer; the caller double-buffers the records by layer parity).  MODE as layer_forward.
template <int NF, int CQ, int R, int PC, int MODE, int ACT, bool BX = false>
__device__ __forceinline__ void layer_forward_ns(const float *__restrict__ W, const Geo &g, int lane, int role,
                                                 float *xown, const float *xother, float (&xr)[R][2 * NF],
                                                 const float (&cr)[R][CQ > 0 ? CQ : 1], float (&ld)[R],
                                                 float *__restrict__ scr) {
    using D = FwdDims<NF, CQ>;
    constexpr int OTL = D::OTL, NT2 = D::NT2;
    const int q = lane >> 4;
    f4 bin[BX ? R : 1][SplitDims<NF, CQ>::NI1];
    if constexpr (BX) build_bin<NF, CQ, PC, R>(xr, cr, bin);
    float own[R][NF];
    if constexpr (NF == 2 && kUseX4) {
        const f4 bias2 = *reinterpret_cast<const f4 *>(W + g.oB2 + q * 4);
        f4 outx[R][4];
#pragma unroll
        for (int rt = 0; rt < R; ++rt)
#pragma unroll
            for (int u = 0; u < 4; ++u) outx[rt][u] = f4{0.f, 0.f, 0.f, 0.f};
        if (role == 0) {
            run_tiles_x4<CQ, R, PC, 0, ACT, BX>(W, g, lane, 0, g.HT, xr, cr, outx, bin);
#pragma unroll
            for (int rt = 0; rt < R; ++rt) {
                float s4[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) s4[i] = swap_add32(outx[rt][0][i], outx[rt][1][i]);
                own[rt][0] = swap_add16(s4[0], s4[2]) + bias2[0];
                own[rt][1] = swap_add16(s4[1], s4[3]) + bias2[1];
            }
        } else {
            run_tiles_x4<CQ, R, PC, 1, ACT, BX>(W, g, lane, g.HT, g.HT, xr, cr, outx, bin);
#pragma unroll
            for (int rt = 0; rt < R; ++rt) {
                float s4[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) s4[i] = swap_add32(outx[rt][2][i], outx[rt][3][i]);
                own[rt][0] = swap_add16(s4[0], s4[2]) + bias2[2];
                own[rt][1] = swap_add16(s4[1], s4[3]) + bias2[3];
            }
        }
    } else {
        f4 out[R][NT2];
#pragma unroll
        for (int ot = 0; ot < NT2; ++ot) {
            const f4 b = *reinterpret_cast<const f4 *>(W + g.oB2 + (ot * 4 + q) * 4);
#pragma unroll
            for (int rt = 0; rt < R; ++rt) out[rt][ot] = b;
        }
        if (role == 0) {
            run_tiles<NF, CQ, R, PC, 0, ACT, BX>(W, g, lane, 0, g.HT, xr, cr, out, bin);
#pragma unroll
            for (int rt = 0; rt < R; ++rt)
#pragma unroll
                for (int f = 0; f < NF; ++f) own[rt][f] = out[rt][f >> 2][f & 3];
        } else {
            run_tiles<NF, CQ, R, PC, (NF >= 4 ? OTL : 0), ACT, BX>(W, g, lane, g.HT, g.HT, xr, cr, out, bin);
#pragma unroll
            for (int rt = 0; rt < R; ++rt)
#pragma unroll
                for (int f = 0; f < NF; ++f) own[rt][f] = out[rt][(NF >= 4 ? OTL : 0) + (f >> 2)][(f & 3) + (NF >= 4 ? 0 : 2)];
        }
    }
#pragma unroll
    for (int rt = 0; rt < R; ++rt)
#pragma unroll
        for (int f = 0; f < NF; ++f) xown[(rt * NF + f) * 64 + lane] = own[rt][f];
    __syncthreads();
#pragma unroll
    for (int rt = 0; rt < R; ++rt) {
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const float oth = xother[(rt * NF + f) * 64 + lane];
            const float tv = role == 0 ? own[rt][f] : oth, sv = role == 0 ? oth : own[rt][f];
            const int e = 2 * f + 1 - PC;
            if (MODE == 1) {
                xr[rt][e] = (xr[rt][e] - tv) * expf(-sv);
            } else {
                const float es = expf(sv), xv = xr[rt][e];
                if (MODE == 2) {    // the pair shares one scratch record: the t wave saves the layer input, the s wave exp(s)
                    if (role == 0) scr[((rt * 2 * NF) + f) * 64 + lane] = xv;
                    else scr[((rt * 2 * NF) + NF + f) * 64 + lane] = es;
                }
                xr[rt][e] = fmaf(xv, es, tv);
                ld[rt] += sv;
            }
        }
    }
}

// One layer in tile-split mode (small and medium batches: k_mfma_train_ts): all kTsWaves waves of a workgroup
// hold the SAME row tiles; wave w takes net w >> 2 and the hidden tiles [tile0, tile0 + nt) of that net (a quarter of them each), so
// a layer costs HT / 4 tile steps instead of HT -- these launches are pure latency chains.  The waves' partial t / s
// outputs meet in LDS (`red`: one record of R * 2 * 64 floats per wave, double buffered by layer parity by the caller,
// one __syncthreads per layer) and every wave adds the eight records in wave order.  MODE as layer_forward; MODE 2: wave 0
// also writes the layer input of the transformed features and exp(s) to scr (read by all waves in the backward).
constexpr int kTsWaves = 8, kTsSlices = 4;
template <int NF, int CQ, int R, int PC, int MODE, int ACT>
__device__ __forceinline__ void layer_forward_ts(const float *__restrict__ W, const Geo &g, int lane, int wave, int tile0,
                                                 int nt, float *red, float (&xr)[R][2 * NF],
                                                 const float (&cr)[R][CQ > 0 ? CQ : 1], float (&ld)[R],
                                                 float *__restrict__ scr, const float *__restrict__ Wnext,
                                                 TilePre<NF, CQ> &pre, bool use_pre) {
    // pre: this layer's opening fragments, loaded by the caller / the previous layer; Wnext (nullable): the layer that
    // follows in this pass -- its opening fragments are requested here, before the rendezvous
    using D = FwdDims<NF, CQ>;
    constexpr int OTL = D::OTL, NT2 = D::NT2;
    const int q = lane >> 4, net = wave >> 2;
    // bias of the second Linear: the lane's slot of the t tile(s) and of the s tile(s), as layer_forward reads it (requested
    // now, needed after the rendezvous)
    float b2t[NF], b2s[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        if (NF >= 4) {
            b2t[f] = W[g.oB2 + ((f >> 2) * 4 + q) * 4 + (f & 3)];
            b2s[f] = W[g.oB2 + ((OTL + (f >> 2)) * 4 + q) * 4 + (f & 3)];
        } else {
            b2t[f] = W[g.oB2 + q * 4 + (f & 1)];
            b2s[f] = W[g.oB2 + q * 4 + 2 + (f & 1)];
        }
    }
    float own[R][NF];           // this wave's share of its net's output for the features each lane owns (no bias)
    if constexpr (NF == 2 && kUseX4) {
        f4 outx[R][4];
#pragma unroll
        for (int rt = 0; rt < R; ++rt)
#pragma unroll
            for (int u = 0; u < 4; ++u) outx[rt][u] = f4{0.f, 0.f, 0.f, 0.f};
        if (nt > 0) {
            if (net == 0) run_tiles_x4<CQ, R, PC, 0, ACT>(W, g, lane, tile0, nt, xr, cr, outx, nullptr, use_pre ? &pre : nullptr);
            else run_tiles_x4<CQ, R, PC, 1, ACT>(W, g, lane, g.HT + tile0, nt, xr, cr, outx, nullptr, use_pre ? &pre : nullptr);
            if (use_pre && Wnext) load_tile_pre<NF, CQ>(Wnext, g, lane, net * g.HT + tile0, nt, pre);
        }
#pragma unroll
        for (int rt = 0; rt < R; ++rt) {
            float s4[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                s4[i] = net == 0 ? swap_add32(outx[rt][0][i], outx[rt][1][i]) : swap_add32(outx[rt][2][i], outx[rt][3][i]);
            own[rt][0] = swap_add16(s4[0], s4[2]);
            own[rt][1] = swap_add16(s4[1], s4[3]);
        }
    } else {
        f4 out[R][NT2];
#pragma unroll
        for (int ot = 0; ot < NT2; ++ot)
#pragma unroll
            for (int rt = 0; rt < R; ++rt) out[rt][ot] = f4{0.f, 0.f, 0.f, 0.f};
        if (nt > 0) {
            if (net == 0) run_tiles<NF, CQ, R, PC, 0, ACT>(W, g, lane, tile0, nt, xr, cr, out, nullptr, use_pre ? &pre : nullptr);
            else run_tiles<NF, CQ, R, PC, (NF >= 4 ? OTL : 0), ACT>(W, g, lane, g.HT + tile0, nt, xr, cr, out, nullptr, use_pre ? &pre : nullptr);
            if (use_pre && Wnext) load_tile_pre<NF, CQ>(Wnext, g, lane, net * g.HT + tile0, nt, pre);
        }
#pragma unroll
        for (int rt = 0; rt < R; ++rt)
#pragma unroll
            for (int f = 0; f < NF; ++f)
                own[rt][f] = net == 0 ? out[rt][f >> 2][f & 3]
                                      : out[rt][(NF >= 4 ? OTL : 0) + (f >> 2)][(f & 3) + (NF >= 4 ? 0 : 2)];
    }
#pragma unroll
    for (int rt = 0; rt < R; ++rt)
#pragma unroll
        for (int f = 0; f < NF; ++f) red[(wave * R * NF + rt * NF + f) * 64 + lane] = own[rt][f];
    __syncthreads();
#pragma unroll
    for (int rt = 0; rt < R; ++rt) {
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            float tv = b2t[f], sv = b2s[f];
#pragma unroll
            for (int w = 0; w < kTsSlices; ++w) {
                tv += red[(w * R * NF + rt * NF + f) * 64 + lane];
                sv += red[((kTsSlices + w) * R * NF + rt * NF + f) * 64 + lane];
            }
            const int e = 2 * f + 1 - PC;
            if (MODE == 1) {
                xr[rt][e] = (xr[rt][e] - tv) * expf(-sv);
            } else {
                const float es = expf(sv), xv = xr[rt][e];
                if (MODE == 2 && wave == 0) {
                    scr[((rt * 2 * NF) + f) * 64 + lane] = xv;
                    scr[((rt * 2 * NF) + NF + f) * 64 + lane] = es;
                }
                xr[rt][e] = fmaf(xv, es, tv);
                ld[rt] += sv;
            }
        }
    }
}

}  // namespace mfma
}  // namespace rnvp
