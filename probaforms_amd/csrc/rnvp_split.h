// rnvp_split.h -- three-term bf16 split operands for v_mfma_f32_16x16x32_bf16 (gfx950), shared by the bx3 flow kernels
// (rnvp_bx3.hip) and the split-GEMM1 form of the training kernel (rnvp_mfma_train.hip).
//
// x = t1 + t2 + t3 with bf16 terms (t1, t2 by truncation, so the residuals are exact in f32).  A product of two split
// operands keeps the six terms a1b1, a1b2, a2b1, a1b3, a2b2, a3b1: every dropped one is below 2^-24 of the result, i.e.
// float32-level accuracy.  Along K, value k of a lane owns dwords 3k .. 3k+2 of the lane's slot list, two bf16 slots each
// (low half first):
//     dword 3k+0 : B = (b1, b2)   A = (a1, a1)
//     dword 3k+1 : B = (b1, b3)   A = (a2, a1)
//     dword 3k+2 : B = (b2, b1)   A = (a2, a3)
// so nval values take ceil(3 nval / 4) MFMAs of 8 slots per lane; unused slots are zero in A.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rnvp {
namespace split {

using f4 = __attribute__((ext_vector_type(4))) float;
using bf8 = __attribute__((ext_vector_type(8))) __bf16;

__device__ __forceinline__ f4 mfma32(f4 a, f4 b, f4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c, 0, 0, 0);
}

__host__ __device__ constexpr int n_mfma(int nval) { return (3 * nval + 3) / 4; }

// weights (A side; pack kernels): t3 rounded to nearest-even; returns the upper halves' bit patterns
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
// dword p (0..2) of a weight's A-side slot triple
__device__ __forceinline__ uint32_t a_dword(float w, int p) {
    uint32_t a1, a2, a3;
    split3_rne(w, a1, a2, a3);
    return p == 0 ? (a1 | (a1 << 16)) : (p == 1 ? (a2 | (a1 << 16)) : (a2 | (a3 << 16)));
}

// B side: the three dwords of one value (third term rounded to nearest-even)
__device__ __forceinline__ void b_dwords(float v, uint32_t &d0, uint32_t &d1, uint32_t &d2) {
    const uint32_t u = __float_as_uint(v);
    const float r1 = v - __uint_as_float(u & 0xffff0000u);
    const uint32_t u1 = __float_as_uint(r1);
    const float r2 = r1 - __uint_as_float(u1 & 0xffff0000u);
    uint32_t u2 = __float_as_uint(r2);
    u2 += 0x7fffu + ((u2 >> 16) & 1u);
    d0 = __builtin_amdgcn_perm(u1, u, 0x07060302u);      // (b1, b2)
    d1 = __builtin_amdgcn_perm(u2, u, 0x07060302u);      // (b1, b3)
    d2 = __builtin_amdgcn_perm(u, u1, 0x07060302u);      // (b2, b1)
}

// B operand fragments of NV values: fr[i], i < n_mfma(NV)
template <int NV>
__device__ __forceinline__ void build_b(const float (&v)[NV], f4 (&fr)[n_mfma(NV)]) {
    constexpr int NI = n_mfma(NV);
    uint32_t dw[4 * NI];
#pragma unroll
    for (int i = 0; i < 4 * NI; ++i) dw[i] = 0u;
#pragma unroll
    for (int k = 0; k < NV; ++k) b_dwords(v[k], dw[3 * k], dw[3 * k + 1], dw[3 * k + 2]);
#pragma unroll
    for (int i = 0; i < NI; ++i)
        fr[i] = f4{__uint_as_float(dw[4 * i]), __uint_as_float(dw[4 * i + 1]), __uint_as_float(dw[4 * i + 2]),
                   __uint_as_float(dw[4 * i + 3])};
}

}  // namespace split
}  // namespace rnvp
