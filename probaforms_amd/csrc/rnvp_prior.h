// rnvp_prior.h -- counter-based standard-normal draws for the prior of the flow
// (/root/reference/probaforms/models/nflow.py:141 `X = self.prior.sample((n,))`; the reference's prior is
// MultivariateNormal(0, I), realnvp.py:189-191, i.e. independent N(0,1) per feature).
//
// z[row][j] is a pure function of (seed, GLOBAL row index, feature index j): Philox4x32-10 keyed by the seed,
// counter (row_lo, row_hi, j / 4, 0), whose four 32-bit outputs become the four normals of features
// 4*(j/4) .. 4*(j/4)+3 by two Box-Muller pairs.  Any sharding of the rows over workgroups, chunks or ranks
// therefore reproduces the single-call draw bit for bit, and the draw can be made inside the inverse kernel
// (z never touches HBM).  Not the reference's CPU stream: that is the host prior (`prior_rng='host'`).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rnvp {

__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
        const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
        c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

// Box-Muller on two 32-bit words: u1 in [2^-24, 1 - 2^-24] (23 random bits + half a step), angle fraction
// in [0, 1) with 24 bits.  Accurate logf / sincospif: the CPU restatement (oracle/) agrees to ~1e-7.
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float &z0, float &z1) {
    const float u1 = (float)(a >> 9) * 1.1920928955078125e-07f + 5.9604644775390625e-08f;   // *2^-23 + 2^-24
    const float f = (float)(b >> 8) * 5.9604644775390625e-08f;                               // *2^-24
    const float rad = sqrtf(-2.0f * logf(u1));
    float s, c;
    sincospif(2.0f * f, &s, &c);
    z0 = rad * c;
    z1 = rad * s;
}

// the four normals of features 4*blk .. 4*blk+3 of global row `row`
__device__ __forceinline__ void prior_normal4(uint64_t seed, int64_t row, int blk, float (&z)[4]) {
    uint32_t c[4] = {(uint32_t)(uint64_t)row, (uint32_t)((uint64_t)row >> 32), (uint32_t)blk, 0u};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    box_muller(c[0], c[1], z[0], z[1]);
    box_muller(c[2], c[3], z[2], z[3]);
}

}  // namespace rnvp
