/*
 * prior_torch_oracle.c -- TEST INFRASTRUCTURE ONLY (imported by tests/ alone; nothing under probaforms_amd/ touches it).
 *
 * CPU restatement of `torch.randn(count)` on a CPU generator, the reference's prior draw
 * (/root/reference/probaforms/models/nflow.py:141 -> MultivariateNormal(0, I).sample -> randn), as the build's
 * rnvp_prior_normal_torch_cpu (probaforms_amd/csrc/rnvp_prior_torch.hip) restates it for the device:
 *   - the mt19937 engine of ATen/core/MT19937RNGEngine.h (624-word state, tempering), 24-bit uniforms
 *     (ATen/core/DistributionsHelper.h uniform_real_distribution<float>);
 *   - ATen/native/cpu/DistributionTemplates.h normal_fill_AVX2 / normal_fill_16_AVX2 -- the path torch's CPU kernel takes for a
 *     contiguous float tensor of >= 16 elements on AVX2 and AVX-512 hosts alike: 24-bit uniforms into the tensor, then per block
 *     of 16  u1 = 1 - data[j], u2 = data[j + 8], radius = sqrt(-2 log256_ps(u1)), theta = float(2 pi) u2,
 *     data[j] = radius cos, data[j + 8] = radius sin; a count that is not a multiple of 16 redraws the last 16 elements from 16
 *     fresh uniforms;
 *   - log256_ps / sincos256_ps of ATen/native/cpu/avx_mathfun.h (the cephes single-precision polynomials), one lane at a time,
 *     with the multiply-adds GCC contracts under -mfma (the FIRST multiplication feeding an addition is fused) written as
 *     fmaf(), so that the SAME float32 arithmetic can run on the GPU.
 * The reference's arithmetic lives in PyTorch: this restatement is pinned against torch.randn itself by
 * tests/test_prior_torch.py (4M-number streams and the block / tail edge cases, bit for bit).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

static inline uint32_t asuint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float asfloat(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static float log_ps(float x) {           /* x in [2^-24, 1] */
    uint32_t xi = asuint(x);
    int32_t imm0 = (int32_t)(xi >> 23);
    xi = (xi & ~0x7f800000u) | 0x3f000000u;
    x = asfloat(xi);
    imm0 -= 0x7f;
    float e = (float)imm0 + 1.0f;
    const int mask = x < 0.707106781186547524f;
    const float tmp = mask ? x : 0.0f;
    x = x - 1.0f;
    e = e - (mask ? 1.0f : 0.0f);
    x = x + tmp;
    const float z = x * x;
    float y = 7.0376836292E-2f;
    y = fmaf(y, x, -1.1514610310E-1f); y = fmaf(y, x, 1.1676998740E-1f); y = fmaf(y, x, -1.2420140846E-1f);
    y = fmaf(y, x, 1.4249322787E-1f); y = fmaf(y, x, -1.6668057665E-1f); y = fmaf(y, x, 2.0000714765E-1f);
    y = fmaf(y, x, -2.4999993993E-1f); y = fmaf(y, x, 3.3333331174E-1f);
    y = y * x;
    y = fmaf(y, z, e * -2.12194440e-4f);
    y = fmaf(-z, 0.5f, y);
    x = x + y;
    return fmaf(e, 0.693359375f, x);
}

static void sincos_ps(float x, float *s, float *c) {        /* x >= 0 */
    float y = x * 1.27323954473516f;
    int32_t imm2 = (int32_t)y;
    imm2 = (imm2 + 1) & ~1;
    y = (float)imm2;
    const uint32_t sign_bit_sin = ((uint32_t)(imm2 & 4)) << 29;
    const int poly_mask = (imm2 & 2) == 0;
    x = fmaf(y, -0.78515625f, x); x = fmaf(y, -2.4187564849853515625e-4f, x); x = fmaf(y, -3.77489497744594108e-8f, x);
    const uint32_t sign_bit_cos = ((uint32_t)(~(imm2 - 2) & 4)) << 29;
    const float z = x * x;
    float yc = 2.443315711809948E-005f;
    yc = fmaf(yc, z, -1.388731625493765E-003f); yc = fmaf(yc, z, 4.166664568298827E-002f);
    yc = yc * z;
    yc = fmaf(yc, z, -(z * 0.5f));
    yc = yc + 1.0f;
    float ys = -1.9515295891E-4f;
    ys = fmaf(ys, z, 8.3321608736E-3f); ys = fmaf(ys, z, -1.6666654611E-1f);
    ys = ys * z;
    ys = fmaf(ys, x, x);
    const float xs = poly_mask ? ys : yc, xc = poly_mask ? yc : ys;
    *s = asfloat(asuint(xs) ^ sign_bit_sin);
    *c = asfloat(asuint(xc) ^ sign_bit_cos);
}

/* ---- mt19937 (ATen/core/MT19937RNGEngine.h) ---- */
static void mt_next_state(uint32_t *s) {
    uint32_t n[624];
    for (int k = 0; k < 624; ++k) {
        const uint32_t u = s[k], v = k + 1 < 624 ? s[k + 1] : n[0];
        const uint32_t far = k + 397 < 624 ? s[k + 397] : n[k + 397 - 624];
        const uint32_t y = (u & 0x80000000u) | (v & 0x7fffffffu);
        n[k] = far ^ (y >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u);
    }
    memcpy(s, n, sizeof(n));
}
static uint32_t mt_draw(uint32_t *mt) {          /* mt[624] = position of the next unread word */
    if (mt[624] >= 624) { mt_next_state(mt); mt[624] = 0; }
    uint32_t y = mt[mt[624]++];
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
    return y;
}
static void fill16(float *d) {                  /* normal_fill_16_AVX2, mean 0, std 1 */
    for (int j = 0; j < 8; ++j) {
        const float u1 = 1.0f - d[j], u2 = d[j + 8];
        const float radius = sqrtf(-2.0f * log_ps(u1));
        const float theta = 6.283185307179586f * u2;
        float sn, cs;
        sincos_ps(theta, &sn, &cs);
        d[j] = fmaf(radius * cs, 1.0f, 0.0f);           /* _mm256_fmadd_ps(radius cos, std, mean): a -0 becomes +0 */
        d[j + 8] = fmaf(radius * sn, 1.0f, 0.0f);
    }
}
/* torch.randn(count), count >= 16, contiguous float */
int prior_torch_randn(uint32_t *mt625, int64_t count, float *out) {
    if (count < 16) return -1;
    for (int64_t i = 0; i < count; ++i) out[i] = (float)(mt_draw(mt625) & 0xffffffu) * 0x1p-24f;
    for (int64_t i = 0; i + 16 <= count; i += 16) fill16(out + i);
    if (count % 16) {
        float *d = out + count - 16;
        for (int i = 0; i < 16; ++i) d[i] = (float)(mt_draw(mt625) & 0xffffffu) * 0x1p-24f;
        fill16(d);
    }
    return 0;
}
