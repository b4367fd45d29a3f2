// Microbenchmark (gfx950): issue rates of the bf16 / f16 MFMA forms next to the f32 one, and whether they
// overlap with VALU work (transcendentals; the and/sub/perm chain of a 3-term bf16 split) on one SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 bf16_overlap.hip -o bf16_overlap ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
using f4 = __attribute__((ext_vector_type(4))) float;
using s4 = __attribute__((ext_vector_type(4))) short;
using bf8 = __attribute__((ext_vector_type(8))) __bf16;
using h4 = __attribute__((ext_vector_type(4))) _Float16;
using u4 = __attribute__((ext_vector_type(4))) unsigned;

// MFMA kinds: 0 none, 1 f32 16x16x4, 2 bf16 16x16x16, 3 bf16 16x16x32, 4 f16 16x16x16
// VALU kinds: 0 none, 1 8 x (exp2 + add + rcp), 2 3-term bf16 split of 8 values, 3 both
template <int MK, int NM, int VK, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) k(float *out, int iters) {
    f4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f4{0, 0, 0, 0};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    s4 sa = {(short)threadIdx.x, 1, 2, 3}, sb = {4, 5, (short)threadIdx.x, 7};
    u4 ua = {threadIdx.x, 1u, 2u, 3u}, ub = {4u, 5u, threadIdx.x, 7u};
    h4 ha = {(_Float16)a, 1, 2, 3}, hb = {(_Float16)b, 1, 2, 3};
    float v[8];
    unsigned pk[12];
    for (int i = 0; i < 12; ++i) pk[i] = 0;
    for (int i = 0; i < 8; ++i) v[i] = a + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < NM; ++u) {
            if (MK == 1) acc[u & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u & 7], 0, 0, 0);
            if (MK == 2) acc[u & 7] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(sa, sb, acc[u & 7], 0, 0, 0);
            if (MK == 3) acc[u & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, ua), __builtin_bit_cast(bf8, ub), acc[u & 7], 0, 0, 0);
            if (MK == 4) acc[u & 7] = __builtin_amdgcn_mfma_f32_16x16x16f16(ha, hb, acc[u & 7], 0, 0, 0);
            if ((VK & 1) && u < 8) v[u] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v[u]));
            if ((VK & 2) && u < 4) {      // 3-term truncation split of v[2u], v[2u+1] -> 3 packed words
                const float x0 = v[2 * u] + (VK == 2 ? 1.0f : 0.f), x1 = v[2 * u + 1];
                const float r0 = x0 - __uint_as_float(__float_as_uint(x0) & 0xffff0000u);
                const float r1 = x1 - __uint_as_float(__float_as_uint(x1) & 0xffff0000u);
                const float q0 = r0 - __uint_as_float(__float_as_uint(r0) & 0xffff0000u);
                const float q1 = r1 - __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
                pk[3 * u] ^= __builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), 0x07060302u);
                pk[3 * u + 1] ^= __builtin_amdgcn_perm(__float_as_uint(r1), __float_as_uint(r0), 0x07060302u);
                pk[3 * u + 2] ^= __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302u);
                if (VK == 2) { v[2 * u] = x0; }
            }
        }
        if (VK & 2) { ua[1] ^= pk[0]; ub[1] ^= pk[1]; }
    }
    float s = 0;
    for (int u = 0; u < 8; ++u) s += acc[u][0] + acc[u][1] + acc[u][2] + acc[u][3];
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int i = 0; i < 12; ++i) s += (float)pk[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MK, int NM, int VK, int WAVES> float run(float *d, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MK, NM, VK, WAVES>), dim3(256), dim3(WAVES * 64), 0, 0, d, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MK, NM, VK, WAVES>), dim3(256), dim3(WAVES * 64), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

template <int WAVES> void table(float *d, int it) {
    const double ns = 1e6 / it;      // ms per launch -> ns per iteration
    printf("%d wave(s)/SIMD, ns per iteration (8 MFMA per iteration unless noted):\n", WAVES / 4);
    printf("  MFMA alone   : f32 16x16x4 %.1f | bf16 16x16x16 %.1f | bf16 16x16x32 %.1f | f16 16x16x16 %.1f\n",
           run<1, 8, 0, WAVES>(d, it) * ns, run<2, 8, 0, WAVES>(d, it) * ns, run<3, 8, 0, WAVES>(d, it) * ns, run<4, 8, 0, WAVES>(d, it) * ns);
    printf("  VALU alone   : 8 x (exp2+add+rcp) %.1f | 3-term split of 8 values %.1f | both %.1f\n",
           run<0, 8, 1, WAVES>(d, it) * ns, run<0, 8, 2, WAVES>(d, it) * ns, run<0, 8, 3, WAVES>(d, it) * ns);
    printf("  8 x bf16 16x16x32 + : trans %.1f | split %.1f | both %.1f\n",
           run<3, 8, 1, WAVES>(d, it) * ns, run<3, 8, 2, WAVES>(d, it) * ns, run<3, 8, 3, WAVES>(d, it) * ns);
    printf("  8 x bf16 16x16x16 + : trans %.1f | split %.1f | both %.1f\n",
           run<2, 8, 1, WAVES>(d, it) * ns, run<2, 8, 2, WAVES>(d, it) * ns, run<2, 8, 3, WAVES>(d, it) * ns);
    printf("  16 x bf16 16x16x32 + : none %.1f | trans %.1f | both %.1f\n",
           run<3, 16, 0, WAVES>(d, it) * ns, run<3, 16, 1, WAVES>(d, it) * ns, run<3, 16, 3, WAVES>(d, it) * ns);
    printf("  8 x f32 16x16x4 +   : trans %.1f | split %.1f | both %.1f\n",
           run<1, 8, 1, WAVES>(d, it) * ns, run<1, 8, 2, WAVES>(d, it) * ns, run<1, 8, 3, WAVES>(d, it) * ns);
}

int main() {
    float *d; hipMalloc(&d, 256 * 512 * 4);
    const int it = 20000;
    table<4>(d, it);
    table<8>(d, it);
    return 0;
}
