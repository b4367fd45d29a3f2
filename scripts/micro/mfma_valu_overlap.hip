// Microbenchmark: do v_mfma_f32_16x16x4_f32 and VALU work overlap on one SIMD (gfx950)?
// Build: hipcc -O3 --offload-arch=gfx950 mfma_valu_overlap.hip -o mfma_valu_overlap ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
using f4 = __attribute__((ext_vector_type(4))) float;

template <int MODE, int WAVES>   // MODE 1: MFMA only, 2: VALU (exp+rcp) only, 3: both interleaved, 4: VALU fma only, 5: MFMA + fma
__global__ void __launch_bounds__(WAVES * 64) k(float *out, int iters) {
    f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MODE == 1 || MODE == 3 || MODE == 5) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u], 0, 0, 0);
            if (MODE == 2 || MODE == 3) {
                v[2 * u] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v[2 * u]));
                v[2 * u + 1] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v[2 * u + 1]));
            }
            if (MODE == 4 || MODE == 5) {
#pragma unroll
                for (int w = 0; w < 8; ++w) v[w] = fmaf(v[w], 1.0001f, 0.5f);
            }
        }
    }
    float s = 0;
    for (int u = 0; u < 4; ++u) s += acc[u][0] + acc[u][1] + acc[u][2] + acc[u][3];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int WAVES> float run(float *d, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, WAVES>), dim3(256), dim3(WAVES * 64), 0, 0, d, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, WAVES>), dim3(256), dim3(WAVES * 64), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
    float *d; hipMalloc(&d, 256 * 512 * 4);
    const int it = 20000;
    printf("per iteration: 4 MFMA 16x16x4 f32 (=128 MFMA cycles); mode2/3: 8 exp + 8 rcp + 8 add; mode4/5: 32 v_fma\n");
    printf("1 wave/SIMD : mfma %.3f ms  trans %.3f ms  mfma+trans %.3f ms | fma %.3f ms  mfma+fma %.3f ms\n",
           run<1, 4>(d, it), run<2, 4>(d, it), run<3, 4>(d, it), run<4, 4>(d, it), run<5, 4>(d, it));
    printf("2 waves/SIMD: mfma %.3f ms  trans %.3f ms  mfma+trans %.3f ms | fma %.3f ms  mfma+fma %.3f ms\n",
           run<1, 8>(d, it), run<2, 8>(d, it), run<3, 8>(d, it), run<4, 8>(d, it), run<5, 8>(d, it));
    return 0;
}
