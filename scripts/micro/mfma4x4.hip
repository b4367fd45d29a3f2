// Probe v_mfma_f32_4x4x1_16B_f32: operand/result lane maps and issue rate on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
using f4 = __attribute__((ext_vector_type(4))) float;

__global__ void probe(float *out) {
    const int l = threadIdx.x;
    // A_b[i] = 100*b + i  (lane 4b+i),  B_b[j] = 1000 + 10*j + 0.001*b
    const float a = 100.f * (l >> 2) + (l & 3), b = 1.f + (l & 3) * 0.01f;
    f4 d = {0, 0, 0, 0};
    d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d, 0, 0, 0);
    for (int i = 0; i < 4; ++i) out[l * 4 + i] = d[i];
}

template <int MODE>
__global__ void __launch_bounds__(512) rate(float *out, int iters) {
    f4 acc[8];
    for (int u = 0; u < 8; ++u) acc[u] = f4{0, 0, 0, 0};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) acc[u] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[u], 0, 0, 0);
            else acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u], 0, 0, 0);
        }
    }
    float s = 0;
    for (int u = 0; u < 8; ++u) s += acc[u][0] + acc[u][1] + acc[u][2] + acc[u][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    float *d; hipMalloc(&d, 1 << 20);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    // expected if D_b[i][j] sits in VGPR i of lane 4b+j: value = A_b[i]*B_b[j] = (100b+i)*(1+0.01j)
    int ok = 1;
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) {
        const int b = l >> 2, j = l & 3;
        const float e = (100.f * b + i) * (1.f + 0.01f * j);
        if (fabsf(h[l * 4 + i] - e) > 1e-3f) { ok = 0; if (l < 8) printf("lane %d reg %d got %f expected %f\n", l, i, h[l * 4 + i], e); }
    }
    printf("layout D_b[i][j] = VGPR i of lane 4b+j : %s\n", ok ? "CONFIRMED" : "NO");
    for (int threads = 256; threads <= 512; threads *= 2)
    for (int mode = 0; mode < 2; ++mode) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        if (mode == 0) hipLaunchKernelGGL(rate<0>, dim3(256), dim3(threads), 0, 0, d, 100); else hipLaunchKernelGGL(rate<1>, dim3(256), dim3(threads), 0, 0, d, 100);
        hipEventRecord(e0);
        const int it = 20000;
        if (mode == 0) hipLaunchKernelGGL(rate<0>, dim3(256), dim3(threads), 0, 0, d, it); else hipLaunchKernelGGL(rate<1>, dim3(256), dim3(threads), 0, 0, d, it);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.3f ms for %d x 8 per wave, %d wave(s)/SIMD -> %.1f ns per instruction per SIMD\n", mode == 0 ? "4x4x1_16B" : "16x16x4", ms, it, threads / 256, ms * 1e6 / (it * 8 * (threads / 256)));
    }
    return 0;
}
