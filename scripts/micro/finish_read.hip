// finish_read.hip -- what bounds k_train_finish's partial sums (round 6): G = 256 per-workgroup partials of P floats are added
// over G by 2 * nrec workgroups, each owning a 1 KB (W1 half) or 512 B (W2 half) slice of every partial.  Compares
//   A  the shipped layout  [G][P]: a block reads G chunks of 1 KB at a stride of P * 4 bytes
//   B  the transposed layout [P / chunk][G][chunk]: a block reads one contiguous G KB region
// with the same thread-to-float4 mapping, 8 loads in flight, fixed-order sums (the bytes are first written by a kernel with
// the training kernel's store shape, so that they sit where the training kernel leaves them).
// hipcc -O3 --offload-arch=gfx950 finish_read.hip -o finish_read && ./finish_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f4 = __attribute__((ext_vector_type(4))) float;
constexpr int G = 256, CH = 256 /* floats per chunk */, NCH = 192 /* chunks per partial: 49152 floats */, T = 512;

__global__ void __launch_bounds__(256) k_write(float *p, size_t n4) {       // every workgroup writes its own partial (layout A) / its chunks (B)
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256)
        reinterpret_cast<f4 *>(p)[i] = f4{1.f, 2.f, 3.f, 4.f};
}
template <int INFL>
__global__ void __launch_bounds__(T) k_sum(const float *__restrict__ p, size_t stride4 /* f4 between partials */, size_t chunk4 /* f4 between chunks */, float *out) {
    __shared__ f4 red[T];
    const int t = threadIdx.x, nf4 = CH / 4, nsub = T / nf4, col = t % nf4, sub = t / nf4;
    const f4 *src = reinterpret_cast<const f4 *>(p) + (size_t)blockIdx.x * chunk4 + col;
    f4 acc = f4{0.f, 0.f, 0.f, 0.f};
    int b = sub;
    for (; b + (INFL - 1) * nsub < G; b += INFL * nsub) {
        f4 v[INFL];
#pragma unroll
        for (int u = 0; u < INFL; ++u) v[u] = __builtin_nontemporal_load(src + (size_t)(b + u * nsub) * stride4);
#pragma unroll
        for (int u = 0; u < INFL; ++u) acc += v[u];
    }
    for (; b < G; b += nsub) acc += __builtin_nontemporal_load(src + (size_t)b * stride4);
    red[t] = acc;
    __syncthreads();
    if (t < nf4) {
        f4 a = red[t];
        for (int s2 = 1; s2 < nsub; ++s2) a += red[s2 * nf4 + t];
        reinterpret_cast<f4 *>(out)[(size_t)blockIdx.x * nf4 + t] = a;
    }
}
int main() {
    const size_t P = (size_t)NCH * CH, n = (size_t)G * P;
    float *a, *out;
    hipMalloc(&a, n * 4); hipMalloc(&out, P * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *name, size_t stride4, size_t chunk4, int infl, bool rewrite) {
        float best = 1e9f, sum = 0.f; const int reps = 20;
        for (int r = 0; r < reps; ++r) {
            if (rewrite) hipLaunchKernelGGL(k_write, dim3(256), dim3(256), 0, 0, a, n / 4);
            hipEventRecord(e0);
            if (infl == 8) hipLaunchKernelGGL(k_sum<8>, dim3(NCH), dim3(T), 0, 0, a, stride4, chunk4, out);
            else hipLaunchKernelGGL(k_sum<4>, dim3(NCH), dim3(T), 0, 0, a, stride4, chunk4, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (r >= 2) { sum += ms; if (ms < best) best = ms; }
        }
        printf("%-64s %2d in flight %s: avg %.2f us, best %.2f us = %.2f TB/s\n", name, infl, rewrite ? "fresh writes" : "re-read      ", sum / (reps - 2) * 1e3, best * 1e3,
               n * 4 / (best * 1e-3) / 1e12);
    };
    for (int rw = 1; rw >= 0; --rw)
        for (int infl : {8, 4}) {
            run("A [G][P]: 1 KB chunks at a stride of one partial", P / 4, CH / 4, infl, rw);
            run("B [chunk][G][256]: one contiguous 256 KB region per block", CH / 4, (size_t)G * CH / 4, infl, rw);
        }
    return 0;
}
