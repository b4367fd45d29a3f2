// rnvp_adam.hip -- torch.optim.Adam over the flat parameter buffer
// (/root/reference/probaforms/models/realnvp.py:205-207,251): amsgrad off, L2 weight decay
// folded into the gradient, bias corrections computed on the host in double like torch does.
// Pure streaming: 16 B read + 12 B written per parameter, float4-vectorised.
#include <cmath>

#include "rnvp_common.h"

namespace rnvp {
namespace {

__global__ void __launch_bounds__(256)
k_adam(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m, float *__restrict__ v,
       int64_t n, AdamK a, const float *__restrict__ loss_in, float *__restrict__ loss_out) {
    // data-parallel step: the all-reduced [gradient | loss] message also carries the batch loss; read it out here
    if (loss_out && blockIdx.x == 0 && threadIdx.x == 0) loss_out[0] = loss_in[0];
    const int64_t n4 = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 pp = reinterpret_cast<float4 *>(p)[i];
        const float4 gg = reinterpret_cast<const float4 *>(g)[i];
        float4 mm = reinterpret_cast<float4 *>(m)[i];
        float4 vv = reinterpret_cast<float4 *>(v)[i];
        adam_one(pp.x, gg.x, mm.x, vv.x, a);
        adam_one(pp.y, gg.y, mm.y, vv.y, a);
        adam_one(pp.z, gg.z, mm.z, vv.z, a);
        adam_one(pp.w, gg.w, mm.w, vv.w, a);
        reinterpret_cast<float4 *>(p)[i] = pp;
        reinterpret_cast<float4 *>(m)[i] = mm;
        reinterpret_cast<float4 *>(v)[i] = vv;
    }
    // tail (n % 4)
    const int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) adam_one(p[i], g[i], m[i], v[i], a);
}

}  // namespace

AdamK make_adam(double lr, double beta1, double beta2, double eps, double wd, int64_t step) {
    AdamK a;
    const double bc1 = 1.0 - std::pow(beta1, (double)step);       // scalar bookkeeping in double, like torch
    const double bc2 = 1.0 - std::pow(beta2, (double)step);
    a.step_size = (float)(lr / bc1);
    a.bc2_sqrt = (float)std::sqrt(bc2);
    a.w1 = (float)(1.0 - beta1);
    a.beta2 = (float)beta2;
    a.w2 = (float)(1.0 - beta2);
    a.wd = (float)wd;
    a.eps = (float)eps;
    a.use_wd = wd != 0.0;
    return a;
}

int adam_step(hipStream_t st, float *p, const float *g, float *m, float *v, int64_t n,
              double lr, double beta1, double beta2, double eps, double wd, int64_t step,
              const float *loss_in, float *loss_out) {
    if ((loss_in == nullptr) != (loss_out == nullptr)) return RNVP_EINVAL;
    if (n == 0 && loss_out) return (int)hipMemcpyAsync(loss_out, loss_in, sizeof(float), hipMemcpyDeviceToDevice, st);
    if (n == 0) return RNVP_OK;
    if (!p || !g || !m || !v || n < 0 || step < 1) return RNVP_EINVAL;
    const uintptr_t al = (uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v;
    if (al & 15) return RNVP_EINVAL;                    // float4 path needs 16-B aligned buffers
    const AdamK a = make_adam(lr, beta1, beta2, eps, wd, step);
    const int threads = 256;
    int64_t blocks = ((n >> 2) + threads - 1) / threads;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_adam, dim3((unsigned)blocks), dim3(threads), 0, st, p, g, m, v, n, a, loss_in, loss_out);
    RNVP_HIP_TRY(hipGetLastError());
    return RNVP_OK;
}

}  // namespace rnvp
