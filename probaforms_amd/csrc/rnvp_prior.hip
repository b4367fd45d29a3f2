// rnvp_prior.hip -- the prior draw of NormalizingFlow.sample (/root/reference/probaforms/models/nflow.py:141)
// as a standalone kernel: z[r][j] = N(0,1)(seed, row_offset + r, j), the counter-based stream of rnvp_prior.h.
// The MFMA inverse kernel makes the same draws in registers (rnvp_sample); this kernel serves the generic
// path, rnvp_prior_normal and the tests that compare the two.
#include "rnvp_common.h"
#include "rnvp_prior.h"

namespace rnvp {
namespace {

__global__ void __launch_bounds__(256)
k_prior_normal(uint64_t seed, int64_t row0, int64_t n, int d, int nblk, float *__restrict__ z) {
    const int64_t total = n * nblk;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / nblk;
        const int blk = (int)(t - r * nblk);
        float v[4];
        prior_normal4(seed, row0 + r, blk, v);
        float *o = z + r * d + 4 * blk;
        if ((d & 3) == 0 && ((uintptr_t)z & 15) == 0) {
            *reinterpret_cast<float4 *>(o) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (4 * blk + e < d) o[e] = v[e];
        }
    }
}

}  // namespace

int prior_normal(hipStream_t st, uint64_t seed, int64_t row0, int64_t n, int d, float *z) {
    const int nblk = (d + 3) / 4;
    const int64_t total = n * nblk;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_prior_normal, dim3((unsigned)blocks), dim3(256), 0, st, seed, row0, n, d, nblk, z);
    RNVP_HIP_TRY(hipGetLastError());
    return RNVP_OK;
}

}  // namespace rnvp
