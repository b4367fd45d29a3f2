// rnvp_prior_torch.hip -- the REFERENCE's prior stream on the device: `torch.randn(count)` of a CPU generator
// (/root/reference/probaforms/models/nflow.py:141 `X = self.prior.sample((n,))`: MultivariateNormal(0, I).sample = randn on the
// global CPU generator), bit for bit, without the host's serial 2 ns per number.
//
// What torch's CPU kernel does for a contiguous float tensor of at least 16 elements (ATen native/cpu/DistributionTemplates.h
// normal_fill_AVX2: the path taken on AVX2 and AVX-512 hosts alike):
//   1. data[i] = (mt19937() & 0xFFFFFF) * 2^-24 for every i                        -- one 32-bit draw per element;
//   2. per block of 16: u1 = 1 - data[j], u2 = data[j + 8] (j < 8);  radius = sqrt(-2 log256_ps(u1)),
//      theta = float(2 pi) u2;  (sin, cos) = sincos256_ps(theta);  data[j] = radius cos,  data[j + 8] = radius sin;
//   3. a count that is not a multiple of 16 redraws the LAST 16 elements from 16 fresh draws.
// log256_ps / sincos256_ps (avx_mathfun.h) are float32 polynomials: restated below operation by operation, with the multiply-adds
// torch's build contracts, they give the host's bits (pinned on the CPU side by oracle/prior_torch_oracle.c against torch.randn,
// on the device by tests/test_prior_torch.py against torch.randn).  The Mersenne Twister itself is serial between its 624-word
// blocks; inside a block the recurrence x[k + 624] = x[k + 397] ^ twist(x[k], x[k + 1]) has three phases of up to 227 independent
// words, each needing only the same thread's previous result besides words of the old block: one workgroup, state double
// buffered in LDS, ONE barrier per block.
// The Python host (probaforms_amd/models/nflow.py HostStreamOnDevice) validates this path against torch.randn on a scratch
// generator once per process and keeps the host draw when they differ (another torch build).
#include "rnvp_common.h"
#include "rnvp_mt19937_jump.h"

namespace rnvp {
namespace {

constexpr int kN = 624, kM = 397;
constexpr int kMtThreads = 256;

__device__ __forceinline__ uint32_t twist(uint32_t u, uint32_t v, uint32_t far) {
    const uint32_t y = (u & 0x80000000u) | (v & 0x7fffffffu);
    return far ^ (y >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u);
}
__device__ __forceinline__ uint32_t temper(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

__device__ __forceinline__ float uniform24(uint32_t word) {
    return (float)(temper(word) & 0xffffffu) * 5.9604644775390625e-08f;        // (x & (2^24 - 1)) * 2^-24: exact
}

// One block step: thread t < 227 forms the new words t, 227 + t and 454 + t (thread 169: word 623) -- each needs the thread's OWN
// previous result as its far operand (new[k - 227]) and otherwise only words of the OLD block, so the three dependent phases of a
// block run register to register without a barrier; word 623 also needs the new word 0, which its thread recomputes.
__device__ __forceinline__ void twist_block(const uint32_t *o, uint32_t *nw, int t, uint32_t &a, uint32_t &b, uint32_t &c) {
    constexpr int D = kN - kM;            // 227
    if (t < D) {
        a = twist(o[t], o[t + 1], o[t + kM]);
        b = twist(o[D + t], o[D + t + 1], a);
        nw[t] = a;
        nw[D + t] = b;
        if (t < kN - 1 - 2 * D) {
            c = twist(o[2 * D + t], o[2 * D + t + 1], b);
            nw[2 * D + t] = c;
        } else if (t == kN - 1 - 2 * D) {            // word 623: its neighbour is the NEW word 0, its far word the new word 396 = b
            const uint32_t n0 = twist(o[0], o[1], o[kM]);
            c = twist(o[kN - 1], n0, b);
            nw[kN - 1] = c;
        }
    }
}

// ---- several workgroups on ONE stream: jump-ahead -----------------------------------------------------------------------------
// The twister is serial from block to block, but its raw word sequence W is an F2-linear recurrence: with g_J(x) = x^J mod phi(x)
// (phi: the characteristic polynomial, degree 19937), W[J + j] = XOR over the set bits i of g_J of W[i + j] for every j.  So the
// block k * B twists after X0 (B: the jump unit, 64 / 256 / 1024 blocks) is a binary convolution of the 33 blocks that follow X0 with a precomputed polynomial
// (rnvp_mt19937_jump.h, generated and self-checked by scripts/mt19937_jump_poly.py) -- and workgroup k of k_mt19937_uniform can start
// there while workgroup 0 is still at the beginning.  k_mt_jump: one workgroup per (segment k >= 1, quarter of the polynomial):
// builds the 33 blocks in LDS (82 KB), thread j XORs W[i + j] over its quarter's set bits; the quarters are XORed by the consumer.
constexpr int kJumpParts = 4;
constexpr int kJumpThreads = 640;
constexpr int kWBlocks = 33;                              // 19937 + 624 <= 33 * 624 raw words
constexpr int kMtMaxSeg = kMtJumpPolys + 1;               // segments of one round: k = 0 .. 31

__global__ void __launch_bounds__(kJumpThreads)
k_mt_jump(const uint32_t *__restrict__ state_in, uint32_t *__restrict__ partial, int unit_blocks) {
    const uint32_t (*polys)[624] = unit_blocks == 64 ? kMtJumpPolyB64 : (unit_blocks == 256 ? kMtJumpPolyB256 : kMtJumpPolyB1024);
    extern __shared__ uint32_t W[];                       // kWBlocks * 624 words
    const int t = threadIdx.x;
    const int k = blockIdx.x / kJumpParts + 1, part = blockIdx.x % kJumpParts;
    for (int i = t; i < kN; i += kJumpThreads) W[i] = state_in[i];
    __syncthreads();
    for (int b = 0; b + 1 < kWBlocks; ++b) {
        uint32_t a, bb, c;
        twist_block(W + b * kN, W + (b + 1) * kN, t, a, bb, c);
        __syncthreads();
    }
    if (t < kN) {
        constexpr int kPer = kN / kJumpParts;             // 156 polynomial words per quarter
        const uint32_t *g = polys[k - 1] + part * kPer;
        const uint32_t *w = W + 32 * part * kPer + t;
        uint32_t acc = 0u;
        for (int i = 0; i < kPer; ++i) {
            uint32_t p = g[i];                            // (uniform over the workgroup: no divergence)
            while (p) {
                const int bit = __ffs(p) - 1;
                acc ^= w[32 * i + bit];
                p &= p - 1;
            }
        }
        partial[((size_t)(k - 1) * kJumpParts + part) * kN + t] = acc;
    }
}

// Workgroup k produces its segment of the stream that starts at (state_in, position state_in[624]): segment 0 the rest of the current
// block and the B blocks after it, segment k >= 1 the blocks k B + 1 .. (k + 1) B (B = seg_blocks, the jump unit), from the jumped
// state k_mt_jump left in `partial`.  `count` uniforms go to out, `tail` more (0 or 16) to tail_out -- as RAW 32-bit words (tempering
// and the conversion to a 24-bit uniform are left to the parallel kernel below: this one is bound by its instruction count); the
// workgroup that produces the last word leaves the advanced state in state_out.  Whole blocks that land in `out` go straight from the
// registers that formed them (one barrier per block); the partial blocks at either end go through LDS.
__global__ void __launch_bounds__(kMtThreads)
k_mt19937_uniform(const uint32_t *__restrict__ state_in, const uint32_t *__restrict__ partial, uint32_t *__restrict__ state_out,
                  int64_t count, int tail, float *__restrict__ out, float *__restrict__ tail_out, int seg_blocks) {
    constexpr int D = kN - kM;
    const int64_t kSegWords = (int64_t)seg_blocks * kN;
    __shared__ uint32_t S[2][kN];
    const int t = threadIdx.x, k = blockIdx.x;
    const int pos0 = (int)state_in[kN];
    const int64_t total = count + tail, first = kN - pos0;
    const int64_t lo = k == 0 ? 0 : first + (int64_t)k * kSegWords;
    int64_t hi = first + (int64_t)(k + 1) * kSegWords;
    if (lo >= total) return;
    if (hi > total) hi = total;
    for (int i = t; i < kN; i += kMtThreads) {
        uint32_t v;
        if (k == 0) v = state_in[i];
        else {
            const uint32_t *pp = partial + (size_t)(k - 1) * kJumpParts * kN + i;
            v = pp[0];
#pragma unroll
            for (int q = 1; q < kJumpParts; ++q) v ^= pp[q * kN];
        }
        S[0][i] = v;
    }
    int pos = k == 0 ? pos0 : kN;
    int cur = 0;
    __syncthreads();
    int64_t done = lo;
    while (done < hi) {
        if (pos >= kN && done + kN <= hi && done + kN <= count) {           // a whole block for `out`
            uint32_t a = 0, b = 0, c = 0;
            twist_block(S[cur], S[cur ^ 1], t, a, b, c);
            float *dst = out + done;
            if (t < D) {                                 // raw words: the parallel kernel tempers and converts them
                dst[t] = __uint_as_float(a);
                dst[D + t] = __uint_as_float(b);
                if (t < kN - 1 - 2 * D) dst[2 * D + t] = __uint_as_float(c);
                else if (t == kN - 1 - 2 * D) dst[kN - 1] = __uint_as_float(c);
            }
            __syncthreads();
            cur ^= 1;
            done += kN;                                  // pos stays at 624: the block is used up
            continue;
        }
        if (pos >= kN) {
            uint32_t a, b, c;
            twist_block(S[cur], S[cur ^ 1], t, a, b, c);
            __syncthreads();
            cur ^= 1;
            pos = 0;
        }
        const int64_t left = hi - done;
        const int m = (int)((int64_t)(kN - pos) < left ? (kN - pos) : left);
        for (int i = t; i < m; i += kMtThreads) {
            const float u = __uint_as_float(S[cur][pos + i]);
            const int64_t g = done + i;
            if (g < count) out[g] = u; else tail_out[g - count] = u;
        }
        pos += m;
        done += m;
    }
    if (hi == total) {                                   // this workgroup produced the last word: hand the state on
        __syncthreads();
        for (int i = t; i < kN; i += kMtThreads) state_out[i] = S[cur][i];
        if (t == 0) state_out[kN] = (uint32_t)pos;
    }
}

// ---- log256_ps / sincos256_ps of ATen/native/cpu/avx_mathfun.h (cephes single-precision polynomials), one lane -----------------
// Plain float32 arithmetic; every fmaf() below is a multiply-add that GCC contracts in torch's build of that header (-mfma: the
// FIRST multiplication feeding an addition is fused), everything else stays a separately rounded multiply / add (contraction is
// switched off for these functions).  oracle/prior_torch_oracle.c holds the same statements for the CPU and is pinned against
// torch.randn bit for bit (tests/test_prior_torch.py).
__device__ __forceinline__ float log_ps(float x) {          // x in [2^-24, 1]
#pragma clang fp contract(off)
    uint32_t xi = __float_as_uint(x);
    int32_t imm0 = (int32_t)(xi >> 23);
    xi = (xi & ~0x7f800000u) | 0x3f000000u;
    x = __uint_as_float(xi);
    imm0 -= 0x7f;
    float e = (float)imm0 + 1.0f;
    const bool mask = x < 0.707106781186547524f;
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

__device__ __forceinline__ void sincos_ps(float x, float *s, float *c) {        // x >= 0
#pragma clang fp contract(off)
    float y = x * 1.27323954473516f;
    int32_t imm2 = (int32_t)y;
    imm2 = (imm2 + 1) & ~1;
    y = (float)imm2;
    const uint32_t sign_bit_sin = ((uint32_t)(imm2 & 4)) << 29;
    const bool poly_mask = (imm2 & 2) == 0;
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
    *s = __uint_as_float(__float_as_uint(xs) ^ sign_bit_sin);
    *c = __uint_as_float(__float_as_uint(xc) ^ sign_bit_cos);
}

// one thread per (block of 16, j < 8): the pair data[16 b + j], data[16 b + j + 8].  tail_only = 0: the count / 16 whole blocks
// (with a tail, the elements from count - 16 on are left alone: they are redrawn); tail_only = 1 (a SECOND launch, after the first
// has read everything it needs): the last 16 elements from the 16 extra words.
__global__ void __launch_bounds__(256)
k_normal_fill_16(float *__restrict__ data, int64_t count, const float *__restrict__ tail_u, int tail_only) {
#pragma clang fp contract(off)
    const int64_t nblk = count / 16;
    const bool has_tail = (count % 16) != 0;
    const int64_t pairs = tail_only ? 8 : nblk * 8;
    const int64_t keep_below = has_tail ? count - 16 : count;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < pairs; p += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = p >> 3;
        const int j = (int)(p & 7);
        // (the twister kernel left raw words: temper, keep 24 bits, scale -- uniform_real_distribution<float>)
        const float ua = uniform24(__float_as_uint(tail_only ? tail_u[j] : data[16 * b + j]));
        const float ub = uniform24(__float_as_uint(tail_only ? tail_u[j + 8] : data[16 * b + j + 8]));
        const float u1 = 1.0f - ua;
        const float radius = sqrtf(-2.0f * log_ps(u1));
        const float theta = 6.283185307179586f * ub;                        // float(2 pi) * u2
        float sn, cs;
        sincos_ps(theta, &sn, &cs);
        const float za = fmaf(radius * cs, 1.0f, 0.0f), zb = fmaf(radius * sn, 1.0f, 0.0f);     // fmadd(radius cos, std, mean): -0 -> +0
        const int64_t ia = tail_only ? count - 16 + j : 16 * b + j, ib = ia + 8;
        if (tail_only || ia < keep_below) data[ia] = za;
        if (tail_only || ib < keep_below) data[ib] = zb;
    }
}

}  // namespace

// The jump unit for a stream of `words`: the smallest one whose 32 segments cover it in one round (a short stream wants short
// segments: 1M words are 1 603 blocks -- two workgroups walking 800 blocks each with the 1024-block unit, twenty-six walking 64 with
// the 64-block one), else the largest
struct JumpUnit { int blocks; };
static JumpUnit pick_jump_unit(int64_t words) {
    const int64_t blocks = (words + kN - 1) / kN + 1;
    if (blocks <= (int64_t)kMtMaxSeg * 64) return JumpUnit{64};
    if (blocks <= (int64_t)kMtMaxSeg * 256) return JumpUnit{256};
    return JumpUnit{1024};
}

// `count` RAW (untempered) 32-bit outputs of the mt19937 whose state is mt_state [625: 624 words + the position], written to out as
// bit patterns; the advanced state is left in mt_state.  workspace: rnvp_prior_torch_workspace_bytes().  (rnvp_randperm.hip: the
// DataLoader shuffle; the normals above are the other user.)
int mt19937_raw_words(hipStream_t st, uint32_t *mt_state, int64_t count, uint32_t *out, void *workspace) {
    uint32_t *state_in = static_cast<uint32_t *>(workspace), *partial = state_in + 640;
    static std::atomic<uint64_t> attr{0};
    const size_t jump_lds = (size_t)kWBlocks * kN * sizeof(uint32_t);
    const int arc = allow_big_lds(reinterpret_cast<const void *>(k_mt_jump), 160 * 1024, attr);
    if (arc) return arc;
    const JumpUnit unit = pick_jump_unit(count);
    const int64_t kSegWords = (int64_t)unit.blocks * kN, kRound = (int64_t)kMtMaxSeg * kSegWords - 1024;
    for (int64_t r0 = 0; r0 < count; r0 += kRound) {
        const int64_t cnt = count - r0 < kRound ? count - r0 : kRound;
        RNVP_HIP_TRY(hipMemcpyAsync(state_in, mt_state, (kN + 1) * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
        int64_t nseg = (cnt + kSegWords - 1) / kSegWords;
        if (nseg > kMtMaxSeg) nseg = kMtMaxSeg;
        if (nseg > 1) {
            hipLaunchKernelGGL(k_mt_jump, dim3((unsigned)((nseg - 1) * kJumpParts)), dim3(kJumpThreads), jump_lds, st, state_in, partial, unit.blocks);
            RNVP_HIP_TRY(hipGetLastError());
        }
        hipLaunchKernelGGL(k_mt19937_uniform, dim3((unsigned)nseg), dim3(kMtThreads), 0, st, state_in, partial, mt_state, cnt, 0,
                           reinterpret_cast<float *>(out) + r0, (float *)nullptr, unit.blocks);
        RNVP_HIP_TRY(hipGetLastError());
    }
    return RNVP_OK;
}

}  // namespace rnvp

extern "C" size_t rnvp_prior_torch_workspace_bytes(void) {
    return ((size_t)640 + (size_t)rnvp::kMtJumpPolys * rnvp::kJumpParts * rnvp::kN) * sizeof(uint32_t);
}

extern "C" int rnvp_prior_normal_torch_cpu(void *stream, uint32_t *mt_state, int64_t count, float *z_out, float *tail16,
                                           void *workspace, size_t workspace_bytes) {
    using namespace rnvp;
    if (count < 16) return RNVP_EUNSUPPORTED;          // torch takes another path (a cached double-precision Box-Muller) there
    if (!mt_state || !z_out || !tail16) return RNVP_EINVAL;
    if (!workspace || workspace_bytes < rnvp_prior_torch_workspace_bytes()) return RNVP_EWORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    uint32_t *state_in = static_cast<uint32_t *>(workspace), *partial = state_in + 640;
    static std::atomic<uint64_t> attr{0};
    const size_t jump_lds = (size_t)kWBlocks * kN * sizeof(uint32_t);
    {
        const int arc = allow_big_lds(reinterpret_cast<const void *>(k_mt_jump), 160 * 1024, attr);
        if (arc) return arc;
    }
    // rounds of at most kMtMaxSeg segments (20.4M words): every round continues from the state the previous one left
    // (a round stops 1024 words short of the segments' capacity so that the 16 extra words of a redrawn tail always fit)
    const JumpUnit unit = pick_jump_unit(count + 16);
    const int64_t kSegWords = (int64_t)unit.blocks * kN, kRound = (int64_t)kMtMaxSeg * kSegWords - 1024;
    for (int64_t r0 = 0; r0 < count; r0 += kRound) {
        const int64_t cnt = count - r0 < kRound ? count - r0 : kRound;
        const bool last = r0 + cnt >= count;
        const int tail = (last && (count % 16)) ? 16 : 0;
        RNVP_HIP_TRY(hipMemcpyAsync(state_in, mt_state, (kN + 1) * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
        // the current block may be nearly unread (position 0), so cnt + tail words need at most this many segments
        int64_t nseg = (cnt + tail + kSegWords - 1) / kSegWords;
        if (nseg > kMtMaxSeg) nseg = kMtMaxSeg;
        if (nseg > 1) {
            hipLaunchKernelGGL(k_mt_jump, dim3((unsigned)((nseg - 1) * kJumpParts)), dim3(kJumpThreads), jump_lds, st, state_in, partial, unit.blocks);
            RNVP_HIP_TRY(hipGetLastError());
        }
        hipLaunchKernelGGL(k_mt19937_uniform, dim3((unsigned)nseg), dim3(kMtThreads), 0, st, state_in, partial, mt_state, cnt, tail,
                           z_out + r0, tail16, unit.blocks);
        RNVP_HIP_TRY(hipGetLastError());
    }
    const int tail = (count % 16) ? 16 : 0;
    const int64_t pairs = (count / 16) * 8;
    int64_t blocks = (pairs + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_normal_fill_16, dim3((unsigned)blocks), dim3(256), 0, st, z_out, count, tail16, 0);
    RNVP_HIP_TRY(hipGetLastError());
    if (tail) {         // the redrawn last 16 overlap the last whole block's inputs: after it, in stream order
        hipLaunchKernelGGL(k_normal_fill_16, dim3(1), dim3(64), 0, st, z_out, count, tail16, 1);
        RNVP_HIP_TRY(hipGetLastError());
    }
    return RNVP_OK;
}
