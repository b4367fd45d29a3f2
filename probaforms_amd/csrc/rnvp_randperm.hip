// rnvp_randperm.hip -- the reference's epoch shuffle on the device: `torch.randperm(n, generator=g)` of a CPU generator, bit for bit.
//
// /root/reference/probaforms/models/realnvp.py:235 builds DataLoader(shuffle=True): every epoch its RandomSampler draws a seed
// from the global generator, seeds a private generator with it and calls torch.randperm(n, generator=...).  On the CPU that is
// (ATen native/TensorFactories.cpp randperm_cpu, n < 2^32 / 20) the identity followed by the sequential Fisher-Yates pass
//     for i = 0 .. n-2:   z = mt19937() % (n - i);   swap(r[i], r[i + z])
// -- 8.5 ms per million rows on the MI355X hosts, serial, once per epoch (probaforms_amd/_engine.py prefetches it on worker
// threads; the first epoch of a fit has nothing to hide it behind).  Here:
//   1. the n - 1 raw generator words come from the device twister of rnvp_prior_torch.hip (jump-ahead: 32 workgroups on one stream);
//   2. H[i] = i + temper(word i) % (n - i), r = identity;
//   3. the swaps run in parallel ROUNDS with deterministic reservations (Shun, Gu, Blelloch, Fineman, Gibbons, "Sequential random
//      permutation, list contraction and tree contraction are highly parallel", SODA 2015): every pending iteration i bids for its two
//      cells i and H[i] with priority "smaller i wins"; an iteration that holds both has no earlier pending iteration touching
//      either cell, so its swap is exactly the sequential one; losers bid again next round.  A third of the pending iterations
//      commits per round (49 rounds for n = 1M); the result is the sequential permutation whatever the thread timing.
//      Bids are 64-bit keys (round << 32 | ~i) under atomicMax, so a new round's bids override the old ones without a reset pass.
//   Grid-wide rounds (two launches each) run while the host's bound on the pending count (x 3/4 per round) exceeds two workgroups'
//   worth; the few hundred iterations left finish inside one workgroup.
// The Python host validates this path against torch.randperm once per process (probaforms_amd/_engine.py) and keeps the host shuffle
// when they differ.
#include "rnvp_common.h"

namespace rnvp {
namespace {

constexpr int kGlobalRounds = 48;        // at most; the host stops launching them once its bound on the pending count fits one workgroup
constexpr int kTailThreads = 1024;

__device__ __forceinline__ uint32_t temper(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}
__device__ __forceinline__ unsigned long long bid_key(int round, uint32_t i) {
    return ((unsigned long long)(uint32_t)(round + 1) << 32) | (unsigned long long)(0xffffffffu - i);
}

// raw[i] (bit patterns left by mt19937_raw_words) -> H[i]; r = identity; every iteration pending; bids cleared
__global__ void __launch_bounds__(256)
k_perm_init(int64_t n, const uint32_t *__restrict__ raw, uint32_t *__restrict__ H, int64_t *__restrict__ r,
            uint32_t *__restrict__ pending, unsigned long long *__restrict__ bids, uint32_t *__restrict__ counts) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) { counts[0] = (uint32_t)(n - 1); counts[1] = 0; }
    if (i >= n) return;
    r[i] = i;
    bids[i] = 0ull;
    if (i < n - 1) {
        H[i] = (uint32_t)i + temper(raw[i]) % (uint32_t)(n - i);
        pending[i] = (uint32_t)i;
    }
}

__global__ void __launch_bounds__(256)
k_perm_bid(int round, const uint32_t *__restrict__ H, const uint32_t *__restrict__ pending, const uint32_t *__restrict__ counts,
           uint32_t *__restrict__ counts_w, unsigned long long *__restrict__ bids) {
    const uint32_t cnt = counts[round & 1];
    if (blockIdx.x == 0 && threadIdx.x == 0) counts_w[(round + 1) & 1] = 0;       // this round's losers are counted from zero
    for (uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x; idx < cnt; idx += gridDim.x * blockDim.x) {   // any grid is correct
        const uint32_t i = pending[idx];
        const unsigned long long key = bid_key(round, i);
        atomicMax(&bids[i], key);
        atomicMax(&bids[H[i]], key);
    }
}

// the winners swap; the losers go to the other pending list (their order there does not matter: priorities are the indices)
__global__ void __launch_bounds__(256)
k_perm_commit(int round, const uint32_t *__restrict__ H, const uint32_t *__restrict__ pending, uint32_t *__restrict__ pending_next,
              uint32_t *__restrict__ counts, const unsigned long long *__restrict__ bids, int64_t *__restrict__ r) {
    const uint32_t cnt = counts[round & 1];
    const uint32_t lane = threadIdx.x & 63;
    // wave-uniform trip count: the losers of a wave claim their slots of the next list with ONE atomic (a counter hit by two thirds
    // of a million lanes individually is where this kernel's time went)
    for (uint32_t base = blockIdx.x * blockDim.x + (threadIdx.x & ~63u); base < cnt; base += gridDim.x * blockDim.x) {
        const uint32_t idx = base + lane;
        bool lose = false;
        uint32_t i = 0;
        if (idx < cnt) {
            i = pending[idx];
            const uint32_t j = H[i];
            const unsigned long long key = bid_key(round, i);
            if (bids[i] == key && bids[j] == key) {
                const int64_t a = r[i], b = r[j];
                r[i] = b; r[j] = a;
            } else {
                lose = true;
            }
        }
        const unsigned long long m = __ballot(lose);
        if (m) {
            uint32_t slot0 = 0;
            if (lane == (uint32_t)__builtin_ctzll(m)) slot0 = atomicAdd(&counts[(round + 1) & 1], (uint32_t)__builtin_popcountll(m));
            slot0 = __shfl(slot0, __builtin_ctzll(m));
            if (lose) pending_next[slot0 + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull))] = i;
        }
    }
}

// the rest inside one workgroup: the same rounds with a workgroup barrier in place of the kernel boundary
__global__ void __launch_bounds__(kTailThreads)
k_perm_tail(int round0, const uint32_t *__restrict__ H, uint32_t *__restrict__ listA, uint32_t *__restrict__ listB,
            uint32_t *__restrict__ counts, unsigned long long *bids, int64_t *r) {
    __shared__ uint32_t cnt_next;
    uint32_t *cur = (round0 & 1) ? listB : listA, *nxt = (round0 & 1) ? listA : listB;
    uint32_t cnt = counts[round0 & 1];
    for (int round = round0; cnt > 0; ++round) {
        if (threadIdx.x == 0) cnt_next = 0;
        __syncthreads();
        for (uint32_t idx = threadIdx.x; idx < cnt; idx += kTailThreads) {
            const uint32_t i = cur[idx];
            const unsigned long long key = bid_key(round, i);
            atomicMax(&bids[i], key);
            atomicMax(&bids[H[i]], key);
        }
        __threadfence();
        __syncthreads();
        for (uint32_t idx = threadIdx.x; idx < cnt; idx += kTailThreads) {
            const uint32_t i = cur[idx], j = H[i];
            const unsigned long long key = bid_key(round, i);
            // (atomic reads: the bids were written by other lanes' atomics of this very kernel)
            if (atomicMax(&bids[i], 0ull) == key && atomicMax(&bids[j], 0ull) == key) {
                const int64_t a = __hip_atomic_load(&r[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int64_t b = __hip_atomic_load(&r[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&r[i], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&r[j], a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                nxt[atomicAdd(&cnt_next, 1u)] = i;
            }
        }
        __threadfence();
        __syncthreads();
        cnt = cnt_next;
        uint32_t *t = cur; cur = nxt; nxt = t;
        __syncthreads();
    }
}

}  // namespace
}  // namespace rnvp

// n rows: H 4n | two pending lists 4n each | bids 8n | raw words 4n | counters | the twister's workspace
extern "C" size_t rnvp_randperm_workspace_bytes(int64_t n) {
    if (n < 1) n = 1;
    return rnvp::align_up((size_t)n * 4, 256) * 4 + rnvp::align_up((size_t)n * 8, 256) + 256 + rnvp_prior_torch_workspace_bytes();
}

extern "C" int rnvp_randperm_torch_cpu(void *stream, uint32_t *mt_state, int64_t n, int64_t *perm_out, void *workspace,
                                       size_t workspace_bytes) {
    using namespace rnvp;
    if (n < 1 || n >= (int64_t)(0xffffffffu / 20)) return RNVP_EUNSUPPORTED;       // torch shuffles larger n another way
    if (!mt_state || !perm_out) return RNVP_EINVAL;
    if (!workspace || workspace_bytes < rnvp_randperm_workspace_bytes(n)) return RNVP_EWORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    char *w = static_cast<char *>(workspace);
    const size_t a4 = align_up((size_t)n * 4, 256);
    uint32_t *H = reinterpret_cast<uint32_t *>(w); w += a4;
    uint32_t *listA = reinterpret_cast<uint32_t *>(w); w += a4;
    uint32_t *listB = reinterpret_cast<uint32_t *>(w); w += a4;
    uint32_t *raw = reinterpret_cast<uint32_t *>(w); w += a4;
    unsigned long long *bids = reinterpret_cast<unsigned long long *>(w); w += align_up((size_t)n * 8, 256);
    uint32_t *counts = reinterpret_cast<uint32_t *>(w); w += 256;
    void *mtws = w;
    if (n > 1) {
        const int rc = mt19937_raw_words(st, mt_state, n - 1, raw, mtws);
        if (rc) return rc;
    }
    const unsigned blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_perm_init, dim3(blocks), dim3(256), 0, st, n, raw, H, perm_out, listA, bids, counts);
    RNVP_HIP_TRY(hipGetLastError());
    if (n < 2) return RNVP_OK;
    // the pending count shrinks by a third per round: the grids shrink with it (the kernels stride over their list, so a grid
    // smaller than the list is slower, never wrong)
    double bound = (double)(n - 1);
    int round = 0;
    for (; round < kGlobalRounds && bound > 2.0 * kTailThreads; ++round) {
        const unsigned g = (unsigned)(((int64_t)bound + 255) / 256);
        uint32_t *cur = (round & 1) ? listB : listA, *nxt = (round & 1) ? listA : listB;
        hipLaunchKernelGGL(k_perm_bid, dim3(g), dim3(256), 0, st, round, H, cur, counts, counts, bids);
        hipLaunchKernelGGL(k_perm_commit, dim3(g), dim3(256), 0, st, round, H, cur, nxt, counts, bids, perm_out);
        RNVP_HIP_TRY(hipGetLastError());
        bound *= 0.75;
    }
    hipLaunchKernelGGL(k_perm_tail, dim3(1), dim3(kTailThreads), 0, st, round, H, listA, listB, counts, bids, perm_out);
    RNVP_HIP_TRY(hipGetLastError());
    return RNVP_OK;
}
