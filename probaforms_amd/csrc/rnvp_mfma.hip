// rnvp_mfma.hip -- register-chained f32 MFMA kernels for the RealNVP coupling stack (gfx950).
// Layout and rationale: rnvp_mfma.h.  Replaces RealNVPLayer.f / .g for every layer and the
// loops of NormalizingFlow.log_prob / .sample (/root/reference/probaforms/models/realnvp.py:
// 91-101,120-129; nflow.py:107-117,141-145) for d <= 64, cdim <= 16 (padded up to the tile
// geometry), one hidden layer, tanh, and the reference's alternating masks.
#include "rnvp_bx3.h"
#include "rnvp_mfma_layer.h"
#include "rnvp_mfma_pack.h"
#include "rnvp_prior.h"

// amdgpu_waves_per_eu(RNVP_WPE, RNVP_WPE): with at most 256 registers per wave hipcc keeps MFMA
// results in VGPRs (above that it allocates AGPRs and pays a v_accvgpr_read per value the VALU
// touches), and it stops trading instruction-level parallelism for a higher occupancy it cannot use.
#ifndef RNVP_WPE
#define RNVP_WPE 2
#endif

namespace rnvp {
namespace mfma {
namespace {

constexpr int kWaves = 4;                // waves per workgroup (one per SIMD)
constexpr int kMaxGrid = 2048;

// ---- weight packing ----------------------------------------------------------------------------
// One thread per float of the packed block (slot values: rnvp_mfma_pack.h); reads the flat reference-order parameters
// (include/rnvp_hip.h "params").  Runs at the head of every call so the packed copy always reflects the caller's current
// parameters (76 k .. 450 k floats: a few microseconds); inside rnvp_fit_epoch* only once per call -- from then on the
// training step's finish kernel re-packs what it updates.
__global__ void __launch_bounds__(256)
k_pack_weights(KShape k, Geo g, const float *__restrict__ params, float *__restrict__ packed, int *err) {
    if (err && blockIdx.x == 0 && threadIdx.x == 0) *err = 0;       // the training step's error word (rnvp_mfma_layer.h spin_nap)
    const int per = g.layer_floats;
    const int64_t total = (int64_t)per * k.L;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int l = (int)(t / per), idx = (int)(t - (int64_t)l * per);
        packed[t] = pack_value(k, g, l, idx, FlatParams(k, params, l));
    }
}

__global__ void __launch_bounds__(1024)
k_sum_partials(const float *__restrict__ part, int G, float scale, float *out) {
    // one workgroup, fixed order: thread t adds part[t], part[t + 1024], ...; wave butterflies; wave sums added in
    // wave order by lane 0 -- deterministic (a single wave walking 8192 partials took 32 us of dependent loads)
    __shared__ float wsum[16];
    const int t = threadIdx.x;
    float a = 0.f;
    for (int i = t; i < G; i += 1024) a += part[i];
    for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off);
    if ((t & 63) == 0) wsum[t >> 6] = a;
    __syncthreads();
    if (t == 0) {
        float r = 0.f;
        for (int w = 0; w < 16; ++w) r += wsum[w];
        out[0] = r * scale;
    }
}

// ---- whole stack: forward (+ log-det + prior) or inverse ---------------------------------------
template <int NF, int CQ, int R, bool INVERSE, int ACT>
__global__ void __launch_bounds__(kWaves * 64) __attribute__((amdgpu_waves_per_eu(RNVP_WPE, RNVP_WPE)))
k_mfma_flow(const float *__restrict__ wp, Geo g, int L, int alt, const float *x,
            const float *__restrict__ c, const int64_t *__restrict__ row_index, int64_t n,
            float *out_x, float *logdet_out, float *logp_out, float *part, uint64_t seed, int64_t row0) {
    // x carries no __restrict__: the inverse may run in place (out_x == x, include/rnvp_hip.h).
    // x == nullptr (inverse only): the rows are prior draws made here, z(seed, row0 + row, feature) of rnvp_prior.h.
    constexpr int D = 8 * NF, CD = 4 * CQ;               // padded sizes; g.d / g.c are the caller's
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, r = lane & 15;
    const int64_t rows_per_wg = (int64_t)kWaves * R * 16;
    const int64_t ngroups = (n + rows_per_wg - 1) / rows_per_wg;
    const float prior_c = 0.5f * (float)g.d * kLog2Pi;
    // exact fit and 16-byte aligned rows: vector row loads / stores; otherwise the guarded scalar path
    const bool full = (g.d == D) && (g.c == CD) && (((uintptr_t)x | (uintptr_t)out_x) & 15) == 0;
    const bool gen = INVERSE && x == nullptr;
    float wave_sum = 0.f;
    for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int64_t base = grp * rows_per_wg + (int64_t)wave * R * 16;
        if (base >= n) continue;                      // whole wave past the end (no barriers here)
        float xr[R][2 * NF], cr[R][CQ > 0 ? CQ : 1], ld[R];
#pragma unroll
        for (int rt = 0; rt < R; ++rt) {
            const int64_t row = base + rt * 16 + r;
            const bool valid = row < n;
            const int64_t src = valid ? (row_index ? row_index[row] : row) : 0;
            if (gen) {
                load_row<NF, CQ, false>(x, c, src, g.d, g.c, full, q, xr[rt], cr[rt]);
#pragma unroll
                for (int b = 0; b < 2 * NF / 4; ++b) {
                    float z4[4];
                    prior_normal4(seed, row0 + row, (q * 2 * NF) / 4 + b, z4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) xr[rt][4 * b + e] = (q * 2 * NF + 4 * b + e < g.d) ? z4[e] : 0.f;
                }
            } else {
                load_row<NF, CQ>(x, c, src, g.d, g.c, full, q, xr[rt], cr[rt]);
            }
            ld[rt] = 0.f;
        }
        for (int lp = 0; lp < L; ++lp) {
            const int l = INVERSE ? L - 1 - lp : lp;
            const float *W = wp + (size_t)l * g.layer_floats;
            if ((l + alt) & 1) layer_forward<NF, CQ, R, 1, INVERSE ? 1 : 0, ACT>(W, g, lane, xr, cr, ld, nullptr);
            else layer_forward<NF, CQ, R, 0, INVERSE ? 1 : 0, ACT>(W, g, lane, xr, cr, ld, nullptr);
        }
#pragma unroll
        for (int rt = 0; rt < R; ++rt) {
            const int64_t row = base + rt * 16 + r;
            const bool valid = row < n;
            if (out_x && valid) store_row<NF>(out_x, row, g.d, full, q, xr[rt]);
            if (!INVERSE) {
                float ss = 0.f;
#pragma unroll
                for (int v = 0; v < 2 * NF; ++v) ss = fmaf(xr[rt][v], xr[rt][v], ss);
                float l1 = ld[rt];
                // a row's features are spread over the 4 lane groups: add across q
                l1 += __shfl_xor(l1, 16); l1 += __shfl_xor(l1, 32);
                ss += __shfl_xor(ss, 16); ss += __shfl_xor(ss, 32);
                const float lp = l1 + (-0.5f * ss - prior_c);          // nflow.py:115
                if (valid && q == 0) {
                    if (logdet_out) logdet_out[row] = l1;
                    if (logp_out) logp_out[row] = lp;
                }
                if (part) {
                    float v = (valid && q == 0) ? lp : 0.f;
                    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
                    wave_sum += v;
                }
            }
        }
    }
    if (!INVERSE && part && lane == 0) part[blockIdx.x * kWaves + wave] = wave_sum;
}


// ---- tile-split forward / inverse: calls of up to kTsFlowMaxRows rows ------------------------------------------------
// Same idea as k_mfma_train_ts (rnvp_mfma_train.hip): such calls are latency chains of 2 * HT tile steps per layer in the
// kernel above; here a workgroup of kTsWaves waves takes 16 rows, every wave holds them, wave w runs a quarter of the
// hidden tiles of net w >> 2 and the partial net outputs meet in LDS once per layer.  Wave 0 writes the results.
#ifndef RNVP_TILE_SPLIT
#define RNVP_TILE_SPLIT 1
#endif
constexpr int64_t kTsFlowMaxRows = 4096;
template <int NF, int CQ, bool INVERSE, int ACT>
__global__ void __launch_bounds__(kTsWaves * 64) __attribute__((amdgpu_waves_per_eu(RNVP_WPE, RNVP_WPE)))
k_mfma_flow_ts(const float *__restrict__ wp, Geo g, int L, int alt, const float *x, const float *__restrict__ c,
               const int64_t *__restrict__ row_index, int64_t n, float *out_x, float *logdet_out, float *logp_out,
               float *part, uint64_t seed, int64_t row0) {
    constexpr int D = 8 * NF, CD = 4 * CQ, R = 1, XW = R * NF * 64;
    extern __shared__ __attribute__((aligned(16))) float lds[];        // 2 x kTsWaves x XW, double buffered by layer parity
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, r = lane & 15;
    const int tps = (g.HT + kTsSlices - 1) / kTsSlices, slice = wave & (kTsSlices - 1);
    const int tile_lo = slice * tps < g.HT ? slice * tps : g.HT;
    const int tile_hi = tile_lo + tps < g.HT ? tile_lo + tps : g.HT;
    const float prior_c = 0.5f * (float)g.d * kLog2Pi;
    const bool full = (g.d == D) && (g.c == CD) && (((uintptr_t)x | (uintptr_t)out_x) & 15) == 0;
    const bool gen = INVERSE && x == nullptr;
    const int64_t row = (int64_t)blockIdx.x * 16 + r;
    const bool valid = row < n;
    const int64_t src = valid ? (row_index ? row_index[row] : row) : 0;
    float xr[R][2 * NF], cr[R][CQ > 0 ? CQ : 1], ld[R];
    if (gen) {
        load_row<NF, CQ, false>(x, c, src, g.d, g.c, full, q, xr[0], cr[0]);
#pragma unroll
        for (int b = 0; b < 2 * NF / 4; ++b) {
            float z4[4];
            prior_normal4(seed, row0 + row, (q * 2 * NF) / 4 + b, z4);
#pragma unroll
            for (int e = 0; e < 4; ++e) xr[0][4 * b + e] = (q * 2 * NF + 4 * b + e < g.d) ? z4[e] : 0.f;
        }
    } else {
        load_row<NF, CQ>(x, c, src, g.d, g.c, full, q, xr[0], cr[0]);
    }
    ld[0] = 0.f;
    constexpr bool PRE = NF <= 4;   // a layer's opening fragments, requested one layer ahead (rnvp_mfma_layer.h)
    TilePre<NF, CQ> pre;
    if (PRE && tile_hi > tile_lo)
        load_tile_pre<NF, CQ>(wp + (size_t)(INVERSE ? L - 1 : 0) * g.layer_floats, g, lane, (wave >> 2) * g.HT + tile_lo, tile_hi - tile_lo, pre);
    for (int lp = 0; lp < L; ++lp) {
        const int l = INVERSE ? L - 1 - lp : lp;
        const float *W = wp + (size_t)l * g.layer_floats;
        const float *Wn = lp + 1 < L ? wp + (size_t)(INVERSE ? l - 1 : l + 1) * g.layer_floats : nullptr;
        float *rb = lds + (size_t)(lp & 1) * kTsWaves * XW;
        if ((l + alt) & 1) layer_forward_ts<NF, CQ, R, 1, INVERSE ? 1 : 0, ACT>(W, g, lane, wave, tile_lo, tile_hi - tile_lo, rb, xr, cr, ld, nullptr, Wn, pre, PRE);
        else layer_forward_ts<NF, CQ, R, 0, INVERSE ? 1 : 0, ACT>(W, g, lane, wave, tile_lo, tile_hi - tile_lo, rb, xr, cr, ld, nullptr, Wn, pre, PRE);
    }
    if (wave != 0) {
        if (!INVERSE && part && lane == 0 && wave < kWaves) part[blockIdx.x * kWaves + wave] = 0.f;
        return;
    }
    if (out_x && valid) store_row<NF>(out_x, row, g.d, full, q, xr[0]);
    if (!INVERSE) {
        float ss = 0.f;
#pragma unroll
        for (int v = 0; v < 2 * NF; ++v) ss = fmaf(xr[0][v], xr[0][v], ss);
        float l1 = ld[0];
        l1 += __shfl_xor(l1, 16); l1 += __shfl_xor(l1, 32);
        ss += __shfl_xor(ss, 16); ss += __shfl_xor(ss, 32);
        const float lpv = l1 + (-0.5f * ss - prior_c);          // nflow.py:115
        if (valid && q == 0) {
            if (logdet_out) logdet_out[row] = l1;
            if (logp_out) logp_out[row] = lpv;
        }
        if (part) {
            float v = (valid && q == 0) ? lpv : 0.f;
            v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
            if (lane == 0) part[blockIdx.x * kWaves] = v;
        }
    }
}

#ifndef RNVP_FLOW_R2
#define RNVP_FLOW_R2 4
#endif
template <int NF, int CQ> struct RowTiles { static constexpr int value = NF == 2 ? RNVP_FLOW_R2 : (NF == 4 ? 2 : 1); };

struct Launch {
    Geo g;
    int grid;
    float *packed;
    float *part;
};

size_t packed_bytes(const KShape &k) {
    const Geo g = make_geo(k.d, k.c, k.nout[0]);
    return align_up((size_t)g.layer_floats * k.L * sizeof(float), 256);
}


template <int NF, int CQ, bool INVERSE, int ACT, int R = RowTiles<NF, CQ>::value>
int launch_flow(hipStream_t st, const KShape &k, const Geo &g, const float *packed, const float *x,
                const float *c, const int64_t *row_index, int64_t n, float *out_x, float *logdet_out,
                float *logp_out, float *part, int *grid_out, uint64_t seed = 0, int64_t row0 = 0) {
    if constexpr (RNVP_TILE_SPLIT && R == RowTiles<NF, CQ>::value) {
        if (k.small_latency && n <= kTsFlowMaxRows && g.HT >= 3) {      // whole call, 1024 rows: C2 55 -> 24 us, C3 217 -> 49, C4 70 -> 42
            const int grid = (int)((n + 15) / 16);
            *grid_out = grid;
            const size_t lds_bytes = (size_t)2 * kTsWaves * NF * 64 * sizeof(float);
            note_dispatch(INVERSE ? RNVP_PROFILE_INVERSE : RNVP_PROFILE_FORWARD, "k_mfma_flow_ts", RNVP_VARIANT_TILESPLIT, 1, kTsWaves,
                          grid, RNVP_PREC_F32, n);
            const KernelEvents ev(INVERSE ? RNVP_PROFILE_INVERSE : RNVP_PROFILE_FORWARD);
            hipExtLaunchKernelGGL((k_mfma_flow_ts<NF, CQ, INVERSE, ACT>), dim3(grid), dim3(kTsWaves * 64), lds_bytes, st, ev.start,
                                  ev.stop, 0, packed, g, k.L, k.alt, x, c, row_index, n, out_x, logdet_out, logp_out, part, seed,
                                  row0);
            RNVP_HIP_TRY(hipGetLastError());
            return RNVP_OK;
        }
    }
    // Small calls: 4 row tiles per wave would leave CUs idle and one wave per SIMD; halve the tile
    // count so twice as many workgroups are in flight (measured on C2: 32768 rows 88 -> 63 us; from
    // 65536 rows up R = 4 wins).  A row's result does not depend on R.
    if constexpr (NF == 2 && R == 4) {
        if (n <= 32768)
            return launch_flow<NF, CQ, INVERSE, ACT, 2>(st, k, g, packed, x, c, row_index, n, out_x, logdet_out, logp_out,
                                                   part, grid_out, seed, row0);
    }
    const int64_t rows_per_wg = (int64_t)kWaves * R * 16;
    const int64_t ngroups = (n + rows_per_wg - 1) / rows_per_wg;
    const int grid = (int)(ngroups < kMaxGrid ? ngroups : kMaxGrid);
    *grid_out = grid;
    note_dispatch(INVERSE ? RNVP_PROFILE_INVERSE : RNVP_PROFILE_FORWARD, "k_mfma_flow", RNVP_VARIANT_ROWPAR, R, kWaves, grid,
                  RNVP_PREC_F32, n);
    {
        const KernelEvents ev(INVERSE ? RNVP_PROFILE_INVERSE : RNVP_PROFILE_FORWARD);
        hipExtLaunchKernelGGL((k_mfma_flow<NF, CQ, R, INVERSE, ACT>), dim3(grid), dim3(kWaves * 64), 0, st, ev.start, ev.stop, 0,
                              packed, g, k.L, k.alt, x, c, row_index, n, out_x, logdet_out, logp_out, part, seed, row0);
    }
    RNVP_HIP_TRY(hipGetLastError());
    return RNVP_OK;
}

template <bool INVERSE>
int dispatch_flow(hipStream_t st, const KShape &k, const Geo &g, const float *packed, const float *x,
                  const float *c, const int64_t *row_index, int64_t n, float *out_x, float *logdet_out,
                  float *logp_out, float *part, int *grid_out, uint64_t seed = 0, int64_t row0 = 0) {
#define RNVP_CASE(nf, cq)                                                                                       \
    if (g.NF == nf && g.CQ == cq) {                                                                             \
        if (k.act == RNVP_ACT_TANH)                                                                             \
            return launch_flow<nf, cq, INVERSE, 0>(st, k, g, packed, x, c, row_index, n, out_x, logdet_out, logp_out, \
                                                   part, grid_out, seed, row0);                                 \
        return launch_flow<nf, cq, INVERSE, 1>(st, k, g, packed, x, c, row_index, n, out_x, logdet_out, logp_out,  \
                                               part, grid_out, seed, row0);                                     \
    }
    RNVP_CASE(2, 1) RNVP_CASE(2, 0) RNVP_CASE(4, 2) RNVP_CASE(8, 4)
#undef RNVP_CASE
    return RNVP_EUNSUPPORTED;
}

}  // namespace

int pack_weights(hipStream_t st, const KShape &k, const Geo &g, const float *params, float *packed, int *err) {
    const int64_t total = (int64_t)g.layer_floats * k.L;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_pack_weights, dim3(blocks), dim3(256), 0, st, k, g, params, packed, err);
    RNVP_HIP_TRY(hipGetLastError());
    return RNVP_OK;
}

bool supported(const KShape &k) {
    if (!k.alt || k.nh != 1) return false;             // tanh and ReLU (realnvp.py:32-37) both run here
    int NF, CQ;
    return pick_tiles(k.d, k.c, &NF, &CQ);       // d <= 64, cdim <= 16; everything else is padding
}

// forward / inverse: [packed weights of either kernel family | per-wave log-prob partials]
static size_t flow_packed_bytes(const KShape &k) {
    const size_t a = packed_bytes(k), b = bx3::packed_bytes(k);
    return a > b ? a : b;
}
static_assert(kMaxGrid * kWaves >= bx3::kMaxGridBx3 * bx3::kWavesBx3, "partials buffer is sized for both kernel families");

size_t workspace_bytes(const KShape &k, int op, int64_t max_rows) {
    if (op == RNVP_OP_TRAIN) return train_workspace_bytes(k, max_rows);
    return flow_packed_bytes(k) + align_up((size_t)kMaxGrid * kWaves * sizeof(float), 256);
}

// rnvp_shape.small_calls = RNVP_SMALL_LATENCY: short calls run the tile-split f32 kernel whatever `auto` would pick for
// long ones (the bx3 kernels need whole stages per workgroup; an explicit precision = bx3 is honoured)
static bool ts_flow(const KShape &k, const Geo &g, int64_t n) {
    return RNVP_TILE_SPLIT && k.small_latency && n <= kTsFlowMaxRows && g.HT >= 3 && (k.prec_flow != RNVP_PREC_BX3 || k.prec_auto);
}

int forward(hipStream_t st, const KShape &k, const float *params, const float *x, const float *c,
            const int64_t *row_index, int64_t n, float *z_out, float *logdet_out, float *logp_out,
            float *logp_sum, void *ws, size_t ws_bytes) {
    if (!ws || ws_bytes < workspace_bytes(k, RNVP_OP_FORWARD, n)) return RNVP_EWORKSPACE;
    const Geo g = make_geo(k.d, k.c, k.nout[0]);
    float *packed = static_cast<float *>(ws);
    float *part = reinterpret_cast<float *>(static_cast<char *>(ws) + flow_packed_bytes(k));
    int grid = 0, waves = kWaves, rc;
    if (k.prec_flow == RNVP_PREC_BX3 && !ts_flow(k, g, n)) {
        rc = bx3::forward(st, k, params, x, c, row_index, n, z_out, logdet_out, logp_out, logp_sum ? part : nullptr, &grid,
                          &waves, ws);
    } else {
        rc = pack_weights(st, k, g, params, packed);
        if (rc) return rc;
        rc = dispatch_flow<false>(st, k, g, packed, x, c, row_index, n, z_out, logdet_out, logp_out,
                                  logp_sum ? part : nullptr, &grid);
    }
    if (rc) return rc;
    if (logp_sum) {
        hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(1024), 0, st, part, grid * waves, 1.0f, logp_sum);
        RNVP_HIP_TRY(hipGetLastError());
    }
    return RNVP_OK;
}

int inverse(hipStream_t st, const KShape &k, const float *params, const float *z, const float *c,
            int64_t n, float *x_out, void *ws, size_t ws_bytes) {
    if (!ws || ws_bytes < workspace_bytes(k, RNVP_OP_INVERSE, n)) return RNVP_EWORKSPACE;
    const Geo g = make_geo(k.d, k.c, k.nout[0]);
    if (k.prec_flow == RNVP_PREC_BX3 && !ts_flow(k, g, n)) return bx3::inverse(st, k, params, z, c, n, x_out, 0, 0, ws);
    float *packed = static_cast<float *>(ws);
    int rc = pack_weights(st, k, g, params, packed);
    if (rc) return rc;
    int grid = 0;
    return dispatch_flow<true>(st, k, g, packed, z, c, nullptr, n, x_out, nullptr, nullptr, nullptr, &grid);
}

// prior draw fused into the inverse: x_out = g(z(seed, row0 + row, .), c); z is never written to memory
int sample(hipStream_t st, const KShape &k, const float *params, const float *c, int64_t n, uint64_t seed,
           int64_t row0, float *x_out, void *ws, size_t ws_bytes) {
    if (!ws || ws_bytes < workspace_bytes(k, RNVP_OP_INVERSE, n)) return RNVP_EWORKSPACE;
    const Geo g = make_geo(k.d, k.c, k.nout[0]);
    if (k.prec_flow == RNVP_PREC_BX3 && !ts_flow(k, g, n)) return bx3::inverse(st, k, params, nullptr, c, n, x_out, seed, row0, ws);
    float *packed = static_cast<float *>(ws);
    int rc = pack_weights(st, k, g, params, packed);
    if (rc) return rc;
    int grid = 0;
    return dispatch_flow<true>(st, k, g, packed, nullptr, c, nullptr, n, x_out, nullptr, nullptr, nullptr, &grid, seed,
                               row0);
}

}  // namespace mfma
}  // namespace rnvp
