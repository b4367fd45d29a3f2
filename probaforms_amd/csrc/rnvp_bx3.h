// rnvp_bx3.h -- geometry of the "bx3" forward / inverse kernels (gfx950): the coupling stack of rnvp_mfma.h with
//   * the first Linear of every s/t net (GEMM1: [x_masked || c] -> hidden) on v_mfma_f32_16x16x32_bf16 with both
//     operands split into THREE bf16 terms (x = x1 + x2 + x3, exact: 3 x 8 mantissa bits) and the six products
//     a1b1, a1b2, a2b1, a1b3, a2b2, a3b1 laid side by side along K -- every dropped product is below 2^-24 of the
//     result, i.e. float32-level accuracy at 1/4 of the f32 MFMA's matrix-pipe time (bf16 MFMA also leaves the VALU
//     free for the tanh, which the f32-input MFMA does not: scripts/micro/bf16_overlap.hip);
//   * the weights of one (layer, net, chunk of hidden tiles) STAGED IN LDS once per workgroup by LDS-DMA
//     (global_load_lds_dwordx4, double buffered) and read from there by all 8 waves, instead of every wave
//     streaming its own fragments through the 64 B/clk vector L1.
// The second Linear (hidden -> s/t) keeps the f32 MFMA forms of rnvp_mfma_layer.h: its B operand is the tanh output
// of this very tile, and splitting that on the fly costs the VALU what the matrix pipe would save.  Measured with
// GEMM2 on the same six-product scheme (g2b, RNVP_BX3_G2B_MIN_NF; kept as a build switch): C4 2.51 -> 2.35-2.46 ms
// per 1M rows but log-prob MAE vs float64 9e-6 -> 1.6e-5 (the activations' third term is truncated); C3 slower.
// Not worth the accuracy margin: off.
//
// K slots of GEMM1 (both operands): input value k of lane group q (k < NF: conditioning feature slot k;
// k >= NF: condition k - NF) owns dwords 3k .. 3k+2 of the lane's slot list, each dword two bf16 slots:
//     dword 3k+0 : B = (b1, b2)   A = (a1, a1)
//     dword 3k+1 : B = (b1, b3)   A = (a2, a1)
//     dword 3k+2 : B = (b2, b1)   A = (a2, a3)
// (low half first).  NI = ceil(3*KS1 / 4) MFMAs of 8 slots per lane cover them; unused slots are zero in A.
#pragma once
#include "rnvp_mfma.h"

namespace rnvp {
namespace bx3 {

struct Geo3 {
    int d, c, h;
    int NF, CQ, HT, KS1;
    int NI;        // 16x16x32 MFMAs per GEMM1 tile
    int NA2;       // f4 fragments of GEMM2 per hidden tile (2 for the 4x4x1 form at NF == 2; OTL f32 fragments, or
                   // 3 * OTL split-bf16 ones where the second Linear runs on bf16 MFMA too: g2b)
    int g2b;       // second Linear on split-bf16 MFMA (wide outputs: the f32 MFMA time exceeds the split's VALU time)
    int NT2;       // out tiles (as mfma::Geo)
    int TC;        // hidden tiles per stage
    int NCH;       // stages per net: ceil(HT / TC)
    int tile_dw;   // dwords of one tile record: (NI + NA2) * 256
    int SD;        // dwords of one stage: TC * tile_dw + the b1 block ([tile][q][4] f32, padded to one 256-dword piece)
    int NP;        // 1 KiB pieces per stage (SD / 256)
    int b2_floats; // per layer: NT2 * 16
};

#ifndef RNVP_BX3_G2B_MIN_NF
#define RNVP_BX3_G2B_MIN_NF 1000
#endif
__host__ __device__ constexpr bool g2b_for(int NF) { return NF >= RNVP_BX3_G2B_MIN_NF; }

__host__ __device__ inline Geo3 make_geo3(int d, int c, int h) {
    Geo3 g;
    g.d = d; g.c = c; g.h = h;
    g.NF = 2; g.CQ = 0;
    mfma::pick_tiles(d, c, &g.NF, &g.CQ);
    g.HT = (h + 15) / 16;
    g.KS1 = g.NF + g.CQ;
    g.NI = (3 * g.KS1 + 3) / 4;
    const int OTL = g.NF >= 4 ? g.NF / 4 : 1;
    g.g2b = g2b_for(g.NF);
    g.NA2 = g.NF == 2 ? 2 : (g.g2b ? 3 * OTL : OTL);
    g.NT2 = g.NF >= 4 ? 2 * OTL : 1;
    g.tile_dw = (g.NI + g.NA2) * 256;
    int tc = 8;
    while (tc > 1 && (tc * g.tile_dw + 256) * 4 > 64 * 1024) tc >>= 1;
    while (tc > 1 && tc / 2 >= g.HT) tc >>= 1;
    g.TC = tc;
    g.NCH = (g.HT + tc - 1) / tc;
    g.SD = tc * g.tile_dw + 256;
    g.NP = g.SD / 256;
    g.b2_floats = g.NT2 * 16;
    return g;
}

bool supported(const KShape &k);
size_t packed_bytes(const KShape &k);          // stages + b2 blocks
int forward(hipStream_t st, const KShape &k, const float *params, const float *x, const float *c,
            const int64_t *row_index, int64_t n, float *z_out, float *logdet_out, float *logp_out, float *part,
            int *grid_out, int *waves_out, void *packed);
int inverse(hipStream_t st, const KShape &k, const float *params, const float *z, const float *c, int64_t n,
            float *x_out, uint64_t seed, int64_t row0, void *packed);
constexpr int kWavesBx3 = 8;
constexpr int kMaxGridBx3 = 1024;

}  // namespace bx3
}  // namespace rnvp
