// rnvp_mfma.h -- geometry of the register-chained f32 MFMA path (gfx950, wave64).
//
// Shapes: any d <= 64, cdim <= 16, one hidden layer of any width, tanh or ReLU, alternating masks; sizes are
// padded up to the tile geometry (d -> 16/32/64, cdim -> 0/4/8/16, h -> multiple of 16) with zero
// weights, so the README shape (d=2, c=1, h=10) and the reference's test shape (d=5, c=3) run here too.
//
// One wave owns R tiles of 16 rows for the whole L-layer stack.  Every contraction is a
// v_mfma_f32_16x16x4_f32 computed TRANSPOSED (features on the MFMA M axis, rows on N), so that
// lane (q = lane >> 4, r = lane & 15) always holds data of row r:
//   * x:  lane (q, r) keeps features j = q*2NF + e, e in [0, 2NF)  (d = 8*NF; a float4 row load),
//   * c:  lane (q, r) keeps conditions q*CQ + g, g in [0, CQ)       (cdim = 4*CQ),
//   * GEMM1  hidden^T[hid x rows] = W1 . in^T : B operand = those registers directly
//            (k-step kk: k = q <-> conditioning slot kk of lane group q, or condition kk - NF),
//   * its accumulator (lane (q, r): hid 16*ht + 4q + reg, row r) IS the B operand of
//     GEMM2  out^T[out x rows] = W2 . h^T  with the k order permuted (k-step reg: k = q),
//   * GEMM2's accumulator rows are ordered so lane (q, r) receives t and s of exactly the
//     features it keeps, and the affine update is purely per-lane.
// Activations therefore never leave registers between the two Linears and across layers; only the
// weights (pre-packed into these fragment orders by k_pack_weights) stream in from L2.
//
// With the reference's alternating masks (realnvp.py:199; rnvp_shape::alt_masks = 1) layer l
// conditions on features of parity pc = (l + alt_masks) & 1 and transforms the others; masked-out columns of W1 and rows of W2
// (dead work in the reference's dense form) are never computed.
#pragma once
#include "rnvp_common.h"
#include "rnvp_split.h"

namespace rnvp {
namespace mfma {

struct Geo {
    int d, c, h;  // the caller's real sizes; the tile geometry below is padded up from them (padded
                  // features / conditions / hidden units carry zero weights and never reach the output)
    int NF;     // padded d / 8: features per parity class per lane
    int CQ;     // padded c / 4: conditions per lane
    int HT;     // padded h / 16: hidden tiles per net
    int KS1;    // NF + CQ: k-steps of GEMM1
    int K4;     // ceil(KS1 / 4): float4 groups of A1 per lane
    int OTL;    // out tiles fed by one hidden tile: max(1, NF / 4)
    int NT2;    // out tiles in total: NF >= 4 ? 2 * OTL : 1 (t and s share one tile when NF == 2)
    int MTI;    // M tiles of the input gradient: max(1, NF / 4)
    // float offsets inside one layer's packed block
    int oA1, oB1, oA2, oB2, oA2T, oA1T, oA2X, oA1X, layer_floats;
    // split-bf16 fragments of the training kernel's GEMM1 (forward and recompute) and of g_h = W2^T g_out (rnvp_split.h):
    // present only when make_geo was asked for them (split = true), else NI1 = NI2 = 0 and both offsets = layer_floats
    int NI1;    // v_mfma_f32_16x16x32_bf16 per GEMM1 tile: ceil(3 KS1 / 4)
    int NI2;    // the same for g_h (NF values per lane): ceil(3 NF / 4)
    int oA1S;   // [tile][NI1][lane][4 dwords]
    int oA2TS;  // [tile][NI2][lane][4 dwords]
};

// smallest instantiated (NF, CQ) that holds d features and c conditions: (2,0) (2,1) (4,2) (8,4)
__host__ __device__ inline bool pick_tiles(int d, int c, int *NF, int *CQ) {
    if (d <= 16 && c == 0) { *NF = 2; *CQ = 0; return true; }
    if (d <= 16 && c <= 4) { *NF = 2; *CQ = 1; return true; }
    if (d <= 32 && c <= 8) { *NF = 4; *CQ = 2; return true; }
    if (d <= 64 && c <= 16) { *NF = 8; *CQ = 4; return true; }
    return false;
}

__host__ __device__ inline Geo make_geo(int d, int c, int h, bool split = false) {
    Geo g;
    g.d = d; g.c = c; g.h = h;
    g.NF = 2; g.CQ = 0;
    pick_tiles(d, c, &g.NF, &g.CQ);
    g.HT = (h + 15) / 16;
    g.KS1 = g.NF + g.CQ; g.K4 = (g.KS1 + 3) / 4;
    g.OTL = g.NF >= 4 ? g.NF / 4 : 1;
    g.NT2 = g.NF >= 4 ? 2 * g.OTL : 1;
    g.MTI = g.NF >= 4 ? g.NF / 4 : 1;
    int o = 0;
    g.oA1 = o; o += 2 * g.HT * g.K4 * 256;          // [tile][k4][lane][4]
    g.oB1 = o; o += 2 * g.HT * 16;                  // [tile][q][4]
    g.oA2 = o; o += 2 * g.HT * g.OTL * 256;         // [tile][otl][lane][4 (rho)]
    g.oB2 = o; o += g.NT2 * 16;                     // [ot][q][4]
    g.oA2T = o; o += 2 * g.HT * g.OTL * 256;        // [tile][otl][lane][4 (rho)]   (backward: W2^T)
    g.oA1T = o; o += 2 * g.HT * g.MTI * 256;        // [tile][mt][lane][4 (rho)]    (backward: W1^T)
    // d == 16 only: per-lane fragments of the 4x4x1 (16-block) MFMA forms of GEMM2 and of the
    // input-gradient product, which avoid the structural zeros of the shared t|s out tile
    g.oA2X = o; o += (g.NF == 2) ? 2 * g.HT * 2 * 256 : 0;   // [tile][og][lane][4 (rho)]
    g.oA1X = o; o += (g.NF == 2) ? 2 * g.HT * 2 * 256 : 0;   // [tile][og][lane][4 (rho)]
    g.NI1 = split ? split::n_mfma(g.KS1) : 0;
    g.NI2 = split ? split::n_mfma(g.NF) : 0;
    g.oA1S = o; o += 2 * g.HT * g.NI1 * 256;
    g.oA2TS = o; o += 2 * g.HT * g.NI2 * 256;
    g.layer_floats = o;
    return g;
}

// feature index of conditioning / transformed slot f of lane group q in a layer of parity pc
__host__ __device__ inline int feat_cond(int NF, int q, int f, int pc) { return q * 2 * NF + 2 * f + pc; }
__host__ __device__ inline int feat_trans(int NF, int q, int f, int pc) { return q * 2 * NF + 2 * f + 1 - pc; }

bool supported(const KShape &k);            // forward / inverse
bool train_supported(const KShape &k);      // fused forward + backward
bool backward_rows_ok(const KShape &k, int64_t n);      // rnvp_backward (per-row seeds, d loss / d x) of n rows runs here
size_t train_workspace_bytes(const KShape &k, int64_t max_rows);
// err (nullable): the training step's error word, cleared by the same launch
int pack_weights(hipStream_t st, const KShape &k, const Geo &g, const float *params, float *packed, int *err = nullptr);
size_t workspace_bytes(const KShape &k, int op, int64_t max_rows);
int forward(hipStream_t st, const KShape &k, const float *params, const float *x, const float *c,
            const int64_t *row_index, int64_t n, float *z_out, float *logdet_out, float *logp_out,
            float *logp_sum, void *ws, size_t ws_bytes);
int inverse(hipStream_t st, const KShape &k, const float *params, const float *z, const float *c,
            int64_t n, float *x_out, void *ws, size_t ws_bytes);
int sample(hipStream_t st, const KShape &k, const float *params, const float *c, int64_t n, uint64_t seed,
           int64_t row0, float *x_out, void *ws, size_t ws_bytes);
int loss_grad(hipStream_t st, const KShape &k, const float *params, const float *x, const float *c,
              const int64_t *row_index, int64_t n, float inv_B, float *grad_out, float *loss_out,
              void *ws, size_t ws_bytes, Seeds sd = Seeds{}, bool packed_valid = false);

// packed_valid: the workspace already holds the packed fragments of `params` (an earlier step of the same rnvp_fit_epoch*
// call left them there): no pack launch.  pack_next: the step's finish kernel re-packs the parameters it updates.
int train_step(hipStream_t st, const KShape &k, float *params, const float *x, const float *c,
               const int64_t *row_index, int64_t n, float inv_B, float *grad_buf, float *loss_out,
               float *exp_avg, float *exp_avg_sq, const AdamK &adam, void *ws, size_t ws_bytes, bool packed_valid = false,
               bool pack_next = false);
// data parallel, after the all-reduce: loss read-out + Adam from the flat gradient + re-pack, one launch
// the data-parallel step in chunks of layers (rnvp_dp.hip, rnvp_dp_set_chunks): training launch alone, then per chunk the partial
// sums, [the caller's all-reduce of the chunk,] Adam + re-pack of the chunk's layers
struct PendingPartials { int glayer_floats, w2c, grid; float inv_B; };
int loss_partials(hipStream_t st, const KShape &k, const float *params, const float *x, const float *c, const int64_t *row_index,
                  int64_t n, float inv_B, void *ws, size_t ws_bytes, bool packed_valid, PendingPartials *pending);
int finish_sum_layers(hipStream_t st, const KShape &k, const PendingPartials &p, int l0, int nl, float *grad, float *loss_out,
                      void *ws, size_t ws_bytes);
int adam_pack_layers(hipStream_t st, const KShape &k, float *params, float *grad, const float *loss_in, float *loss_out,
                     float *exp_avg, float *exp_avg_sq, const AdamK &adam, void *ws, size_t ws_bytes, int l0, int nl);
int adam_pack(hipStream_t st, const KShape &k, float *params, float *grad, const float *loss_in, float *loss_out,
              float *exp_avg, float *exp_avg_sq, const AdamK &adam, void *ws, size_t ws_bytes);

}  // namespace mfma
}  // namespace rnvp
