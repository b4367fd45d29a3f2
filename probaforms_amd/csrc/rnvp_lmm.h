// rnvp_lmm.h -- geometry and entry points of the "lmm" kernels (rnvp_lmm.hip): any-shape coupling stack on f32 MFMA
// with wave-private LDS-resident activations.
#pragma once
#include "rnvp_common.h"

namespace rnvp {
namespace lmm {

struct LGeo {
    int nlin;                              // Linears per net: n_hidden + 1
    int nnets;                             // nets that share this geometry: 2 L (t, s per coupling layer); 1 for a CVAE encoder / decoder
    int nin[kMaxLin], nout[kMaxLin];
    int MT[kMaxLin], KS[kMaxLin];          // forward: out tiles (16), k-steps (4) over the inputs
    int MTt[kMaxLin], KSt[kMaxLin];        // transposed (input gradient): in tiles, k-steps over the outputs
    int PT[kMaxLin];                       // weight gradient: in tiles incl. the ones (bias) column
    int offF[kMaxLin], offT[kMaxLin];      // float offsets of the fragment blocks inside one net's packed image
    int offG[kMaxLin];                     // ... of Linear k's [m][p][64 lanes][4] block inside one net's partial gradient
    int offA[kMaxLin], offP[kMaxLin];      // ... of the dumped input / pre-activation-gradient tiles of one (row tile, net)
    int net_floats, gnet_floats, dump_floats, pairs_per_net, quads_per_net;
    int hs, hmax, wmax;
    size_t lds_flow, lds_train;            // bytes of one wave's LDS image
};

LGeo make_lgeo(const KShape &k);
bool use_lmm(const KShape &k, int op);     // policy (rnvp_shape::family: VALU never, otherwise whenever the LDS image fits)
size_t workspace_bytes(const KShape &k, int op, int64_t max_rows);
int forward(hipStream_t st, const KShape &k, const float *params, const uint8_t *masks, const float *x, const float *c,
            const int64_t *row_index, int64_t n, float *z_out, float *logdet_out, float *logp_out, float *logp_sum,
            void *ws, size_t ws_bytes);
int inverse(hipStream_t st, const KShape &k, const float *params, const uint8_t *masks, const float *z, const float *c,
            int64_t n, float *x_out, void *ws, size_t ws_bytes);
int loss_grad(hipStream_t st, const KShape &k, const float *params, const uint8_t *masks, const float *x, const float *c,
              const int64_t *row_index, int64_t n, float inv_B, float *grad_out, float *loss_out, void *ws, size_t ws_bytes,
              Seeds sd);
// rnvp_lmm64.hip: the training call on 64-row blocks with the weight gradients inside the kernel (loss_grad dispatches to it)
bool use_train64(const KShape &k, int64_t n);
size_t train64_workspace_bytes(const KShape &k, int64_t max_rows);
int loss_grad64(hipStream_t st, const KShape &k, const float *params, const uint8_t *masks, const float *x, const float *c,
                const int64_t *row_index, int64_t n, float inv_B, float *grad_out, float *loss_out, void *ws, size_t ws_bytes,
                Seeds sd);

// ---- CVAE (encoder / decoder MLPs of any depth and width) on the same building blocks ---------------------------------
// op: RNVP_OP_TRAIN (cvae_loss_grad), RNVP_OP_FORWARD (cvae_encode), RNVP_OP_INVERSE (cvae_decode)
bool cvae_fits(const CvaeK &k, int op);
size_t cvae_workspace_bytes(const CvaeK &k, int64_t max_rows);
int cvae_loss_grad(hipStream_t st, const CvaeK &k, const float *params, const float *x, const float *c, const int64_t *row_index,
                   const float *eps, int64_t n, float inv_B, float klw, float *grad_out, float *loss_out, void *ws, size_t ws_bytes);
int cvae_forward(hipStream_t st, const CvaeK &k, const float *params, bool encode, const float *in, const float *c, int64_t n,
                 float *out0, float *out1, void *ws, size_t ws_bytes);

}  // namespace lmm
}  // namespace rnvp
