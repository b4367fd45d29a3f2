// rnvp_mfma_train_nf2.hip -- the training kernels of rnvp_mfma_train_dev.h instantiated for the tile geometry (NF 2, CQ 1): d <= 16 with conditions
// (one translation unit per geometry so that they compile in parallel; the host side is rnvp_mfma_train.hip).
#include "rnvp_mfma_train_dev.h"

namespace rnvp {
namespace mfma {

int launch_train_2_1(hipStream_t st, const KShape &k, const Geo &g, const TrainPlan &pl, const float *packed, const float *x,
       const float *c, const int64_t *row_index, int64_t n, float inv_B, float *gpart, float *losspart, float *scratch,
       int *grid_out, Seeds sd, PartialLayout *lay) {
    return launch_train<2, 1>(st, k, g, pl, packed, x, c, row_index, n, inv_B, gpart, losspart, scratch, grid_out, sd, lay);
}

}  // namespace mfma
}  // namespace rnvp
