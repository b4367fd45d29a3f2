// rnvp_resident.h -- one epoch of a small flow at a small batch size in one persistent workgroup (rnvp_resident.hip)
#pragma once
#include "rnvp_common.h"

namespace rnvp {
namespace resident {

// the model (and the batch's activations) fit one CU's LDS: parameters + per-wave gradient stages + per-wave images
bool fits(const KShape &k, int64_t batch_size);
// perm: n_epochs permutations of n rows back to back; loss_hist: n_epochs x ceil(n / batch_size) batch losses
int fit_epoch(hipStream_t st, const KShape &k, float *params, const uint8_t *masks, const float *x, const float *c,
              const int64_t *perm, int64_t n, int64_t batch_size, int64_t n_epochs, float *loss_hist, float *exp_avg, float *exp_avg_sq,
              double lr, double beta1, double beta2, double eps, double weight_decay, int64_t first_step);

// the conditional VAE's batch loop (cvae.py:235-252) the same way: one hidden layer per MLP, latent <= 8
bool cvae_fits(const CvaeK &k, int family, int64_t batch_size);
int cvae_fit_epoch(hipStream_t st, const CvaeK &k, float *params, const float *x, const float *c, const int64_t *perm,
                   const float *eps, int64_t n, int64_t batch_size, float kl_weight, float *loss_hist, float *exp_avg,
                   float *exp_avg_sq, double lr, double beta1, double beta2, double adam_eps, double weight_decay, int64_t first_step);

}  // namespace resident
}  // namespace rnvp
