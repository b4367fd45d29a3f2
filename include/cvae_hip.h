/*
 * cvae_hip.h -- C ABI of the conditional-VAE entry points of librnvp_hip.so (SURVEY.md 8(f) rank 1,
 * BASELINE.json configs[4]).  Same conventions as rnvp_hip.h (device pointers, caller-owned
 * workspace, caller's stream, int status).  Replaces the arithmetic of
 * /root/reference/probaforms/models/cvae.py: Encoder.forward (39-64), Decoder.forward (92-113),
 * CVAE.sample_z / custom_loss / compute_loss (186-203) and the autograd backward of
 * `loss.backward()` (243-246).  The optimizer step is rnvp_adam_step (rnvp_hip.h).
 *
 * params: flat float32: encoder hidden Linears (weight [out,in], bias) in order; then the two heads
 *   stored as ONE Linear: mu.weight, log_sigma.weight, mu.bias, log_sigma.bias; then the decoder's
 *   Linears (weight, bias) in order.  (Each tensor is a contiguous [out,in] block, so the
 *   reference's state_dict tensors can simply be views into this buffer.)
 * x [n,d], c [n,cdim] (NULL iff cdim == 0), eps [n_rows, lat] (the N(0,1) draws of sample_z,
 *   cvae.py:187, in BATCH order), z [n, lat]; all float32 row-major.
 */
#ifndef CVAE_HIP_H
#define CVAE_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct cvae_shape {
    int32_t d;            /* var_size                                   cvae.py:170-173 */
    int32_t c;            /* cond_size, 0 for C=None                    cvae.py:159-162 */
    int32_t lat;          /* latent_dim                                 cvae.py:145     */
    int32_t n_hidden;     /* len(hidden)                                              */
    int32_t hidden[8];
    int32_t act;          /* 0 tanh, 1 relu                             cvae.py:26-32   */
    int32_t family;       /* RNVP_FAMILY_* (rnvp_hip.h), per call.  0 AUTO: the register-chained MFMA kernels where the
                             shape allows, else the any-shape MFMA kernels (RNVP_PATH_LMM) while a 16-row tile's LDS
                             image fits, else one thread per row; 1 VALU: always one thread per row; 2 LMM: the
                             any-shape MFMA kernels also where AUTO would pick the register-chained ones (test /
                             measurement aid: every path is checked against the oracle)                          */
} cvae_shape;

size_t cvae_param_count(const cvae_shape *shape);
size_t cvae_workspace_bytes(const cvae_shape *shape, int64_t max_rows);

/* RNVP_PATH_MFMA (d <= 16, c <= 4, lat <= 4, one tanh hidden layer, shape->family == 0), RNVP_PATH_LMM (any depth, width,
 * activation, d, c, lat whose tile image fits the 160 KB of LDS) or RNVP_PATH_GENERIC: which kernels cvae_loss_grad runs. */
int cvae_kernel_path(const cvae_shape *shape);

/* loss = KL_weight * (1/B) sum_b KL_b + (1/(B d)) sum_{b,j} (x - x_rec)^2 with 1/B = inv_B, and its
 * gradient wrt every parameter (grad_out [P]; NULL = loss only, the per-epoch evaluation of
 * cvae.py:254-259).  row_index (nullable) gathers x/c rows; eps is already in batch order. */
int cvae_loss_grad(void *stream, const cvae_shape *shape, const float *params,
                   const float *x, const float *c, const int64_t *row_index, const float *eps,
                   int64_t n_rows, float inv_B, float kl_weight,
                   float *grad_out, float *loss_out, void *workspace, size_t workspace_bytes);

/* cvae_loss_grad followed by rnvp_adam_step (rnvp_hip.h) on the same stream: the three lines `loss = compute_loss(...)`,
 * `loss.backward()`, `opt.step()` of the training loop (cvae.py:243-246).  On the MFMA path the optimizer is fused into the
 * kernel that scatters the gradient into parameter order (one launch fewer; identical arithmetic).  grad_buf [P] is scratch
 * owned by the caller; `step` is the 1-based Adam step. */
int cvae_train_step(void *stream, const cvae_shape *shape, float *params,
                    const float *x, const float *c, const int64_t *row_index, const float *eps,
                    int64_t n_rows, float inv_B, float kl_weight,
                    float *grad_buf, float *loss_out, float *exp_avg, float *exp_avg_sq,
                    double lr, double beta1, double beta2, double adam_eps, double weight_decay, int64_t step,
                    void *workspace, size_t workspace_bytes);

/*
 * One epoch of the batch loop of CVAE.fit (cvae.py:235-252) in one call: for every consecutive slice of `batch_size`
 * entries of `perm` (the epoch's DataLoader permutation, int64 [n] on the device; the last slice may be ragged) run
 * cvae_train_step with inv_B = 1 / rows_in_slice on eps[s0 .. s0 + rows) (eps [n, latent] holds the epoch's sample_z draws in
 * batch order) and store that batch's loss in loss_hist[k] ([ceil(n / batch_size)], required).  `first_step` is the 1-based
 * Adam step of the first batch; grad_buf [P] and the workspace as for cvae_train_step at batch_size rows.
 * Small models -- one hidden layer of at most 16 units (32 while the widest input has at most 15 columns), d <= 16,
 * latent <= 8, batch_size <= 128: the reference's defaults hidden=(10,), latent 2, batch 32 (cvae.py:145) -- run as ONE
 * persistent launch per epoch with parameters, Adam state and gradient stages resident in one CU's LDS
 * (rnvp_resident.hip; cvae_fit_epoch_resident says whether).  That form agrees with the batch-by-batch loop to rounding
 * (another summation order) and reproduces itself bit for bit; family == 1 pins the loop.
 */
int cvae_fit_epoch(void *stream, const cvae_shape *shape, float *params,
                   const float *x, const float *c, const int64_t *perm, const float *eps,
                   int64_t n, int64_t batch_size, float kl_weight,
                   float *grad_buf, float *loss_hist, float *exp_avg, float *exp_avg_sq,
                   double lr, double beta1, double beta2, double adam_eps, double weight_decay, int64_t first_step,
                   void *workspace, size_t workspace_bytes);
int cvae_fit_epoch_resident(const cvae_shape *shape, int64_t batch_size);

/* x_out [n,d] = Decoder([z || c])                                       cvae.py:108-113, 284-290
 * workspace (nullable; cvae_workspace_bytes) lets shapes on RNVP_PATH_MFMA run the MFMA kernels (the packed
 * weights live there); NULL runs the generic kernels. */
int cvae_decode(void *stream, const cvae_shape *shape, const float *params,
                const float *z, const float *c, int64_t n_rows, float *x_out,
                void *workspace, size_t workspace_bytes);

/* mu [n,lat], log_sigma [n,lat] = Encoder([x || c])                     cvae.py:57-64; workspace as above */
int cvae_encode(void *stream, const cvae_shape *shape, const float *params,
                const float *x, const float *c, int64_t n_rows, float *mu_out, float *log_sigma_out,
                void *workspace, size_t workspace_bytes);

#ifdef __cplusplus
}
#endif
#endif
