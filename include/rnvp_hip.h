/*
 * rnvp_hip.h -- C ABI of librnvp_hip.so: the MI355X (gfx950) implementation of the
 * conditional RealNVP hot path of hse-cs/probaforms.
 *
 * The reference has no FFI: its replaceable seam is the Python `InvertibleLayer`
 * protocol (/root/reference/probaforms/models/nflow.py:30-67) consumed by
 * `NormalizingFlow.log_prob/sample` (nflow.py:90-145) and `RealNVP.fit/sample`
 * (/root/reference/probaforms/models/realnvp.py:210-282).  Each entry point below
 * names the reference code it replaces.  INTEGRATION.md shows the ctypes binding a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer except `shape` is a DEVICE pointer;
 *   - the caller owns all device memory including the workspace (no hidden hipMalloc);
 *   - kernels are enqueued on `stream` (a hipStream_t passed as void*) and the call
 *     returns without synchronising; safe to capture into a hipGraph;
 *   - return value: 0 ok; <0 argument error (RNVP_E*); >0 a hipError_t;
 *   - no global mutable state other than one-time kernel attribute setup (per device) and the
 *     mutex-guarded event table of rnvp_profile_* => thread-safe per stream.
 *
 * Data layout
 *   x, z : [n, d] float32 row-major;  c : [n, cdim] float32 row-major (NULL iff cdim==0)
 *   masks: [L, d] uint8 in {0,1};  1 = pass-through and fed to the s/t nets (realnvp.py:92),
 *          0 = transformed.  The reference builds (j + l) % 2 (realnvp.py:199).
 *   params: flat float32 in `nf.parameters()` order (realnvp.py:69-70,196-204): for each
 *          layer: net t then net s; per net, for each Linear: weight [out,in] row-major, bias.
 *   row_index: optional int64 [n_rows]; row r of the batch is x[row_index[r]] (the
 *          DataLoader shuffle of realnvp.py:237 fused into the row loads). NULL = identity.
 */
#ifndef RNVP_HIP_H
#define RNVP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RNVP_MAX_HIDDEN 8

#define RNVP_OK            0
#define RNVP_EINVAL       (-1)   /* bad shape / NULL pointer / negative size            */
#define RNVP_EUNSUPPORTED (-2)   /* shape exceeds what the kernels can hold in LDS      */
#define RNVP_EWORKSPACE   (-3)   /* workspace smaller than rnvp_workspace_bytes() says  */

#define RNVP_ACT_TANH 0
#define RNVP_ACT_RELU 1          /* anything but 'tanh' is ReLU, realnvp.py:32-37       */

/* which kernel family serves a call (rnvp_kernel_path); informational */
#define RNVP_PATH_GENERIC 0      /* any shape, VALU + LDS, one thread per row           */
#define RNVP_PATH_MFMA    1      /* register-chained MFMA path (one hidden layer, alternating masks, d <= 64, cdim <= 16) */
#define RNVP_PATH_LMM     2      /* any shape (several hidden layers, user masks, wide rows) on f32 MFMA with
                                    LDS-resident activations; taken whenever a 16-row tile's LDS image fits (<= 76 KB) */

typedef struct rnvp_shape {
    int32_t L;                        /* n_layers                  realnvp.py:160,196   */
    int32_t d;                        /* var_size                  realnvp.py:182       */
    int32_t c;                        /* cond_size, 0 for C=None   realnvp.py:183-186   */
    int32_t n_hidden;                 /* len(hidden)               realnvp.py:22        */
    int32_t hidden[RNVP_MAX_HIDDEN];
    int32_t act;                      /* RNVP_ACT_*                                     */
    int32_t alt_masks;                /* caller's declaration about `masks`:
                                         0 = arbitrary {0,1} table, read from the masks pointer;
                                         1 = masks[l][j] == (j + l) % 2, the reference's own
                                             construction (realnvp.py:199);
                                         2 = masks[l][j] == (j + l + 1) % 2 (the same pattern seen
                                             from an odd layer, e.g. one RealNVPLayer on its own).
                                         1/2 let the library skip the masked-out (dead) columns and
                                         pick the MFMA path; the masks pointer is then not read.   */
    int32_t precision;                /* RNVP_PREC_*: arithmetic of the first Linear of the s/t nets in the forward /
                                         inverse / sampling kernels of the MFMA path and in the FORWARD phase of its
                                         training kernels (row-parallel launches, d <= 32; the backward is always f32).
                                         The reference computes in float32 throughout, realnvp.py:226-228; both
                                         settings meet its 1e-5 bar and the same gradient tolerances */
    int32_t small_calls;              /* RNVP_SMALL_*: how forward / inverse / sampling calls of at most 4096 rows run */
    int32_t family;                   /* RNVP_FAMILY_*: which kernels serve a shape OUTSIDE RNVP_PATH_MFMA (several hidden
                                         layers, user masks, d > 64, cdim > 16).  Per call, not process state: two threads
                                         may run different families at once.  Test / measurement aid; 0 in production */
} rnvp_shape;

#define RNVP_FAMILY_AUTO 0       /* RNVP_PATH_LMM whenever a 16-row tile's LDS image fits, else RNVP_PATH_GENERIC          */
#define RNVP_FAMILY_VALU 1       /* always RNVP_PATH_GENERIC (one thread per row, VALU + LDS)                               */
#define RNVP_FAMILY_LMM  2       /* as AUTO (kept distinct so that a test can say what it asks for)                          */
#define RNVP_FAMILY_LMM16 3      /* RNVP_PATH_LMM with the training call pinned to the 16-rows-per-workgroup form (k_lmm_train)   */
#define RNVP_FAMILY_LMM64 4      /* ... to the 64-rows-per-workgroup form with in-kernel weight gradients (k_lmm_train64)
                                    wherever its LDS image and register slots fit; AUTO picks it from 8192 rows per call on   */

#define RNVP_SMALL_INVARIANT 0   /* default: a row's result never depends on how the rows are split into calls (chunks
                                    of a pipelined draw, shards of ranks and the one-shot call agree bit for bit)      */
#define RNVP_SMALL_LATENCY   1   /* calls of at most 4096 rows run the tile-split kernels (a workgroup's waves share 16 rows
                                    and split the hidden tiles): 2-4x lower latency; the hidden tiles are summed in
                                    another order, so results differ from larger calls in the last bits              */

#define RNVP_PREC_AUTO 0         /* the faster of the two for the shape: BX3 for d > 16 or cdim > 4 (measured 1.3-1.4x on every
                                    operation); for d <= 16 with one hidden layer of more than 64 units BX3 too (training kernel's
                                    forward phase 3-5 %, flow kernels 5-9 % with warm clocks: barrier-free form); F32 else */
#define RNVP_PREC_F32  1         /* f32-input MFMA (v_mfma_f32_16x16x4_f32): bitwise an fmaf chain          */
#define RNVP_PREC_BX3  2         /* operands split into three bf16 terms, six bf16 MFMA products per pair:
                                    float32-level accuracy (every dropped product < 2^-24), weights staged in LDS */

/* operations, for rnvp_workspace_bytes */
#define RNVP_OP_FORWARD  0
#define RNVP_OP_INVERSE  1
#define RNVP_OP_TRAIN    2

#define RNVP_HIP_VERSION 600     /* rnvp_version(): bumped whenever a struct layout or an argument list changes; a
                                    binding must refuse a library that reports another number                      */
int         rnvp_version(void);
const char *rnvp_status_string(int status);

/* number of float32 parameters of the whole flow (2 nets x L layers). */
size_t rnvp_param_count(const rnvp_shape *shape);

/* bytes of device workspace `op` needs for up to max_rows rows in one call. */
size_t rnvp_workspace_bytes(const rnvp_shape *shape, int op, int64_t max_rows);

/* RNVP_PATH_* the library will use for this shape/masks (masks is a HOST pointer here,
 * may be NULL meaning the default alternating masks). */
int rnvp_kernel_path(const rnvp_shape *shape, const uint8_t *host_masks, int op);

/*
 * Forward transform + log-det + prior log-prob.
 * Replaces RealNVPLayer.f (realnvp.py:73-101) for every layer and
 * NormalizingFlow.log_prob (nflow.py:107-117).  With shape->L == 1 and `params`/`masks`
 * pointing at one layer it is exactly RealNVPLayer.f: z_out = X_new, logdet_out = log_det.
 *   z_out      [n_rows, d]  transformed rows              (nullable)
 *   logdet_out [n_rows]     sum over layers of log_det    (nullable)
 *   logp_out   [n_rows]     logdet + N(0,I).log_prob(z)   (nullable)   nflow.py:115
 *   logp_sum   [1]          sum over rows of logp_out     (nullable; overwritten)
 */
int rnvp_forward_logprob(void *stream, const rnvp_shape *shape,
                         const float *params, const uint8_t *masks,
                         const float *x, const float *c, const int64_t *row_index,
                         int64_t n_rows,
                         float *z_out, float *logdet_out, float *logp_out, float *logp_sum,
                         void *workspace, size_t workspace_bytes);

/*
 * Inverse transform (sampling): layers in reverse order.
 * Replaces RealNVPLayer.g (realnvp.py:104-129) and the loop of NormalizingFlow.sample
 * (nflow.py:142-143); the caller supplies z (prior draw, nflow.py:141).  x_out may alias z.
 */
int rnvp_inverse(void *stream, const rnvp_shape *shape,
                 const float *params, const uint8_t *masks,
                 const float *z, const float *c, int64_t n_rows, float *x_out,
                 void *workspace, size_t workspace_bytes);

/*
 * Prior draw of NormalizingFlow.sample (nflow.py:141, `X = self.prior.sample((n,))`; the prior is
 * MultivariateNormal(0, I), realnvp.py:189-191) as a counter-based stream: z_out[r][j] is a pure function
 * of (seed, row_offset + r, j) -- Philox4x32-10 keyed by `seed`, counter (row, j / 4), Box-Muller -- so any
 * split of the rows over calls, chunks or ranks (each passing its first global row as row_offset)
 * reproduces the single-call draw bit for bit.  This is the build's 'device' prior; it is NOT the
 * reference's CPU generator stream (the host supplies that one as `z` to rnvp_inverse).
 *   z_out [n_rows, d]
 */
int rnvp_prior_normal(void *stream, uint64_t seed, int64_t row_offset, int64_t n_rows, int32_t d,
                      float *z_out);

/*
 * The REFERENCE's prior stream drawn on the device: exactly the values `torch.randn(count)` produces from a CPU generator
 * whose Mersenne-Twister state is `mt_state` (nflow.py:141: prior.sample = randn on the global CPU generator).  mt_state
 * [625] (device): the generator's 624 state words + the position of the next unread word (624 = block used up); advanced in
 * place as the host draw would advance it, so the caller can hand it back to its generator.  z_out [count], count >= 16
 * (torch draws smaller tensors another way: RNVP_EUNSUPPORTED); tail16 [16]: device scratch.  Restates torch's CPU kernel for
 * contiguous float tensors (mt19937, 24-bit uniforms, 16-element Box-Muller blocks on the cephes polynomials of avx_mathfun.h
 * with the multiply-adds torch's build contracts); the Python host checks it against torch.randn itself once per process and
 * keeps the host draw if another torch build disagrees.  The twister is serial from one 624-word block to the next; up to 32
 * workgroups nevertheless share ONE stream: segment k starts from the state k * B blocks ahead (B = 64, 256 or 1024 by the length of the draw), obtained as a binary convolution
 * of the next 33 blocks with a precomputed jump polynomial x^J mod phi (csrc/rnvp_mt19937_jump.h, scripts/mt19937_jump_poly.py);
 * a second kernel tempers the words and applies the Box-Muller blocks.  workspace: rnvp_prior_torch_workspace_bytes() bytes.
 */
size_t rnvp_prior_torch_workspace_bytes(void);
int rnvp_prior_normal_torch_cpu(void *stream, uint32_t *mt_state, int64_t count, float *z_out, float *tail16,
                                void *workspace, size_t workspace_bytes);

/*
 * The REFERENCE's epoch shuffle drawn on the device: exactly `torch.randperm(n, generator=g)` of a CPU generator whose
 * Mersenne-Twister state is `mt_state` [625] (as above; advanced by the n - 1 draws the host call makes).  realnvp.py:235's
 * DataLoader(shuffle=True) calls it once per epoch on a private generator seeded from the global one (RandomSampler); on the host
 * it is a serial Fisher-Yates pass (8.5 ms per million rows).  Here the swap targets come from the device twister and the swaps
 * run in parallel rounds with deterministic reservations (csrc/rnvp_randperm.hip): the result is the sequential permutation bit
 * for bit.  perm_out [n] int64; 1 <= n < 2^32 / 20 (torch shuffles larger n another way: RNVP_EUNSUPPORTED).  The Python host
 * checks it against torch.randperm once per process and keeps the host shuffle if another torch build disagrees.
 */
size_t rnvp_randperm_workspace_bytes(int64_t n);
int rnvp_randperm_torch_cpu(void *stream, uint32_t *mt_state, int64_t n, int64_t *perm_out, void *workspace,
                            size_t workspace_bytes);

/*
 * rnvp_prior_normal fused into rnvp_inverse: x_out = g(z(seed, row_offset + r, .), c[r]) for the n_rows
 * rows of this call; on the MFMA path z is drawn in registers and never written to memory.
 * Replaces nflow.py:141-143 as a whole.  Same workspace as RNVP_OP_INVERSE.
 */
int rnvp_sample(void *stream, const rnvp_shape *shape,
                const float *params, const uint8_t *masks, const float *c,
                int64_t n_rows, uint64_t seed, int64_t row_offset, float *x_out,
                void *workspace, size_t workspace_bytes);

/*
 * loss = -(sum_rows logp) * inv_B and d(loss)/d(params) for one (shard of a) batch.
 * Replaces `loss = -nf.log_prob(X, C); opt.zero_grad(); loss.backward()`
 * (realnvp.py:246-250).  inv_B = 1 / (global batch size) so that shards from several GPUs
 * add up to the reference's batch mean.  grad_out [P] and loss_out [1] are OVERWRITTEN
 * (n_rows == 0 writes zeros).  Deterministic: no float atomics.
 */
int rnvp_loss_grad(void *stream, const rnvp_shape *shape,
                   const float *params, const uint8_t *masks,
                   const float *x, const float *c, const int64_t *row_index,
                   int64_t n_rows, float inv_B,
                   float *grad_out, float *loss_out,
                   void *workspace, size_t workspace_bytes);

/*
 * rnvp_loss_grad for a flow whose prior is NOT N(0, I): the reference trains whatever object the user put in
 * `self.prior` (realnvp.py:189 builds the default only `if self.prior is None`; nflow.py:115 calls its log_prob).
 * The caller evaluates the prior on z = f(x) (rnvp_forward_logprob) and passes
 *   gz [n_rows, d] = d loss / d z = -inv_B * d prior.log_prob(z) / d z     (row r of the BATCH, not row_index[r]);
 * this call seeds the hand-derived backward with it.  loss_out[0] = -(sum_rows log_det) * inv_B only: the caller
 * adds -(sum_rows prior.log_prob(z)) * inv_B.  grad_out as rnvp_loss_grad.
 */
int rnvp_loss_grad_zseed(void *stream, const rnvp_shape *shape,
                         const float *params, const uint8_t *masks,
                         const float *x, const float *c, const int64_t *row_index,
                         int64_t n_rows, float inv_B, const float *gz,
                         float *grad_out, float *loss_out,
                         void *workspace, size_t workspace_bytes);

/*
 * Backward of rnvp_forward_logprob for ANY scalar loss of its outputs: the vector-Jacobian product autograd forms in the
 * reference when a user differentiates through `layer.f(X, C)` / `nf.log_prob(X, C)` (nflow.py:107-117; the reference's
 * own fit does exactly this at realnvp.py:246-250).  Given, per batch row r,
 *   gz  [n_rows, d] = d loss / d z[r]        gld [n_rows] = d loss / d logdet[r]
 * it returns
 *   grad_out [rnvp_param_count]  d loss / d params        gx_out [n_rows, d]  d loss / d x[r]   (nullable)
 * (the gradient w.r.t. the conditions is not formed here: rnvp_backward_cond below).  Rows are batch rows:
 * with row_index the inputs are gathered, gz / gld / gx_out are not.  With shape->L == 1 this is the backward of one
 * RealNVPLayer.f.  Same kernels, workspace (RNVP_OP_TRAIN) and determinism as rnvp_loss_grad.
 */
int rnvp_backward(void *stream, const rnvp_shape *shape,
                  const float *params, const uint8_t *masks,
                  const float *x, const float *c, const int64_t *row_index,
                  int64_t n_rows, const float *gz, const float *gld,
                  float *grad_out, float *gx_out,
                  void *workspace, size_t workspace_bytes);

/*
 * rnvp_backward that ALSO forms the gradient with respect to the conditions.  In the reference C enters every layer through
 * `torch.cat((X * mask, C), dim=1)` (realnvp.py:92) and receives a gradient whenever it requires one -- a learned condition
 * encoder in a user's own training loop.
 *   gc_out [n_rows, c]  d loss / d c[r]   (nullable; ignored when shape->c == 0)       the other arguments as rnvp_backward.
 * Served by the any-shape 16-row MFMA kernel for every shape whose tile image fits its LDS budget, whatever
 * rnvp_shape::family says, and by the one-thread-per-row VALU kernel behind it (since round 6: e.g. hidden = (512,) on 16-d
 * rows; RNVP_EUNSUPPORTED only where even an 8-row tile of that kernel exceeds a CU's LDS, e.g. hidden = (2048, 2048));
 * workspace: rnvp_backward_cond_workspace_bytes(shape, n_rows), 0 for a shape no kernel serves.  Deterministic (no float
 * atomics).
 */
size_t rnvp_backward_cond_workspace_bytes(const rnvp_shape *shape, int64_t max_rows);
int rnvp_backward_cond(void *stream, const rnvp_shape *shape,
                       const float *params, const uint8_t *masks,
                       const float *x, const float *c, const int64_t *row_index,
                       int64_t n_rows, const float *gz, const float *gld,
                       float *grad_out, float *gx_out, float *gc_out,
                       void *workspace, size_t workspace_bytes);

/*
 * Backward THROUGH THE INVERSE x = g(z, c) (rnvp_inverse): the reference's `RealNVPLayer.g` / `NormalizingFlow.sample` are
 * ordinary autograd graphs (realnvp.py:120-129, nflow.py:141-145 -- the sampled tensor comes back with requires_grad), so
 * reverse-KL and other sample-based losses differentiate through them.  Given
 *   z [n_rows, d], c [n_rows, c]   the inputs of the inverse        gx [n_rows, d] = d loss / d x[r]
 * it returns
 *   grad_out [rnvp_param_count]  d loss / d params     gz_out [n_rows, d]  d loss / d z[r]   (nullable)
 *   gc_out [n_rows, c]  d loss / d c[r]  (nullable)
 * Per layer (applied L-1 .. 0, differentiated 0 .. L-1):  x_u = (y_u - t(y_m, c)) exp(-s(y_m, c))  gives
 *   d/dt = -gx_u exp(-s),  d/ds = -gx_u x_u,  gy_u = gx_u exp(-s),  gy_m = gx_m + (d/d y_m through both nets).
 * With shape->L == 1 this is the backward of one RealNVPLayer.g.  Kernel, workspace and determinism as rnvp_backward_cond.
 */
int rnvp_inverse_backward(void *stream, const rnvp_shape *shape,
                          const float *params, const uint8_t *masks,
                          const float *z, const float *c, int64_t n_rows, const float *gx,
                          float *grad_out, float *gz_out, float *gc_out,
                          void *workspace, size_t workspace_bytes);

/*
 * torch.optim.Adam step over the flat parameter buffer (realnvp.py:205-207,251):
 * betas/eps as given, amsgrad off, L2 weight decay folded into the gradient.
 * `step` is the 1-based step number of THIS update.
 */
int rnvp_adam_step(void *stream, float *params, const float *grad,
                   float *exp_avg, float *exp_avg_sq, int64_t n_params,
                   double lr, double beta1, double beta2, double eps,
                   double weight_decay, int64_t step);

/*
 * Second half of a DATA-PARALLEL training step (the reference has no distributed code): after
 * every rank's rnvp_loss_grad wrote its shard's gradient and loss share into one [P + 1] buffer and the caller
 * all-reduced (SUM) that buffer over RCCL, this single launch reads the batch loss out of grad_loss[P] into
 * loss_out[0] (the value realnvp.py:254 appends to loss_history) and applies rnvp_adam_step with
 * grad_loss[0..P) -- identical arithmetic on every rank, so the replicas stay bit-identical.
 */
int rnvp_dp_finish_step(void *stream, float *params, const float *grad_loss,
                        float *exp_avg, float *exp_avg_sq, int64_t n_params,
                        double lr, double beta1, double beta2, double eps,
                        double weight_decay, int64_t step, float *loss_out);

/*
 * rnvp_loss_grad followed by rnvp_adam_step on the same stream (single-GPU training step,
 * realnvp.py:246-254).  loss_out [1] receives the batch loss (the value the reference
 * appends to loss_history).  grad_buf [P] is scratch owned by the caller.
 *
 * A failure INSIDE an enqueued training launch cannot come back as a status.  The one such failure the library knows -- a
 * bounded wait of the register-chained training kernel's barrier-free gradient flush timing out (a protocol error; only the
 * forced test has ever produced it) -- is reported through the loss: the step applies NO parameter update and no re-pack and
 * the batch loss (rnvp_train_step, rnvp_fit_epoch*, rnvp_loss_grad) is the quiet NaN with the bit pattern RNVP_PROTOCOL_NAN_BITS;
 * under data parallelism every rank receives it through the all-reduced loss.  Every later batch of the same call reports it too
 * (fail-stop until the next call re-packs).  A NaN that training produced by itself carries another payload.
 */
#define RNVP_PROTOCOL_NAN_BITS 0x7fc0deadu
int rnvp_train_step(void *stream, const rnvp_shape *shape,
                    float *params, const uint8_t *masks,
                    const float *x, const float *c, const int64_t *row_index,
                    int64_t n_rows, float inv_B,
                    float *grad_buf, float *loss_out,
                    float *exp_avg, float *exp_avg_sq,
                    double lr, double beta1, double beta2, double eps,
                    double weight_decay, int64_t step,
                    void *workspace, size_t workspace_bytes);

/*
 * One epoch of the single-GPU batch loop of RealNVP.fit (realnvp.py:237-254) in one call: for every
 * consecutive slice of `batch_size` entries of `perm` (the epoch's DataLoader permutation, int64 [n]
 * on the device; the last slice may be ragged) run rnvp_train_step with inv_B = 1 / rows_in_slice
 * and store that batch's loss in loss_hist[k].  `first_step` is the 1-based Adam step number of the
 * first batch.  Removes the per-batch host round trip through Python; the kernels are enqueued
 * back to back on `stream`.  Returns the first non-zero status.
 */
int rnvp_fit_epoch(void *stream, const rnvp_shape *shape,
                   float *params, const uint8_t *masks,
                   const float *x, const float *c, const int64_t *perm, int64_t n, int64_t batch_size,
                   float *grad_buf, float *loss_hist,
                   float *exp_avg, float *exp_avg_sq,
                   double lr, double beta1, double beta2, double eps, double weight_decay,
                   int64_t first_step, void *workspace, size_t workspace_bytes);

/*
 * n_epochs epochs in one call: `perms` holds n_epochs permutations of the n rows back to back (int64 [n_epochs][n]), loss_hist
 * [n_epochs][ceil(n / batch_size)]; everything else as rnvp_fit_epoch, which is this call with n_epochs = 1.  Where
 * rnvp_fit_epoch_resident says so the whole call is ONE persistent launch (the parameters never leave LDS between epochs);
 * otherwise the epochs' batches are enqueued back to back.  RealNVP.fit uses it when no per-epoch progress text is asked for.
 */
int rnvp_fit_epochs(void *stream, const rnvp_shape *shape,
                    float *params, const uint8_t *masks,
                    const float *x, const float *c, const int64_t *perms, int64_t n, int64_t batch_size, int64_t n_epochs,
                    float *grad_buf, float *loss_hist,
                    float *exp_avg, float *exp_avg_sq,
                    double lr, double beta1, double beta2, double eps, double weight_decay,
                    int64_t first_step, void *workspace, size_t workspace_bytes);

/*
 * 1 when rnvp_fit_epoch runs this shape at this batch size as ONE persistent launch per epoch ("resident" fit,
 * rnvp_resident.hip): one hidden layer of at most 16 units (at most 32 while d + cdim <= 15) or two or three of at most 32
 * each, d <= 16, d + cdim <= 31, at most 16 layers, batch_size <= 128, and the model with its per-wave gradient stages inside one CU's 160 KB of LDS -- the
 * reference's default network (hidden=(10,), 8 layers, batch_size=32: realnvp.py:161-176).  Parameters stay in LDS for
 * the whole epoch, a step is a register-to-register MFMA chain per 16-row wave -- at batch_size <= 64 per wave pair, one
 * wave per net, with the weight gradients on helper waves (rnvp_resident_ns.hip) -- and Adam runs in place: 10.8 us per step
 * instead of 41 for the defaults.  The result agrees with the batch-by-batch loop to rounding (another summation order)
 * and reproduces itself bit for bit.  0: the loop of rnvp_train_step described above.  family == RNVP_FAMILY_VALU pins
 * the loop (test / measurement aid).
 */
int rnvp_fit_epoch_resident(const rnvp_shape *shape, int64_t batch_size);

/*
 * Data-parallel fit (the reference has no counterpart: its loop, realnvp.py:235-254, is single-process).
 * One process per GPU; every rank walks the SAME permutation and takes a contiguous share of each global batch.
 *
 * The library owns its RCCL communicator (librccl is dlopen'ed on first use: single-GPU callers never load it):
 *   rnvp_dp_unique_id   rank 0 fills 128 host bytes; the caller sends them to every rank (e.g. torch.distributed.broadcast)
 *   rnvp_dp_init        collective over all ranks; *comm_out is the handle for the calls below
 *   rnvp_dp_destroy     frees it
 *   rnvp_dp_all_reduce  in-place SUM of `count` floats on `stream` (what rnvp_fit_epoch_dp issues per batch)
 *
 * rnvp_fit_epoch_dp: all batches of one epoch, enqueued on ONE stream by one call -- per batch rnvp_loss_grad on this
 * rank's rows of perm[s0 : s0 + rows] (share [rank * rows / world ...), remainder to the low ranks; gradients scaled by
 * 1 / rows_global inside the kernel), the all-reduce of grad_loss[0 .. P] (P gradients + this rank's share of the batch
 * loss), then rnvp_dp_finish_step: the batch loss into loss_hist[k] and the identical Adam step on every rank.
 * grad_loss [P + 1]; everything else as rnvp_fit_epoch.  comm == NULL runs the same loop for one rank without RCCL.
 */
int rnvp_dp_unique_id(void *id_out_128_host_bytes);
int rnvp_dp_init(const void *id_128_host_bytes, int rank, int world, void **comm_out);
int rnvp_dp_destroy(void *comm);
int rnvp_dp_all_reduce(void *stream, void *comm, float *buf, int64_t count);
int rnvp_fit_epoch_dp(void *stream, void *comm, const rnvp_shape *shape,
                      float *params, const uint8_t *masks,
                      const float *x, const float *c, const int64_t *perm, int64_t n, int64_t batch_size,
                      float *grad_loss, float *loss_hist, float *exp_avg, float *exp_avg_sq,
                      double lr, double beta1, double beta2, double eps, double weight_decay,
                      int64_t first_step, void *workspace, size_t workspace_bytes);

/*
 * The same loop with the exchange supplied by the CALLER: `all_reduce(ctx, stream, buf, count)` must leave, on every rank,
 * the element-wise SUM over ranks of buf[0 .. count) in buf, ordered after the work already enqueued on `stream` and before
 * whatever is enqueued on it afterwards (an RCCL / MPI call on that stream, or a host-synchronous exchange); non-zero return =
 * failure, handed back to the caller.  `rank` / `world` place this process in the job.  rnvp_fit_epoch_dp is this call with
 * the library's RCCL communicator as the exchange.  Lets an embedding that owns its own communicator (torch.distributed
 * process groups, MPI) run the in-library batch loop, and lets the shard arithmetic be tested with two ranks on one GPU.
 */
typedef int (*rnvp_all_reduce_fn)(void *ctx, void *stream, float *buf, int64_t count);
int rnvp_fit_epoch_dp_cb(void *stream, rnvp_all_reduce_fn all_reduce, void *ctx, int rank, int world,
                         const rnvp_shape *shape, float *params, const uint8_t *masks,
                         const float *x, const float *c, const int64_t *perm, int64_t n, int64_t batch_size,
                         float *grad_loss, float *loss_hist, float *exp_avg, float *exp_avg_sq,
                         double lr, double beta1, double beta2, double eps, double weight_decay,
                         int64_t first_step, void *workspace, size_t workspace_bytes);

/*
 * The step's exchange in CHUNKS OF LAYERS (round 6; register-chained kernels only, other shapes ignore it).  Unchunked, a rank's
 * step is strictly serial: training kernel -> partial sums -> all-reduce of [gradient | loss] -> Adam + re-pack, so the
 * cross-GPU latency of the all-reduce is paid in full.  With `chunks` = K > 1 the layers are cut into K groups, last layers
 * first (the first message carries the batch loss), and per group:  partial sums of the group -> all-reduce of the group's
 * slice of the flat gradient -> Adam + re-pack of the group's layers.
 *   rnvp_dp_set_chunks(comm, K)    the library's RCCL communicator: group j's all-reduce runs on a side stream of the communicator
 *                                  while the caller's stream sums group j + 1 (events order the two; the next batch starts behind the
 *                                  last group's Adam).  K = 1 (default) restores the one-message loop.  1 <= K <= 8.
 *   rnvp_fit_epoch_dp_cb_chunked   the caller's exchange, called K times per batch on `stream` (no overlap: the exchange is the
 *                                  caller's to schedule) -- the same cut, for process groups that are not RCCL and for tests.
 * Per parameter the arithmetic is the unchunked loop's (same partial sums in the same order, same Adam): identical bits on one
 * rank and wherever the exchange's sum does not depend on how the message is cut (two ranks; any exchange adding in rank order).
 * Every rank must use the same K.  One GPU cannot show the benefit (nothing to hide: K = 4 costs ~2 more launches' worth of
 * queue time per step on a one-rank communicator, bench.py secondary_configs.dp8_rank_steps); it is there for the multi-GPU job.
 */
int rnvp_dp_set_chunks(void *comm, int chunks);
int rnvp_fit_epoch_dp_cb_chunked(void *stream, rnvp_all_reduce_fn all_reduce, void *ctx, int rank, int world, int chunks,
                                 const rnvp_shape *shape, float *params, const uint8_t *masks,
                                 const float *x, const float *c, const int64_t *perm, int64_t n, int64_t batch_size,
                                 float *grad_loss, float *loss_hist, float *exp_avg, float *exp_avg_sq,
                                 double lr, double beta1, double beta2, double eps, double weight_decay,
                                 int64_t first_step, void *workspace, size_t workspace_bytes);

/*
 * Measurement aid (bench.py): while enabled, the hot kernel of each call -- the fused forward+backward
 * kernel of rnvp_loss_grad / rnvp_train_step (RNVP_PROFILE_TRAIN), the stack kernel of
 * rnvp_forward_logprob (RNVP_PROFILE_FORWARD) and of rnvp_inverse / rnvp_sample (RNVP_PROFILE_INVERSE) --
 * is bracketed by a pair of HIP events recorded on the caller's stream.  rnvp_profile_read synchronises
 * on the events of one kind and returns the number of bracketed launches and their summed duration in
 * milliseconds, then resets that kind.  At most `capacity` launches per kind are bracketed between two
 * reads (later ones run un-timed).  The event table is guarded by a mutex; has no equivalent in the
 * reference.
 */
#define RNVP_PROFILE_TRAIN   0
#define RNVP_PROFILE_FORWARD 1
#define RNVP_PROFILE_INVERSE 2
int rnvp_profile_enable(int capacity);          /* capacity <= 0 disables and frees the events */
int rnvp_profile_read(int kind, int *n_launches, float *total_ms);

/*
 * Which kernel the CALLING THREAD's most recent call of a kind (RNVP_PROFILE_TRAIN / _FORWARD / _INVERSE) launched as
 * its hot kernel: every launch site writes this record itself, next to the launch, so it cannot drift from the
 * dispatch rules.  Tests assert it (a later change of a threshold cannot silently move a benchmark size onto an
 * untested kernel) and bench.py labels its JSON line from it.  Thread-local; no equivalent in the reference.
 */
#define RNVP_VARIANT_NONE        0
#define RNVP_VARIANT_ROWPAR      1   /* row-parallel: 4 waves per workgroup, each with both nets of its row tiles          */
#define RNVP_VARIANT_NETSPLIT    2   /* k_mfma_train<.., NS = 1>: 8 waves, wave pairs share row tiles, one net each        */
#define RNVP_VARIANT_WIDE        3   /* k_mfma_train_wide: 8 row-owning waves per workgroup (d in (16, 32])                */
#define RNVP_VARIANT_TILESPLIT   4   /* k_mfma_train_ts / k_mfma_flow_ts: the workgroup's waves split the hidden tiles     */
#define RNVP_VARIANT_BX3_STAGED  5   /* k_flow_bx3<.., STAGED>: split-bf16 GEMM1, weights staged in LDS by LDS-DMA         */
#define RNVP_VARIANT_BX3_DIRECT  6   /* k_flow_bx3, barrier-free form (d <= 16)                                           */
#define RNVP_VARIANT_LMM         7   /* any-shape MFMA kernels, LDS-resident activations                                  */
#define RNVP_VARIANT_VALU        8   /* one thread per row                                                                */
#define RNVP_VARIANT_RESIDENT    9   /* persistent one-workgroup epoch (rnvp_resident*.hip)                               */
typedef struct rnvp_dispatch {
    int32_t variant;      /* RNVP_VARIANT_*                                                              */
    int32_t row_tiles;    /* 16-row tiles per wave                                                       */
    int32_t waves;        /* waves per workgroup                                                         */
    int32_t grid;         /* workgroups                                                                  */
    int32_t gemm1_fwd;    /* RNVP_PREC_F32 / RNVP_PREC_BX3: arithmetic the first Linear ran in (forward phase) */
    int32_t launches;     /* kernel launches the whole call enqueued (pack, hot kernel, follow-ups)      */
    int64_t rows;         /* rows of the hot launch                                                      */
    char    kernel[48];   /* name of the hot kernel as rocprofv3 lists it (without template arguments)   */
} rnvp_dispatch;
int rnvp_last_dispatch(int kind, rnvp_dispatch *out);

#ifdef __cplusplus
}
#endif
#endif /* RNVP_HIP_H */
