// rnvp_dp.hip -- the data-parallel batch loop of RealNVP.fit inside the library.
//
// The reference has no distributed code; its loop (/root/reference/probaforms/models/realnvp.py:235-254) runs one batch
// after the other in Python.  Sharded over N GPUs every batch needs ONE exchange, the all-reduce of the flat
// [gradient | loss] buffer.  Driving that from Python (loss + gradient on torch's stream, torch.distributed's all-reduce
// on RCCL's own stream, Adam back on torch's stream) costs two cross-stream event hops and three host round trips per
// batch -- measured 54 us of queue bubbles per 427 us batch.  Here the whole epoch is enqueued by one call on ONE stream:
//     for every batch:  rnvp_loss_grad (this rank's rows, scaled by 1 / B_global)
//                       ncclAllReduce(SUM) of [gradient | loss] on the SAME stream (RCCL, over xGMI between GPUs)
//                       rnvp_dp_finish_step (batch loss out of the message + the identical Adam step on every rank)
// RCCL is loaded with dlopen at rnvp_dp_init (no link-time dependency: single-GPU users never touch it); the communicator
// is the library's own, built from an id the caller distributes with whatever it has (torch.distributed broadcast).
#include <dlfcn.h>

#include <mutex>

#include "rnvp_common.h"
#include "rnvp_mfma.h"

namespace {

// the few RCCL entry points used, with the types of <rccl/rccl.h> restated (ncclUniqueId is 128 opaque bytes)
struct UniqueId { char internal[128]; };
typedef int (*fn_get_unique_id)(UniqueId *);
typedef int (*fn_comm_init_rank)(void **comm, int nranks, UniqueId id, int rank);
typedef int (*fn_comm_destroy)(void *comm);
typedef int (*fn_all_reduce)(const void *send, void *recv, size_t count, int dtype, int op, void *comm, hipStream_t st);
typedef const char *(*fn_error_string)(int);
constexpr int kNcclFloat32 = 7, kNcclSum = 0;

struct Rccl {
    void *handle = nullptr;
    fn_get_unique_id get_unique_id = nullptr;
    fn_comm_init_rank comm_init_rank = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_all_reduce all_reduce = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

int load_rccl() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.handle) return RNVP_OK;
    void *h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return RNVP_EUNSUPPORTED;
    Rccl r;
    r.get_unique_id = reinterpret_cast<fn_get_unique_id>(dlsym(h, "ncclGetUniqueId"));
    r.comm_init_rank = reinterpret_cast<fn_comm_init_rank>(dlsym(h, "ncclCommInitRank"));
    r.comm_destroy = reinterpret_cast<fn_comm_destroy>(dlsym(h, "ncclCommDestroy"));
    r.all_reduce = reinterpret_cast<fn_all_reduce>(dlsym(h, "ncclAllReduce"));
    if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.all_reduce) return RNVP_EUNSUPPORTED;
    r.handle = h;
    g_rccl = r;
    return RNVP_OK;
}

constexpr int kMaxChunks = 8;
struct DpComm {
    void *comm;
    int rank, world;
    // rnvp_dp_set_chunks: the step's [gradient | loss] message in chunks of layers, each chunk's all-reduce on `side` while the
    // main stream sums the next chunk's partials
    int chunks = 1;
    hipStream_t side = nullptr;
    hipEvent_t ev_sum[kMaxChunks] = {}, ev_red[kMaxChunks] = {};
};

// chunk j of `chunks` over L layers, LAST layers first (chunk 0 ends at L and carries the batch loss, so that every later chunk's
// Adam launch can read the all-reduced loss for the cross-rank error check): layers [l0, l1)
inline void chunk_layers(int L, int chunks, int j, int *l0, int *l1) {
    *l1 = L - (int)((int64_t)L * j / chunks);
    *l0 = L - (int)((int64_t)L * (j + 1) / chunks);
}

}  // namespace

extern "C" {

int rnvp_dp_unique_id(void *id_out) {
    if (!id_out) return RNVP_EINVAL;
    int rc = load_rccl();
    if (rc) return rc;
    UniqueId id;
    if (g_rccl.get_unique_id(&id) != 0) return RNVP_EUNSUPPORTED;
    std::memcpy(id_out, &id, sizeof(id));
    return RNVP_OK;
}

int rnvp_dp_init(const void *id, int rank, int world, void **comm_out) {
    if (!id || !comm_out || world < 1 || rank < 0 || rank >= world) return RNVP_EINVAL;
    int rc = load_rccl();
    if (rc) return rc;
    UniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    void *comm = nullptr;
    if (g_rccl.comm_init_rank(&comm, world, uid, rank) != 0 || !comm) return RNVP_EUNSUPPORTED;
    *comm_out = new DpComm{comm, rank, world};
    return RNVP_OK;
}

int rnvp_dp_set_chunks(void *comm, int chunks) {
    if (!comm || chunks < 1 || chunks > kMaxChunks) return RNVP_EINVAL;
    DpComm *c = static_cast<DpComm *>(comm);
    if (chunks > 1 && !c->side) {
        // all or nothing: a communicator never keeps a side stream without its events (rnvp_dp_destroy would destroy handles that
        // were never created)
        hipStream_t side = nullptr;
        hipEvent_t ev[2 * kMaxChunks] = {};
        hipError_t e = hipStreamCreateWithFlags(&side, hipStreamNonBlocking);
        int made = 0;
        for (; e == hipSuccess && made < 2 * kMaxChunks; ++made) e = hipEventCreateWithFlags(&ev[made], hipEventDisableTiming);
        if (e != hipSuccess) {
            for (int j = 0; j < made - 1; ++j) (void)hipEventDestroy(ev[j]);
            if (side) (void)hipStreamDestroy(side);
            return (int)e;
        }
        for (int j = 0; j < kMaxChunks; ++j) { c->ev_sum[j] = ev[2 * j]; c->ev_red[j] = ev[2 * j + 1]; }
        c->side = side;
    }
    c->chunks = chunks;
    return RNVP_OK;
}

int rnvp_dp_destroy(void *comm) {
    if (!comm) return RNVP_OK;
    DpComm *c = static_cast<DpComm *>(comm);
    if (c->side) {
        (void)hipStreamSynchronize(c->side);
        for (int j = 0; j < kMaxChunks; ++j) { (void)hipEventDestroy(c->ev_sum[j]); (void)hipEventDestroy(c->ev_red[j]); }
        (void)hipStreamDestroy(c->side);
    }
    if (c->comm && g_rccl.comm_destroy) (void)g_rccl.comm_destroy(c->comm);
    delete c;
    return RNVP_OK;
}

int rnvp_dp_all_reduce(void *stream, void *comm, float *buf, int64_t count) {
    if (!comm || !buf || count < 0) return RNVP_EINVAL;
    DpComm *c = static_cast<DpComm *>(comm);
    if (g_rccl.all_reduce(buf, buf, (size_t)count, kNcclFloat32, kNcclSum, c->comm, static_cast<hipStream_t>(stream)) != 0)
        return RNVP_EUNSUPPORTED;
    return RNVP_OK;
}

static int rccl_all_reduce_cb(void *ctx, void *stream, float *buf, int64_t count) {
    return rnvp_dp_all_reduce(stream, ctx, buf, count);
}

static int fit_epoch_dp_impl(void *stream, rnvp_all_reduce_fn all_reduce, void *ctx, int rank, int world, int chunks, DpComm *overlap,
                             const rnvp_shape *shape, float *params, const uint8_t *masks,
                             const float *x, const float *c, const int64_t *perm, int64_t n, int64_t batch_size,
                             float *grad_loss, float *loss_hist, float *exp_avg, float *exp_avg_sq,
                             double lr, double beta1, double beta2, double eps, double weight_decay,
                             int64_t first_step, void *workspace, size_t workspace_bytes);

int rnvp_fit_epoch_dp(void *stream, void *comm, const rnvp_shape *shape, float *params, const uint8_t *masks,
                      const float *x, const float *c, const int64_t *perm, int64_t n, int64_t batch_size,
                      float *grad_loss, float *loss_hist, float *exp_avg, float *exp_avg_sq,
                      double lr, double beta1, double beta2, double eps, double weight_decay,
                      int64_t first_step, void *workspace, size_t workspace_bytes) {
    // comm == NULL: one rank, no exchange (the same step sequence, for tests of the loop itself)
    DpComm *dc = static_cast<DpComm *>(comm);
    return fit_epoch_dp_impl(stream, dc ? rccl_all_reduce_cb : nullptr, comm, dc ? dc->rank : 0, dc ? dc->world : 1, dc ? dc->chunks : 1,
                             (dc && dc->chunks > 1) ? dc : nullptr, shape, params, masks, x, c, perm, n, batch_size, grad_loss, loss_hist,
                             exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, first_step, workspace, workspace_bytes);
}

int rnvp_fit_epoch_dp_cb_chunked(void *stream, rnvp_all_reduce_fn all_reduce, void *ctx, int rank, int world, int chunks,
                                 const rnvp_shape *shape, float *params, const uint8_t *masks,
                                 const float *x, const float *c, const int64_t *perm, int64_t n, int64_t batch_size,
                                 float *grad_loss, float *loss_hist, float *exp_avg, float *exp_avg_sq,
                                 double lr, double beta1, double beta2, double eps, double weight_decay,
                                 int64_t first_step, void *workspace, size_t workspace_bytes) {
    if (chunks < 1 || chunks > kMaxChunks) return RNVP_EINVAL;
    return fit_epoch_dp_impl(stream, all_reduce, ctx, rank, world, chunks, nullptr, shape, params, masks, x, c, perm, n, batch_size,
                             grad_loss, loss_hist, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, first_step, workspace,
                             workspace_bytes);
}

int rnvp_fit_epoch_dp_cb(void *stream, rnvp_all_reduce_fn all_reduce, void *ctx, int rank, int world,
                         const rnvp_shape *shape, float *params, const uint8_t *masks,
                         const float *x, const float *c, const int64_t *perm, int64_t n, int64_t batch_size,
                         float *grad_loss, float *loss_hist, float *exp_avg, float *exp_avg_sq,
                         double lr, double beta1, double beta2, double eps, double weight_decay,
                         int64_t first_step, void *workspace, size_t workspace_bytes) {
    return fit_epoch_dp_impl(stream, all_reduce, ctx, rank, world, 1, nullptr, shape, params, masks, x, c, perm, n, batch_size, grad_loss,
                             loss_hist, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, first_step, workspace, workspace_bytes);
}

}  // extern "C"

// chunks > 1 (register-chained kernels only): per batch the training launch, then for every chunk of layers (last layers first)
// the partial sums of the chunk -> its all-reduce -> Adam + re-pack of the chunk.  With `overlap` (the library's communicator,
// rnvp_dp_set_chunks) chunk j's all-reduce runs on the communicator's side stream while the main stream sums chunk j + 1: the
// cross-GPU latency of the exchange hides under the sums instead of following them.  The arithmetic per parameter is that of the
// unchunked loop (same partial sums, same Adam): identical bits on one rank and wherever the exchange's sum does not depend on how
// the message is cut (two ranks; any exchange that adds in rank order).
static int fit_epoch_dp_impl(void *stream, rnvp_all_reduce_fn all_reduce, void *ctx, int rank, int world, int chunks, DpComm *overlap,
                             const rnvp_shape *shape, float *params, const uint8_t *masks,
                             const float *x, const float *c, const int64_t *perm, int64_t n, int64_t batch_size,
                             float *grad_loss, float *loss_hist, float *exp_avg, float *exp_avg_sq,
                             double lr, double beta1, double beta2, double eps, double weight_decay,
                             int64_t first_step, void *workspace, size_t workspace_bytes) {
    if (n < 0 || batch_size < 1 || !perm || !loss_hist || !grad_loss || first_step < 1) return RNVP_EINVAL;
    if (world < 1 || rank < 0 || rank >= world || (world > 1 && !all_reduce)) return RNVP_EINVAL;
    rnvp::KShape ks;
    int rc = rnvp::make_kshape(shape, &ks);
    if (rc) return rc;
    const int64_t P = (int64_t)rnvp_param_count(shape);
    if (P <= 0) return RNVP_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // register-chained kernels: the packed weight fragments live in the workspace for the whole call -- packed by the first
    // batch's rnvp_loss_grad, then re-packed by every batch's Adam launch (rnvp::mfma::adam_pack)
    const bool chained = rnvp::mfma::train_supported(ks);
    const int nchunks = (chained && chunks > 1) ? (chunks < ks.L ? chunks : ks.L) : 1;
    bool packed_valid = false;
    int64_t k = 0;
    for (int64_t s0 = 0; s0 < n; s0 += batch_size, ++k) {
        const int64_t rows = (n - s0 < batch_size) ? n - s0 : batch_size;
        // contiguous share of the global batch; the remainder rows go to the low ranks (a ragged batch may leave high
        // ranks with no rows: rnvp_loss_grad then writes zeros)
        const int64_t base = rows / world, rem = rows % world;
        const int64_t lo = s0 + rank * base + (rank < rem ? rank : rem);
        const int64_t mine = base + (rank < rem ? 1 : 0);
        if (chained && nchunks > 1) {
            // (every rank issues the SAME sequence of exchanges, also one whose share of a ragged batch is empty: it contributes zeros)
            if (!params || (!masks && !ks.alt) || !x || (ks.c > 0 && !c) || !exp_avg || !exp_avg_sq) return RNVP_EINVAL;
            rnvp::mfma::PendingPartials pend{};
            if (mine > 0) {
                rc = rnvp::mfma::loss_partials(st, ks, params, x, c, perm + lo, mine, 1.0f / (float)rows, workspace, workspace_bytes,
                                               packed_valid, &pend);
                if (rc) return rc;
            } else {
                RNVP_HIP_TRY(hipMemsetAsync(grad_loss, 0, (size_t)(P + 1) * sizeof(float), st));
            }
            const int64_t per_layer = P / ks.L;
            for (int j = 0; j < nchunks; ++j) {
                int l0, l1;
                chunk_layers(ks.L, nchunks, j, &l0, &l1);
                if (mine > 0) {
                    rc = rnvp::mfma::finish_sum_layers(st, ks, pend, l0, l1 - l0, grad_loss, j == 0 ? grad_loss + P : nullptr, workspace,
                                                       workspace_bytes);
                    if (rc) return rc;
                }
                float *msg = grad_loss + (int64_t)l0 * per_layer;
                const int64_t cnt = (int64_t)(l1 - l0) * per_layer + (j == 0 ? 1 : 0);
                if (!all_reduce) continue;
                if (overlap) {
                    RNVP_HIP_TRY(hipEventRecord(overlap->ev_sum[j], st));
                    RNVP_HIP_TRY(hipStreamWaitEvent(overlap->side, overlap->ev_sum[j], 0));
                    rc = all_reduce(ctx, overlap->side, msg, cnt);
                    if (rc) return rc;
                    RNVP_HIP_TRY(hipEventRecord(overlap->ev_red[j], overlap->side));
                } else {
                    rc = all_reduce(ctx, stream, msg, cnt);
                    if (rc) return rc;
                }
            }
            const rnvp::AdamK ak = rnvp::make_adam(lr, beta1, beta2, eps, weight_decay, first_step + k);
            for (int j = 0; j < nchunks; ++j) {
                int l0, l1;
                chunk_layers(ks.L, nchunks, j, &l0, &l1);
                if (overlap && all_reduce) RNVP_HIP_TRY(hipStreamWaitEvent(st, overlap->ev_red[j], 0));
                rc = rnvp::mfma::adam_pack_layers(st, ks, params, grad_loss, grad_loss + P, j == 0 ? loss_hist + k : nullptr, exp_avg,
                                                  exp_avg_sq, ak, workspace, workspace_bytes, l0, l1 - l0);
                if (rc) return rc;
            }
            packed_valid = true;
            continue;
        }
        if (chained && mine > 0) {
            if (!params || (!masks && !ks.alt) || !x || (ks.c > 0 && !c)) return RNVP_EINVAL;
            rc = rnvp::mfma::loss_grad(st, ks, params, x, c, perm + lo, mine, 1.0f / (float)rows, grad_loss, grad_loss + P, workspace,
                                       workspace_bytes, rnvp::Seeds{}, packed_valid);
        } else {
            rc = rnvp_loss_grad(stream, shape, params, masks, x, c, perm + lo, mine, 1.0f / (float)rows, grad_loss,
                                grad_loss + P, workspace, workspace_bytes);
        }
        if (rc) return rc;
        if (all_reduce) {               // (a one-rank communicator still runs its trivial exchange: the call sequence of N ranks)
            rc = all_reduce(ctx, stream, grad_loss, P + 1);
            if (rc) return rc;
        }
        if (chained) {
            if (!exp_avg || !exp_avg_sq) return RNVP_EINVAL;
            rc = rnvp::mfma::adam_pack(st, ks, params, grad_loss, grad_loss + P, loss_hist + k, exp_avg, exp_avg_sq,
                                       rnvp::make_adam(lr, beta1, beta2, eps, weight_decay, first_step + k), workspace, workspace_bytes);
            packed_valid = true;
        } else {
            rc = rnvp_dp_finish_step(stream, params, grad_loss, exp_avg, exp_avg_sq, P, lr, beta1, beta2, eps, weight_decay,
                                     first_step + k, loss_hist + k);
        }
        if (rc) return rc;
    }
    return RNVP_OK;
}
