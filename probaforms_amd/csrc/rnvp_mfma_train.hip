// rnvp_mfma_train.hip -- host side of the fused forward + backward step on f32 MFMA (gfx950): workspace plan, dispatch over
// the tile geometries (device code and launch templates: rnvp_mfma_train_dev.h, instantiated per geometry in
// rnvp_mfma_train_nf{2,4,8}.hip) and the second stage that turns per-workgroup partial gradients into the flat
// reference-order gradient (+ Adam).  Replaces `loss = -nf.log_prob(X, C); loss.backward()` and `opt.step()`
// (/root/reference/probaforms/models/realnvp.py:246-251).
#include "rnvp_mfma_train_dev.h"

namespace rnvp {
namespace mfma {

#define RNVP_LAUNCH_DECL(nf, cq)                                                                                          \
    int launch_train_##nf##_##cq(hipStream_t st, const KShape &k, const Geo &g, const TrainPlan &pl, const float *packed,  \
                                 const float *x, const float *c, const int64_t *row_index, int64_t n, float inv_B,         \
                                 float *gpart, float *losspart, float *scratch, int *grid_out, Seeds sd, PartialLayout *lay);
RNVP_LAUNCH_DECL(2, 1) RNVP_LAUNCH_DECL(2, 0) RNVP_LAUNCH_DECL(4, 2) RNVP_LAUNCH_DECL(8, 4)
#undef RNVP_LAUNCH_DECL

namespace {

// ---- stage 2: segment sums -> flat reference-order gradient -----------------------------------------
// One thread per flat parameter; finds where the packed gradient keeps it (or that the masks make
// it dead: exactly zero, as in the reference) and sums the workgroup partials in index order.
__global__ void __launch_bounds__(256)
k_mfma_reduce(KShape k, Geo g, int NTI, int glayer_floats, int w2c, const float *__restrict__ seg, int S,
              const float *__restrict__ losspart, int G, float inv_B, float *__restrict__ grad, float *loss,
              float *adam_p, float *adam_m, float *adam_v, AdamK adam) {
    // w2c: the launch wrote the compact dW2 records of Dims::w2c (128 floats per hidden tile after the dW1 tiles)
    const size_t P = (size_t)2 * k.npn * k.L;
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) {
        // the last block's first wave reduces the per-wave loss partials (fixed order: deterministic)
        if (loss && blockIdx.x == gridDim.x - 1 && threadIdx.x >= 192) {
            const int lane = threadIdx.x - 192;
            float a = 0.f;
            for (int i = lane; i < G * kWaves; i += 64) a += losspart[i];
            for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off);
            if (lane == 0) loss[0] = -a * inv_B;
        }
        return;
    }
    const int l = (int)(p / (2 * (size_t)k.npn));
    int idx = (int)(p - (size_t)l * 2 * k.npn);
    const int net = idx >= k.npn;
    idx -= net * k.npn;
    const int pc = (l + k.alt) & 1;
    const int NF = g.NF, CQ = g.CQ, HT = g.HT, OTL = g.OTL, h = k.nout[0], nin = k.d + k.c;
    const int KSP = 4 * NTI;
    const int tblk = w2c ? NTI * 256 + 128 : (NTI + OTL) * 256;      // floats of one hidden tile's record
    const int netblock = HT * tblk;
    int loc = -1;
    if (idx < k.boff[0]) {                                     // W1 [h][d + c]
        const int hid = idx / nin, col = idx - hid * nin;
        int jn = -1;                                           // (k.d, k.c, h are the REAL sizes; NF, CQ the padded tiles)
        if (col < k.d) {
            const int qq = col / (2 * NF), e = col % (2 * NF);
            if ((e & 1) == pc) jn = qq * KSP + (e >> 1);
        } else if (CQ > 0) {
            const int ci = col - k.d;
            jn = (ci / CQ) * KSP + NF + (ci % CQ);
        }
        if (jn >= 0) {
            const int i = hid & 15;
            loc = net * netblock + (hid >> 4) * tblk + (jn >> 4) * 256 + (16 * (i >> 2) + (jn & 15)) * 4 + (i & 3);
        }
    } else if (idx < k.woff[1]) {                              // b1 [h]: the ones column
        const int hid = idx - k.boff[0], i = hid & 15, jn = NF + CQ;
        loc = net * netblock + (hid >> 4) * tblk + (jn >> 4) * 256 + (16 * (i >> 2) + (jn & 15)) * 4 + (i & 3);
    } else if (idx < k.boff[1]) {                              // W2 [d][h]
        const int j = (idx - k.woff[1]) / h, hid = (idx - k.woff[1]) - j * h;
        const int qo = j / (2 * NF), e = j % (2 * NF);
        if ((e & 1) == 1 - pc) {
            const int f = e >> 1;
            int otl, io;
            if (NF >= 4) { otl = f >> 2; io = 4 * qo + (f & 3); }
            else { otl = 0; io = 4 * qo + 2 * net + f; }
            const int i = hid & 15;
            if (w2c) {      // column 2 qo + f of this net: block cb = col >> 2, kept by lane group 2 cb + (i >> 1) & 1 (layer_bwd)
                const int col = 2 * qo + f, lane = 16 * (2 * (col >> 2) + ((i >> 1) & 1)) + 4 * (i >> 2) + (col & 3);
                loc = net * netblock + (hid >> 4) * tblk + NTI * 256 + lane * 2 + (i & 1);
            } else {
                loc = net * netblock + (hid >> 4) * tblk + (NTI + otl) * 256 + (16 * (i >> 2) + io) * 4 + (i & 3);
            }
        }
    } else {                                                   // b2 [d]
        const int j = idx - k.boff[1];
        const int qo = j / (2 * NF), e = j % (2 * NF);
        if ((e & 1) == 1 - pc) {
            const int f = e >> 1;
            int ot, reg;
            if (NF >= 4) { ot = net * OTL + (f >> 2); reg = f & 3; }
            else { ot = 0; reg = 2 * net + f; }
            loc = 2 * netblock + (ot * 4 + qo) * 4 + reg;
        }
    }
    float a = 0.f;
    if (loc >= 0) {
        const float *src = seg + (size_t)l * glayer_floats + loc;
        const size_t stride = (size_t)glayer_floats * k.L;
        if (S == kSeg) {            // the usual case: all sixteen loads in flight, then added in segment order
            float v[kSeg];
#pragma unroll
            for (int b = 0; b < kSeg; ++b) v[b] = __builtin_nontemporal_load(src + (size_t)b * stride);
#pragma unroll
            for (int b = 0; b < kSeg; ++b) a += v[b];
        } else {
            for (int b = 0; b < S; ++b) a += src[(size_t)b * stride];      // segment order: deterministic
        }
    }
    grad[p] = a;
    if (adam_p) adam_one(adam_p[p], a, adam_m[p], adam_v[p], adam);  // fused optimizer (rnvp_train_step)
}

template <int NF, int CQ>
TrainPlan make_plan(const Geo &g, int L) {
    using DM = Dims<NF, CQ>;
    TrainPlan p;
    const int netblock = g.HT * (DM::NTI + DM::OTL) * 256;
    p.glayer_floats = 2 * netblock + DM::NT2 * 16;
    p.RMAX = TrainRows<NF, CQ>::value;
    p.lds_bytes = ((size_t)kWaves * DM::SLOT + (size_t)kWaves * DM::template tb<TrainRows<NF, CQ>::value>()) * sizeof(float);
    p.scratch_per_wave = (size_t)L * p.RMAX * 2 * NF * 64;
    return p;
}

bool plan_for(const Geo &g, int L, TrainPlan *p) {
    if (g.NF == 2 && g.CQ == 1) { *p = make_plan<2, 1>(g, L); return true; }
    if (g.NF == 2 && g.CQ == 0) { *p = make_plan<2, 0>(g, L); return true; }
    if (g.NF == 4 && g.CQ == 2) { *p = make_plan<4, 2>(g, L); return true; }
    if (g.NF == 8 && g.CQ == 4) { *p = make_plan<8, 4>(g, L); return true; }
    return false;
}

}  // namespace

// rnvp_backward (per-row d loss / d logdet, d loss / d x out) is served by the tile-split kernel only: the row-parallel
// kernels have no register to spare for the extra output (measured: +64 B of scratch per lane in the C2 kernel).  Larger
// calls go to the any-shape MFMA kernels (rnvp_lmm.hip), which take every shape.
bool backward_rows_ok(const KShape &k, int64_t n) {
    if (!train_supported(k)) return false;
    const Geo g = make_geo(k.d, k.c, k.nout[0], kTrainSplit);
    return n <= ts_max_rows(g);
}

bool train_supported(const KShape &k) {
    if (!supported(k)) return false;
    const Geo g = make_geo(k.d, k.c, k.nout[0], kTrainSplit);
    TrainPlan pl;
    if (!plan_for(g, k.L, &pl)) return false;
    return pl.lds_bytes <= 160 * 1024;
}

size_t train_workspace_bytes(const KShape &k, int64_t max_rows) {
    (void)max_rows;
    const Geo g = make_geo(k.d, k.c, k.nout[0], kTrainSplit);
    TrainPlan pl;
    if (!plan_for(g, k.L, &pl)) return 0;
    size_t b = align_up((size_t)g.layer_floats * k.L * sizeof(float), 256);                    // packed weights
    b += align_up((size_t)kMaxGridTrain * pl.glayer_floats * k.L * sizeof(float), 256);          // partials
    b += align_up((size_t)kSeg * pl.glayer_floats * k.L * sizeof(float), 256);                   // segment sums
    b += align_up((size_t)kMaxGridTrain * kWaves * sizeof(float), 256);                          // loss partials
    b += align_up((size_t)kMaxGridTrain * kWaves * pl.scratch_per_wave * sizeof(float), 256);    // saved activations
    return b;
}

static int loss_grad_impl(hipStream_t st, const KShape &k, const float *params, const float *x, const float *c,
                          const int64_t *row_index, int64_t n, float inv_B, float *grad_out, float *loss_out,
                          void *ws, size_t ws_bytes, float *adam_p, float *adam_m, float *adam_v, AdamK adam,
                          Seeds sd = Seeds{}) {
    if (!ws || ws_bytes < train_workspace_bytes(k, n)) return RNVP_EWORKSPACE;
    const Geo g = make_geo(k.d, k.c, k.nout[0], kTrainSplit);
    if ((sd.gld || sd.gx) && n > ts_max_rows(g)) return RNVP_EUNSUPPORTED;       // see backward_rows_ok
    TrainPlan pl;
    if (!plan_for(g, k.L, &pl)) return RNVP_EUNSUPPORTED;
    char *w = static_cast<char *>(ws);
    float *packed = reinterpret_cast<float *>(w);
    w += align_up((size_t)g.layer_floats * k.L * sizeof(float), 256);
    float *gpart = reinterpret_cast<float *>(w);
    w += align_up((size_t)kMaxGridTrain * pl.glayer_floats * k.L * sizeof(float), 256);
    float *seg = reinterpret_cast<float *>(w);
    w += align_up((size_t)kSeg * pl.glayer_floats * k.L * sizeof(float), 256);
    float *losspart = reinterpret_cast<float *>(w);
    w += align_up((size_t)kMaxGridTrain * kWaves * sizeof(float), 256);
    float *scratch = reinterpret_cast<float *>(w);
    int rc = pack_weights(st, k, g, params, packed);
    if (rc) return rc;
    int grid = 0;
    PartialLayout lay{pl.glayer_floats, 0};
    if (g.NF == 2 && g.CQ == 1) rc = launch_train_2_1(st, k, g, pl, packed, x, c, row_index, n, inv_B, gpart, losspart, scratch, &grid, sd, &lay);
    else if (g.NF == 2 && g.CQ == 0) rc = launch_train_2_0(st, k, g, pl, packed, x, c, row_index, n, inv_B, gpart, losspart, scratch, &grid, sd, &lay);
    else if (g.NF == 4 && g.CQ == 2) rc = launch_train_4_2(st, k, g, pl, packed, x, c, row_index, n, inv_B, gpart, losspart, scratch, &grid, sd, &lay);
    else if (g.NF == 8 && g.CQ == 4) rc = launch_train_8_4(st, k, g, pl, packed, x, c, row_index, n, inv_B, gpart, losspart, scratch, &grid, sd, &lay);
    else return RNVP_EUNSUPPORTED;
    if (rc) return rc;
    const size_t P = (size_t)2 * k.npn * k.L;
    const unsigned blocks = (unsigned)(P / 256 + 2);     // last block's last wave is always past P: it sums the loss
    int NTI = (g.KS1 + 1 + 3) / 4;
    const size_t n4 = (size_t)lay.glayer_floats * k.L / 4;
    // second level (<= kSeg segment sums per parameter) is folded into the scatter to flat order; up to kSeg
    // workgroups ARE the segments (small batches: one launch fewer per step)
    const int S = grid < kSeg ? grid : kSeg;
    if (grid > kSeg) {
        hipLaunchKernelGGL(k_sum_segments, dim3((unsigned)((n4 + 255) / 256), kSeg), dim3(256), 0, st, gpart, grid, n4, seg);
        RNVP_HIP_TRY(hipGetLastError());
    } else {
        seg = gpart;
    }
    hipLaunchKernelGGL(k_mfma_reduce, dim3(blocks), dim3(256), 0, st, k, g, NTI, lay.glayer_floats, lay.w2c, seg, S, losspart,
                       grid, inv_B, grad_out, loss_out, adam_p, adam_m, adam_v, adam);
    RNVP_HIP_TRY(hipGetLastError());
    note_launches(RNVP_PROFILE_TRAIN, grid > kSeg ? 4 : 3);       // pack, hot kernel, (segment sums,) scatter (+ Adam)
    return RNVP_OK;
}

int loss_grad(hipStream_t st, const KShape &k, const float *params, const float *x, const float *c,
              const int64_t *row_index, int64_t n, float inv_B, float *grad_out, float *loss_out,
              void *ws, size_t ws_bytes, Seeds sd) {
    return loss_grad_impl(st, k, params, x, c, row_index, n, inv_B, grad_out, loss_out, ws, ws_bytes, nullptr,
                          nullptr, nullptr, AdamK{}, sd);
}

// loss + gradient + Adam with the optimizer fused into the final scatter kernel (one launch and one
// round trip of the gradient fewer than rnvp_loss_grad + rnvp_adam_step; same arithmetic, bit for bit)
int train_step(hipStream_t st, const KShape &k, float *params, const float *x, const float *c,
               const int64_t *row_index, int64_t n, float inv_B, float *grad_buf, float *loss_out,
               float *exp_avg, float *exp_avg_sq, const AdamK &adam, void *ws, size_t ws_bytes) {
    return loss_grad_impl(st, k, params, x, c, row_index, n, inv_B, grad_buf, loss_out, ws, ws_bytes, params, exp_avg,
                          exp_avg_sq, adam);
}

}  // namespace mfma
}  // namespace rnvp
