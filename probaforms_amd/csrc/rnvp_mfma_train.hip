// rnvp_mfma_train.hip -- host side of the fused forward + backward step on f32 MFMA (gfx950): workspace plan, dispatch over
// the tile geometries (device code and launch templates: rnvp_mfma_train_dev.h, instantiated per geometry in
// rnvp_mfma_train_nf{2,4,8}.hip) and the second stage that turns per-workgroup partial gradients into the flat
// reference-order gradient (+ Adam).  Replaces `loss = -nf.log_prob(X, C); loss.backward()` and `opt.step()`
// (/root/reference/probaforms/models/realnvp.py:246-251).
#include "rnvp_mfma_train_dev.h"
#include "rnvp_mfma_pack.h"

namespace rnvp {
namespace mfma {

#define RNVP_LAUNCH_DECL(nf, cq)                                                                                          \
    int launch_train_##nf##_##cq(hipStream_t st, const KShape &k, const Geo &g, const TrainPlan &pl, const float *packed,  \
                                 const float *x, const float *c, const int64_t *row_index, int64_t n, float inv_B,         \
                                 float *gpart, float *losspart, float *scratch, int *grid_out, Seeds sd, PartialLayout *lay);
RNVP_LAUNCH_DECL(2, 1) RNVP_LAUNCH_DECL(2, 0) RNVP_LAUNCH_DECL(4, 2) RNVP_LAUNCH_DECL(8, 4)
#undef RNVP_LAUNCH_DECL

namespace {

// ---- stage 2: per-workgroup partial gradients -> flat reference-order gradient (+ Adam, + next step's fragments) --------
// Where the gradient record of one hidden tile (layer of parity pc, net `net`) keeps a parameter: float offset inside the
// tile's record, or -1 when the masks make it dead (exactly zero, as in the reference).  i: hidden unit inside the tile.
// The record: NTI tiles of 256 (dW1 | db1: lane (q', jn & 15) x 4 hidden units), then dW2 -- OTL tiles of 256, or the compact
// 128 floats of Dims::w2c.  NF and CQ are powers of two: shifts, no divisions.
struct RecMap {
    int NF, CQ, OTL, NTI, KSP, sh2, shc, w2c, d, c;
    __device__ RecMap(const KShape &k, const Geo &g, int NTI_, int w2c_)
        : NF(g.NF), CQ(g.CQ), OTL(g.OTL), NTI(NTI_), KSP(4 * NTI_), sh2(31 - __clz(2 * g.NF)), shc(g.CQ > 0 ? 31 - __clz(g.CQ) : 0),
          w2c(w2c_), d(k.d), c(k.c) {}
    __device__ int slot1(int i, int jn) const { return (jn >> 4) * 256 + (16 * (i >> 2) + (jn & 15)) * 4 + (i & 3); }
    __device__ int w1(int pc, int i, int col) const {              // W1[16 ht + i][col]
        if (col < d) {
            const int qq = col >> sh2, e = col & (2 * NF - 1);
            return (e & 1) == pc ? slot1(i, qq * KSP + (e >> 1)) : -1;
        }
        if (CQ == 0) return -1;
        const int ci = col - d;
        return slot1(i, (ci >> shc) * KSP + NF + (ci & (CQ - 1)));
    }
    __device__ int b1(int i) const { return slot1(i, NF + CQ); }        // the ones column
    __device__ int w2(int pc, int net, int i, int j) const {       // W2[j][16 ht + i]
        const int qo = j >> sh2, e = j & (2 * NF - 1);
        if ((e & 1) != 1 - pc) return -1;
        const int f = e >> 1;
        if (w2c) {      // column 2 qo + f of this net: block cb = col >> 2, kept by lane group 2 cb + (i >> 1) & 1 (layer_bwd)
            const int col = 2 * qo + f, lane = 16 * (2 * (col >> 2) + ((i >> 1) & 1)) + 4 * (i >> 2) + (col & 3);
            return NTI * 256 + lane * 2 + (i & 1);
        }
        const int otl = NF >= 4 ? f >> 2 : 0, io = NF >= 4 ? 4 * qo + (f & 3) : 4 * qo + 2 * net + f;
        return (NTI + otl) * 256 + (16 * (i >> 2) + io) * 4 + (i & 3);
    }
    __device__ int b2(int pc, int net, int j) const {              // offset inside the layer's db2 record [ot][q][4]
        const int qo = j >> sh2, e = j & (2 * NF - 1);
        if ((e & 1) != 1 - pc) return -1;
        const int f = e >> 1;
        const int ot = NF >= 4 ? net * OTL + (f >> 2) : 0, reg = NF >= 4 ? f & 3 : 2 * net + f;
        return (ot * 4 + qo) * 4 + reg;
    }
};

// parameter source of pack_slot for ONE hidden tile of one net, from the workgroup's LDS copy of the freshly updated
// values: W1 half [16 hidden units][nin] then 16 of b1; W2 half [d][16 hidden units]; bias-2 workgroups [2 nets][d]
struct TileParams {
    const float *pw;
    int nin, ht, d;
    __device__ float w1(int, int hid, int col) const { return pw[(hid - 16 * ht) * nin + col]; }
    __device__ float b1(int, int hid) const { return pw[16 * nin + hid - 16 * ht]; }
    __device__ float w2(int, int feat, int hid) const { return pw[feat * 16 + hid - 16 * ht]; }
    __device__ float b2(int net, int feat) const { return pw[net * d + feat]; }
};

// ONE launch behind the training kernel (round 3 ran three: segment sums, scatter + Adam, and the next step's re-pack).
// Every gradient record (layer, net, hidden tile) is served by TWO workgroups: half 0 owns rows 16 ht .. 16 ht + 15 of W1 and b1
// (the record's dW1 | db1 tiles), half 1 the matching columns of W2 (its dW2 part); L more workgroups own the second-Linear biases
// of one layer each; the last one adds the loss partials.  Stages, each optional (`mode`):
//   kFinSum : add the workgroup's part of the record over the G per-workgroup partials of the training launch -- threads split into
//             sub-groups, sub-group s adds partials s, s + nsub, ... in index order (eight loads in flight), the sub-sums are added
//             in s order: a fixed order for a given launch geometry, no float atomics -- and scatter it into the flat gradient;
//             without it the flat gradient is an INPUT (data parallel: the all-reduced message);
//   kFinAdam: torch.optim.Adam on the workgroup's parameters (realnvp.py:251);
//   kFinPack: rewrite the packed fragments that hold exactly these parameters from the updated values (kept in LDS), so that the
//             next batch's training kernel starts without a pack launch.
constexpr int kFinSum = 1, kFinAdam = 2, kFinPack = 4;
constexpr int kFinThreads = 512;
constexpr int kFinRecMax = 1024;                  // floats of the largest half record: NTI 4 x 256 (d = 64, cdim = 16)
constexpr int kFinParMax = 16 * 81;               // LDS copy of a workgroup's parameters: 16 x (d + cdim + 1) <= 16 x 81, d x 16 <= 1024
constexpr int kFinPer = 3;                        // parameters per thread: ceil(16 x 81 / 512)

#ifndef RNVP_FIN_INFLIGHT
#define RNVP_FIN_INFLIGHT 8
#endif
constexpr int kFinInFlight = RNVP_FIN_INFLIGHT;
__global__ void __launch_bounds__(kFinThreads)
k_train_finish(KShape k, Geo g, int NTI, int glayer_floats, int w2c, int mode, int l0, int nl, const float *__restrict__ gpart, int G,
               const float *__restrict__ losspart, int n_loss, float inv_B, const float *loss_in, float *loss_out,
               float *grad, float *params, float *adam_m, float *adam_v, AdamK adam, float *packed, const int *err) {
    __shared__ __attribute__((aligned(16))) float rec[kFinRecMax];
    __shared__ f4 red[kFinThreads];
    __shared__ float pw[kFinParMax];
    const int t = threadIdx.x, b = blockIdx.x;
    // the launch serves the layers l0 .. l0 + nl - 1 (all of them, or one chunk of the data-parallel step: rnvp_dp.hip); the loss
    // block exists where loss_out is given
    const int HT = g.HT, nrec = nl * 2 * HT;
    // a wave of the training launch gave up a bounded wait (spin_nap): its workgroup's partial is incomplete.  No Adam step, no
    // re-pack -- the parameters keep their last good values -- and the loss says so (kProtocolNaN)
    // (data parallel: ANY rank's error reaches every rank through the all-reduced loss -- NaN + x keeps the NaN's payload)
    const bool bad = (err && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) ||
                     (!(mode & kFinSum) && loss_in && __float_as_uint(loss_in[0]) == kProtocolNaN);
    if (bad) mode &= ~(kFinAdam | kFinPack);
    if (b == 2 * nrec + nl) {               // the loss: partials added in a fixed order (one wave), or read out of the message
        if (t < 64 && loss_out) {
            if (mode & kFinSum) {
                float a = 0.f;
                for (int i = t; i < n_loss; i += 64) a += losspart[i];
                for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off);
                if (t == 0) loss_out[0] = bad ? __uint_as_float(kProtocolNaN) : -a * inv_B;
            } else if (t == 0 && loss_in) {
                loss_out[0] = bad ? __uint_as_float(kProtocolNaN) : loss_in[0];
            }
        }
        return;
    }
    const bool bias2 = b >= 2 * nrec;
    const int r = b >> 1, half = bias2 ? 2 : b & 1;           // 0: W1 | b1, 1: W2, 2: the layer's b2
    const int l = l0 + (bias2 ? b - 2 * nrec : r / (2 * HT));
    const int net = bias2 ? 0 : (r / HT) & 1, ht = bias2 ? 0 : r % HT;
    const int tblk = w2c ? NTI * 256 + 128 : (NTI + g.OTL) * 256, netblock = HT * tblk;
    // this workgroup's floats inside the layer's record
    const int rec_off = bias2 ? 2 * netblock : net * netblock + ht * tblk + (half ? NTI * 256 : 0);
    const int rec_n = bias2 ? g.NT2 * 16 : (half ? tblk - NTI * 256 : NTI * 256);
    const int pc = (l + k.alt) & 1;
    const int h = k.nout[0], nin = k.d + k.c, d = k.d;
    // the workgroup's parameters: gradient out of the record (or in from the all-reduced message), Adam, LDS copy for the re-pack.
    // Up to kFinPer per thread; their loads (parameter, both moments) are issued BEFORE the partial sums are read, so that the
    // two memory latencies of a small step overlap.
    const RecMap map(k, g, NTI, w2c);
    const int npar = half == 2 ? 2 * d : (half == 1 ? 16 * d : 16 * (nin + 1));
    const int base_in_rec = half == 1 ? NTI * 256 : 0;      // rec[] starts at this workgroup's part
    size_t pidx[kFinPer];
    float ga[kFinPer], pv[kFinPer], mm[kFinPer], vv[kFinPer];
    int locs[kFinPer];
    bool live[kFinPer];
#pragma unroll
    for (int u = 0; u < kFinPer; ++u) {
        const int e = t + u * kFinThreads;
        live[u] = false; ga[u] = 0.f; pv[u] = 0.f; mm[u] = 0.f; vv[u] = 0.f; pidx[u] = 0; locs[u] = -1;
        if (e >= npar) continue;
        int pnet = net, idx, i = 0, loc;
        if (half == 2) {
            pnet = e >= d;
            const int j = e - pnet * d;
            idx = k.boff[1] + j;
            loc = map.b2(pc, pnet, j);
        } else if (half == 1) {
            const int j = e >> 4;
            i = e & 15;
            idx = k.woff[1] + j * h + 16 * ht + i;
            loc = map.w2(pc, net, i, j);
        } else if (e < 16 * nin) {
            i = e / nin;
            const int col = e - i * nin;
            idx = k.woff[0] + (16 * ht + i) * nin + col;
            loc = map.w1(pc, i, col);
        } else {
            i = e - 16 * nin;
            idx = k.boff[0] + 16 * ht + i;
            loc = map.b1(i);
        }
        if (16 * ht + i >= h) continue;                     // a padded hidden unit: no parameter behind it
        live[u] = true;
        pidx[u] = ((size_t)l * 2 + pnet) * k.npn + idx;
        locs[u] = loc;
        if (!(mode & kFinSum)) ga[u] = grad[pidx[u]];
        if (mode & (kFinAdam | kFinPack)) pv[u] = params[pidx[u]];
        if (mode & kFinAdam) { mm[u] = adam_m[pidx[u]]; vv[u] = adam_v[pidx[u]]; }
    }
    if (mode & kFinSum) {
        const int nf4 = rec_n / 4;                          // <= 256
        const size_t stride4 = (size_t)glayer_floats * k.L / 4;
        const int nsub = kFinThreads / nf4;
        const int col = t % nf4, sub = t / nf4;
        f4 acc = f4{0.f, 0.f, 0.f, 0.f};
        if (sub < nsub) {
            const f4 *src = reinterpret_cast<const f4 *>(gpart + (size_t)l * glayer_floats + rec_off) + col;
            int bb = sub;
            // kFinInFlight loads per thread before the first add: the partials come out of L2 / MALL / HBM at 1-2 us a round trip
            for (; bb + (kFinInFlight - 1) * nsub < G; bb += kFinInFlight * nsub) {
                f4 v[kFinInFlight];
#pragma unroll
                for (int u = 0; u < kFinInFlight; ++u) v[u] = __builtin_nontemporal_load(src + (size_t)(bb + u * nsub) * stride4);
#pragma unroll
                for (int u = 0; u < kFinInFlight; ++u) acc += v[u];
            }
            for (; bb < G; bb += nsub) acc += __builtin_nontemporal_load(src + (size_t)bb * stride4);
        }
        red[t] = acc;
        __syncthreads();
        if (t < nf4) {
            f4 a = red[t];
            for (int s2 = 1; s2 < nsub; ++s2) a += red[s2 * nf4 + t];
            *reinterpret_cast<f4 *>(rec + 4 * t) = a;
        }
        __syncthreads();
    }
    if (mode & kFinSum) {
#pragma unroll
        for (int u = 0; u < kFinPer; ++u) ga[u] = (live[u] && locs[u] >= 0) ? rec[locs[u] - base_in_rec] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < kFinPer; ++u) {
        const int e = t + u * kFinThreads;
        if (e >= npar) continue;
        if (live[u]) {
            if (mode & kFinSum) grad[pidx[u]] = ga[u];
            if (mode & kFinAdam) {
                adam_one(pv[u], ga[u], mm[u], vv[u], adam);
                params[pidx[u]] = pv[u]; adam_m[pidx[u]] = mm[u]; adam_v[pidx[u]] = vv[u];
            }
        }
        pw[e] = pv[u];
    }
    if (!(mode & kFinPack)) return;
    __syncthreads();
    float *pk = packed + (size_t)l * g.layer_floats;
    const TileParams src{pw, nin, ht, d};
    if (half == 2) {
        for (int s2 = t; s2 < g.NT2 * 16; s2 += kFinThreads) pk[g.oB2 + s2] = pack_slot(k, g, pc, kPkB2, 0, 0, s2, src);
        return;
    }
    const int T = net * HT + ht;
    // the packed arrays that hold this workgroup's parameters, the slots of tile T of each
    const int arrs[2][5] = {{kPkA1, kPkB1, kPkA1T, kPkA1X, kPkA1S}, {kPkA2, kPkA2T, kPkA2X, kPkA2TS, -1}};
#pragma unroll
    for (int a = 0; a < 5; ++a) {
        const int arr = arrs[half][a];
        if (arr < 0) continue;
        const int per = pack_per_tile(g, arr);
        float *dst = pk + pack_offset(g, arr) + T * per;
        for (int s2 = t; s2 < per; s2 += kFinThreads) dst[s2] = pack_slot(k, g, pc, arr, net, ht, s2, src);
    }
}

template <int NF, int CQ>
TrainPlan make_plan(const Geo &g, int L) {
    using DM = Dims<NF, CQ>;
    TrainPlan p;
    const int netblock = g.HT * (DM::NTI + DM::OTL) * 256;
    p.glayer_floats = 2 * netblock + DM::NT2 * 16;
    p.RMAX = TrainRows<NF, CQ>::value;
    p.lds_bytes = ((size_t)kWaves * DM::SLOT + (size_t)kWaves * DM::template tb<TrainRows<NF, CQ>::value>()) * sizeof(float) + kSyncBytes;
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
    const Geo g = make_geo(k.d, k.c, k.nout[0], kTrainSplit);
    TrainPlan pl;
    if (!plan_for(g, k.L, &pl)) return 0;
    size_t b = align_up((size_t)g.layer_floats * k.L * sizeof(float), 256);                    // packed weights
    b += align_up((size_t)kMaxGridTrain * pl.glayer_floats * k.L * sizeof(float), 256);          // partials
    b += align_up((size_t)kMaxGridTrain * kWaves * sizeof(float), 256) + 256;                    // loss partials, the error word
    b += align_up((size_t)kMaxGridTrain * kWaves * pl.scratch_per_wave * sizeof(float), 256);    // saved activations
    return b;
}

static int launch_finish(hipStream_t st, const KShape &k, const Geo &g, int glayer_floats, int w2c, int mode, const float *gpart,
                         int G, const float *losspart, float inv_B, const float *loss_in, float *loss_out, float *grad,
                         float *params, float *adam_m, float *adam_v, const AdamK &adam, float *packed, const int *err,
                         int l0 = 0, int nl = -1) {
    const int NTI = (g.KS1 + 1 + 3) / 4;
    if (nl < 0) nl = k.L - l0;
    const unsigned blocks = (unsigned)(2 * nl * 2 * g.HT + nl + 1);         // two per gradient record, one per layer (b2), the loss
    hipLaunchKernelGGL(k_train_finish, dim3(blocks), dim3(kFinThreads), 0, st, k, g, NTI, glayer_floats, w2c, mode, l0, nl, gpart, G, losspart,
                       G * kWaves, inv_B, loss_in, loss_out, grad, params, adam_m, adam_v, adam, packed, err);
    RNVP_HIP_TRY(hipGetLastError());
    return RNVP_OK;
}

// One training step on the register-chained kernels: [pack the fragments] -> fused forward + backward (per-workgroup partial
// gradients) -> k_train_finish (sum, scatter to flat order [, Adam [, re-pack for the next step]]).
//   packed_valid: the workspace already holds the fragments of `params` (the previous step of the same rnvp_fit_epoch* call
//                 re-packed them): no pack launch;   pack_next: this step's finish kernel re-packs what Adam updates.
static int step_impl(hipStream_t st, const KShape &k, const float *params, const float *x, const float *c,
                     const int64_t *row_index, int64_t n, float inv_B, float *grad_out, float *loss_out,
                     void *ws, size_t ws_bytes, float *adam_p, float *adam_m, float *adam_v, AdamK adam, Seeds sd,
                     bool packed_valid, bool pack_next, PendingPartials *pending = nullptr) {
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
    float *losspart = reinterpret_cast<float *>(w);
    w += align_up((size_t)kMaxGridTrain * kWaves * sizeof(float), 256) + 256;
    float *scratch = reinterpret_cast<float *>(w);
    int rc = packed_valid ? RNVP_OK : pack_weights(st, k, g, params, packed, error_word(losspart));
    if (rc) return rc;
    int grid = 0;
    PartialLayout lay{pl.glayer_floats, 0};
    if (g.NF == 2 && g.CQ == 1) rc = launch_train_2_1(st, k, g, pl, packed, x, c, row_index, n, inv_B, gpart, losspart, scratch, &grid, sd, &lay);
    else if (g.NF == 2 && g.CQ == 0) rc = launch_train_2_0(st, k, g, pl, packed, x, c, row_index, n, inv_B, gpart, losspart, scratch, &grid, sd, &lay);
    else if (g.NF == 4 && g.CQ == 2) rc = launch_train_4_2(st, k, g, pl, packed, x, c, row_index, n, inv_B, gpart, losspart, scratch, &grid, sd, &lay);
    else if (g.NF == 8 && g.CQ == 4) rc = launch_train_8_4(st, k, g, pl, packed, x, c, row_index, n, inv_B, gpart, losspart, scratch, &grid, sd, &lay);
    else return RNVP_EUNSUPPORTED;
    if (rc) return rc;
    if (pending) {      // the caller sums the partials itself, in chunks of layers (finish_sum_layers): data-parallel step
        pending->glayer_floats = lay.glayer_floats; pending->w2c = lay.w2c; pending->grid = grid; pending->inv_B = inv_B;
        note_launches(RNVP_PROFILE_TRAIN, packed_valid ? 1 : 2);
        return RNVP_OK;
    }
    const int mode = kFinSum | (adam_p ? kFinAdam : 0) | (adam_p && pack_next ? kFinPack : 0);
    rc = launch_finish(st, k, g, lay.glayer_floats, lay.w2c, mode, gpart, grid, losspart, inv_B, nullptr, loss_out, grad_out, adam_p,
                       adam_m, adam_v, adam, packed, error_word(losspart));
    if (rc) return rc;
    note_launches(RNVP_PROFILE_TRAIN, packed_valid ? 2 : 3);       // [pack,] hot kernel, finish
    return RNVP_OK;
}

// the workspace regions of step_impl (one place for the chunked data-parallel calls below)
struct WsView { float *packed, *gpart, *losspart; };
static bool ws_view(const KShape &k, void *ws, size_t ws_bytes, Geo *g, TrainPlan *pl, WsView *v) {
    if (!ws || ws_bytes < train_workspace_bytes(k, 1)) return false;
    *g = make_geo(k.d, k.c, k.nout[0], kTrainSplit);
    if (!plan_for(*g, k.L, pl)) return false;
    char *w = static_cast<char *>(ws);
    v->packed = reinterpret_cast<float *>(w);
    w += align_up((size_t)g->layer_floats * k.L * sizeof(float), 256);
    v->gpart = reinterpret_cast<float *>(w);
    w += align_up((size_t)kMaxGridTrain * pl->glayer_floats * k.L * sizeof(float), 256);
    v->losspart = reinterpret_cast<float *>(w);
    return true;
}

// DATA-PARALLEL step in chunks of layers (rnvp_dp.hip): the training launch alone ...
int loss_partials(hipStream_t st, const KShape &k, const float *params, const float *x, const float *c, const int64_t *row_index,
                  int64_t n, float inv_B, void *ws, size_t ws_bytes, bool packed_valid, PendingPartials *pending) {
    return step_impl(st, k, params, x, c, row_index, n, inv_B, nullptr, nullptr, ws, ws_bytes, nullptr, nullptr, nullptr, AdamK{}, Seeds{},
                     packed_valid, false, pending);
}
// ... then, per chunk of layers [l0, l0 + nl): the partial sums into the flat gradient (the batch loss with the chunk that is
// handed loss_out) -- the same per-record arithmetic as the one-launch finish, bit for bit ...
int finish_sum_layers(hipStream_t st, const KShape &k, const PendingPartials &p, int l0, int nl, float *grad, float *loss_out,
                      void *ws, size_t ws_bytes) {
    Geo g; TrainPlan pl; WsView v;
    if (!ws_view(k, ws, ws_bytes, &g, &pl, &v)) return RNVP_EWORKSPACE;
    return launch_finish(st, k, g, p.glayer_floats, p.w2c, kFinSum, v.gpart, p.grid, v.losspart, p.inv_B, nullptr, loss_out, grad, nullptr,
                         nullptr, nullptr, AdamK{}, v.packed, error_word(v.losspart), l0, nl);
}
// ... and, behind the chunk's all-reduce, Adam + re-pack of exactly those layers (loss_in: the all-reduced batch loss, read by
// every chunk for the cross-rank error check, written to loss_out by the chunk that is handed one)
int adam_pack_layers(hipStream_t st, const KShape &k, float *params, float *grad, const float *loss_in, float *loss_out,
                     float *exp_avg, float *exp_avg_sq, const AdamK &adam, void *ws, size_t ws_bytes, int l0, int nl) {
    Geo g; TrainPlan pl; WsView v;
    if (!ws_view(k, ws, ws_bytes, &g, &pl, &v)) return RNVP_EWORKSPACE;
    return launch_finish(st, k, g, pl.glayer_floats, 0, kFinAdam | kFinPack, nullptr, 0, nullptr, 0.f, loss_in, loss_out, grad, params,
                         exp_avg, exp_avg_sq, adam, v.packed, error_word(v.losspart), l0, nl);
}

int loss_grad(hipStream_t st, const KShape &k, const float *params, const float *x, const float *c,
              const int64_t *row_index, int64_t n, float inv_B, float *grad_out, float *loss_out,
              void *ws, size_t ws_bytes, Seeds sd, bool packed_valid) {
    return step_impl(st, k, params, x, c, row_index, n, inv_B, grad_out, loss_out, ws, ws_bytes, nullptr, nullptr, nullptr,
                     AdamK{}, sd, packed_valid, false);
}

// loss + gradient + Adam with the optimizer (and the next step's re-pack) fused into the finish kernel: same arithmetic as
// rnvp_loss_grad + rnvp_adam_step, bit for bit
int train_step(hipStream_t st, const KShape &k, float *params, const float *x, const float *c,
               const int64_t *row_index, int64_t n, float inv_B, float *grad_buf, float *loss_out,
               float *exp_avg, float *exp_avg_sq, const AdamK &adam, void *ws, size_t ws_bytes, bool packed_valid,
               bool pack_next) {
    return step_impl(st, k, params, x, c, row_index, n, inv_B, grad_buf, loss_out, ws, ws_bytes, params, exp_avg,
                     exp_avg_sq, adam, Seeds{}, packed_valid, pack_next);
}

// second half of a DATA-PARALLEL step (after the all-reduce of [gradient | loss]): the batch loss out of the message, Adam
// from the all-reduced flat gradient, and the re-pack for the next batch -- one launch (what rnvp_dp_finish_step + the next
// step's pack launch did)
int adam_pack(hipStream_t st, const KShape &k, float *params, float *grad, const float *loss_in, float *loss_out,
              float *exp_avg, float *exp_avg_sq, const AdamK &adam, void *ws, size_t ws_bytes) {
    if (!ws || ws_bytes < train_workspace_bytes(k, 1)) return RNVP_EWORKSPACE;
    const Geo g = make_geo(k.d, k.c, k.nout[0], kTrainSplit);
    TrainPlan pl;
    if (!plan_for(g, k.L, &pl)) return RNVP_EUNSUPPORTED;
    // the workspace layout of step_impl: packed fragments, partials, loss partials (+ the error word this rank's training launch raises)
    char *w = static_cast<char *>(ws) + align_up((size_t)g.layer_floats * k.L * sizeof(float), 256) +
              align_up((size_t)kMaxGridTrain * pl.glayer_floats * k.L * sizeof(float), 256);
    return launch_finish(st, k, g, pl.glayer_floats, 0, kFinAdam | kFinPack, nullptr, 0, nullptr, 0.f, loss_in, loss_out, grad, params,
                         exp_avg, exp_avg_sq, adam, static_cast<float *>(ws), error_word(reinterpret_cast<float *>(w)));
}

}  // namespace mfma
}  // namespace rnvp
