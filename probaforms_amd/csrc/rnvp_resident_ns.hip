// rnvp_resident_ns.hip -- resident fit (rnvp_resident.hip), net-split form for batches of at most 64 rows (the reference's default: 32).
//
// A resident step is a chain of small dependent GEMMs, and at batch 32 the one-wave-per-tile kernel leaves most of a CU
// idle.  Here a step is cut along the two lines the algorithm offers (8 waves; described for up to two row tiles):
//   * the two nets of a coupling layer are independent: every 16-row tile gets a PAIR of waves, wave 2t walks the t nets
//     of tile t, wave 2t+1 its s nets; both keep the tile's x, condition and gradient.  Per layer the pair meets twice
//     through LDS -- the forward exchanges the nets' outputs (both waves then apply the coupling), the backward the nets'
//     input gradients (both then update the gradient at the layer input).  The exchange buffers alternate by parity, so one
//     workgroup barrier per meeting is enough; what does not depend on the partner (the next layer's weights, its saved
//     activations, e^s, tanh') is placed between posting and the barrier;
//   * the weight gradients feed nothing inside a step: the chain only LEAVES their operands -- net input, output gradient,
//     hidden activations, pre-activation gradient -- as transposed tiles in LDS (two per-layer records, used in turn), and
//     waves 4-7 contract them over the rows one layer behind the chain, between the same barriers: each takes one net and
//     one of its two contractions over BOTH row tiles (summed inside the MFMA accumulators, tile 0 first), so there is one
//     gradient stage however many tiles a batch has.
// Batches of 33 to 64 rows (three or four row tiles) use all eight waves for the chain; every wave then takes one
// weight-gradient job (net, contraction, pair of tiles) behind its own posting, and the two pairs' stages are added in the
// Adam phase.
// The Adam scalars of a step (double precision division and square root) are computed by one of those waves while the
// chain runs its forward; all eight share the Adam phase.  Products with padding (the fourth slot of a hidden tile of at
// most 12 units, the second k-step of an input of at most 3 columns) are exact zeros and are not issued.
// Same arithmetic per element as the one-wave form; the two input gradients are added in the same order (s + t) by both
// waves of a pair.  Measured (scripts/resident_time.py, profiles/): 10.8 us per step of the defaults, 15.1 in the one-wave form.
#include "rnvp_resident_dev.h"

namespace rnvp {
namespace resident {
namespace {

constexpr int kNsWaves = 8, kNsThreads = 64 * kNsWaves;      // waves 0-3: (tile, net) chains; 4-7: weight gradients; all: Adam
constexpr int kNsSlots = 2;                                  // the backward's per-layer records: the chain fills one, the helpers read the other

struct NsPlan {
    int W, P, mv_lds;                                  // row tiles, parameters, moments in LDS
    int stg_net, stg_floats;
    int oPAR, oM, oV, oSTG, oDB2, oRED, oADK, oEX, oFW, oBW, oT2;     // float offsets
    int total_floats;
};

// floats of the forward's per-(tile, layer) record and of the backward's per-(tile, slot) record
constexpr int ns_fw_layer(int mt, int dt) { return 2 * 64 * dt + 2 * mt * 16 * TS; }
constexpr int ns_bw_slot(int mt, int kit, int dt) { return 4 * (kit < 4 ? kit : 4) * TS + 2 * 4 * dt * TS + 2 * mt * 16 * TS; }

// MT hidden tiles, KIT k-steps of the net input, DT slots of x that hold features
template <int MT, int KIT, int ACT, int DT>
__global__ void __launch_bounds__(kNsThreads)
k_fit_resident_ns(KShape s, NsPlan pl, float *__restrict__ params, const uint8_t *__restrict__ masks, const float *__restrict__ x,
                  const float *__restrict__ c, const int64_t *__restrict__ perm, int64_t n, int64_t batch, int64_t n_epochs,
                  float *__restrict__ loss_hist, float *__restrict__ exp_avg, float *__restrict__ exp_avg_sq, double lr, double beta1,
                  double beta2, double eps, double wd, double b1t, double b2t) {
    constexpr int NIT = KIT > 4 ? 2 : 1;
    constexpr int KE = KIT < 4 ? KIT : 4;              // slots of the first input tile that hold inputs (incl. the ones element)
    constexpr int KXT = KIT < DT ? KIT : DT;
    constexpr int RIN = 4 * KE, RGO = 4 * DT;          // rows (elements) kept of the input tile / the output-gradient tile
    constexpr int XF = 64 * DT, HT = 16 * TS;
    constexpr int FW_LAYER = ns_fw_layer(MT, DT);      // [x | s | H of the t net | H of the s net]
    constexpr int BW_SLOT = ns_bw_slot(MT, KIT, DT);   // [input tile | g_out t, s | g_pre t, s]
    constexpr int oGO = RIN * TS, oGP = oGO + 2 * RGO * TS;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, nthreads = blockDim.x, lane = tid & 63, q = lane >> 4, r = lane & 15, i = r;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // known uniform: scalar branches and address arithmetic
    const int tile = wave >> 1, role = wave & 1;          // role 0: the t net, 1: the s net
    const int d = s.d, cd = s.c, L = s.L, P = pl.P, h = s.nout[0], nin0 = d + cd, npn = s.npn;
    const int w0 = s.woff[0], b0 = s.boff[0], w1 = s.woff[1], b1o = s.boff[1];
    const int pi = 4 * (i & 3) + (i >> 2);
    // wave-uniform trims: products whose one factor is padding everywhere are exact zeros -- not issued at all
    const bool k1 = KIT == 2 && nin0 + 1 <= 4;             // the net input (and its ones element) fits the first k-step
    const bool h3 = MT == 1 && h <= 12;                    // the fourth slot of the hidden tile (units 12-15) is padding
    float *PAR = lds + pl.oPAR, *MM = lds + pl.oM, *VV = lds + pl.oV, *RED = lds + pl.oRED;
    AdamK *ADK = reinterpret_cast<AdamK *>(lds + pl.oADK);
    const int ctile = tile < pl.W ? tile : 0;             // waves without a row tile never touch the per-tile areas of the chain
    float *FWt = lds + pl.oFW + (size_t)ctile * L * FW_LAYER, *BWt = lds + pl.oBW + (size_t)ctile * kNsSlots * BW_SLOT;
    f4 *EX = reinterpret_cast<f4 *>(lds + pl.oEX) + (size_t)ctile * 256;               // [parity][role][lane]
    for (int e = tid; e < pl.total_floats; e += nthreads) lds[e] = 0.f;
    __syncthreads();
    for (int p = tid; p < P; p += nthreads) {
        PAR[p] = params[p];
        if (pl.mv_lds) { MM[p] = exp_avg[p]; VV[p] = exp_avg_sq[p]; }
    }
    __syncthreads();
    const float prior_c = 0.5f * (float)d * kLog2Pi;
    // perm holds n_epochs permutations of n rows back to back; every epoch is cut into the same batches (the last one ragged).
    // Batch cursors advance by increments: no 64-bit division inside the loop
    const int64_t nb_e = (n + batch - 1) / batch, nb = nb_e * n_epochs;
    struct Cursor { int64_t ep, k; };
    auto advance = [&](Cursor &cu) { if (++cu.k == nb_e) { cu.k = 0; ++cu.ep; } };
    auto rows_at = [&](const Cursor &cu) -> int64_t { return (n - cu.k * batch < batch) ? n - cu.k * batch : batch; };

    // ---- per-lane constants (as rnvp_resident.hip) ----
    const int gW1 = w0 + pi * nin0 + q, gW2 = w1 + pi * h + q, gW1t = w0 + q * nin0 + pi, gB1 = b0 + q, gB2 = b1o + q;
    int gW2t[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) gW2t[e] = w1 + (4 * e + q) * h + pi;
    f4 hm[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) hm[m][e] = 16 * m + 4 * e + q < h ? 1.f : 0.f;
    uint64_t mbits = 0;
    for (int l = 0; l < L; ++l)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = 4 * e + q;
            const bool mk = masks ? (j < d && masks[l * d + j] != 0) : (((j + l + (s.alt == 2 ? 1 : 0)) & 1) != 0);   // (no read past the [L, d] table)
            if (j >= d || mk) mbits |= 1ull << (4 * l + e);
        }
    bool xok[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) xok[e] = 4 * e + q < d;
    int sS2[MT][4], sS1[MT][NIT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int fo = 4 * q + e, hid_n = 16 * m + i, hid_m = 16 * m + 4 * q + e;
            sS2[m][e] = (fo < d && hid_n < h) ? w1 + fo * h + hid_n : npn + (lane & 15);
#pragma unroll
            for (int nt = 0; nt < NIT; ++nt) {
                const int j = 16 * nt + i;
                sS1[m][nt][e] = hid_m >= h ? npn + (lane & 15) : (j < nin0 ? w0 + hid_m * nin0 + j : (j == nin0 ? b0 + hid_m : npn + (lane & 15)));
            }
        }
    // transposed tiles: this lane's elements 4e + q of row r go to [element][TS] + r; a contraction over the rows reads
    // element i (clamped to the rows a compact tile keeps: what the clamp duplicates lands in the stage's dump zone)
    const int tp = q * TS + r;                            // + 4 e TS
    const int tgI = (i < RIN ? i : RIN - 1) * TS + q, tgG = (i < RGO ? i : RGO - 1) * TS + q, tgH = i * TS + q;   // + 4 ks

    auto row_of = [&](const Cursor &cu) -> int64_t {      // source row of this lane's row in the cursor's batch (-1: none)
        if (cu.ep >= n_epochs) return -1;
        const int64_t rr = (int64_t)tile * 16 + r;
        return rr < rows_at(cu) ? perm[cu.ep * n + cu.k * batch + rr] : -1;
    };
    auto load_rows = [&](int64_t src, f4 &xo, f4 (&co)[NIT]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) xo[e] = (src >= 0 && xok[e]) ? x[src * d + 4 * e + q] : 0.f;
#pragma unroll
        for (int nt = 0; nt < NIT; ++nt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = 16 * nt + 4 * e + q;
                co[nt][e] = (src >= 0 && j >= d && j < nin0) ? c[src * cd + (j - d)] : 0.f;
            }
    };
    // A fragments of this wave's net of one layer: forward (biases, W1, W2) and backward (W2^T for g_h, W1^T for g_in)
    struct FwdW { float b1[MT][4], a1[MT][KIT], b2[4], a2[MT][4]; };
    struct BwdW { float a2t[MT][KXT], a1t[MT][4]; };
    auto load_fwd = [&](int l, FwdW &w) {
        const float *pn = PAR + (size_t)(2 * l + role) * npn;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { w.b1[m][e] = pn[gB1 + 16 * m + 4 * e]; w.a2[m][e] = pn[gW2 + 16 * m + 4 * e]; }
#pragma unroll
            for (int k = 0; k < KIT; ++k) w.a1[m][k] = pn[gW1 + m * 16 * nin0 + 4 * k];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) w.b2[e] = pn[gB2 + 4 * e];
    };
    auto load_bwd = [&](int l, BwdW &w) {
        const float *pn = PAR + (size_t)(2 * l + role) * npn;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int e = 0; e < KXT; ++e) w.a2t[m][e] = pn[gW2t[e] + 16 * m];
#pragma unroll
            for (int e = 0; e < 4; ++e) w.a1t[m][e] = pn[gW1t + (16 * m + 4 * e) * nin0];
        }
    };

    // weight gradients of one net of one layer, summed over the batch's row tiles inside the MFMA accumulators (tile 0's rows
    // first: a fixed order) and written to the ONE stage: d W2 (+ d b2, the two tiles' DPP row sums added) and / or d W1 (+ d b1)
    auto wgrad_tiles = [&](int jn, int jl, int sl, int t0, int stage, auto parts_tag, auto ntiles_tag) {
        constexpr int NTL = decltype(ntiles_tag)::value;      // row tiles: all operands are requested before the first product
        constexpr bool do_w2 = (decltype(parts_tag)::value & 1) != 0, do_w1 = (decltype(parts_tag)::value & 2) != 0;
        float *stg = lds + pl.oSTG + (size_t)stage * pl.stg_floats + (size_t)(2 * jl + jn) * pl.stg_net;
        if (do_w2 && lane < 16 && lane < d) {      // d b2: the tiles' DPP row sums
            const float *db = lds + pl.oDB2 + ((t0 * 2 * L) + 2 * jl + jn) * 16 + lane;
            stg[b1o + lane] = NTL > 1 ? db[0] + db[2 * 16 * L] : db[0];
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            f4 dw2 = f4{0.f, 0.f, 0.f, 0.f}, dw1[NIT];
#pragma unroll
            for (int nt = 0; nt < NIT; ++nt) dw1[nt] = f4{0.f, 0.f, 0.f, 0.f};
            float goT[NTL][4], hT[NTL][4], inT[NTL][NIT][4], gpT[NTL][4];
#pragma unroll
            for (int jt = 0; jt < NTL; ++jt) {
                const float *bws = lds + pl.oBW + (size_t)((t0 + jt) * kNsSlots + sl) * BW_SLOT;
                const float *fwl = lds + pl.oFW + (size_t)((t0 + jt) * L + jl) * FW_LAYER;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    if (do_w2) {
                        goT[jt][ks] = bws[oGO + jn * RGO * TS + tgG + 4 * ks];
                        hT[jt][ks] = fwl[2 * XF + (jn * MT + m) * HT + tgH + 4 * ks];
                    }
                    if (do_w1) {
                        inT[jt][0][ks] = bws[tgI + 4 * ks];
                        if (NIT > 1) inT[jt][NIT - 1][ks] = lds[pl.oT2 + (t0 + jt) * HT + tgH + 4 * ks];
                        gpT[jt][ks] = bws[oGP + (jn * MT + m) * HT + tgH + 4 * ks];
                    }
                }
            }
#pragma unroll
            for (int jt = 0; jt < NTL; ++jt)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    if (do_w2) dw2 = mfma16(goT[jt][ks], hT[jt][ks], dw2);                             // [out feature 4q+e][hidden 16m + i]
                    if (do_w1) {
#pragma unroll
                        for (int nt = 0; nt < NIT; ++nt) dw1[nt] = mfma16(gpT[jt][ks], inT[jt][nt][ks], dw1[nt]);   // [hidden 16m + 4q+e][input 16nt + i]
                    }
                }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (do_w2) stg[sS2[m][e]] = dw2[e];
                if (do_w1) {
#pragma unroll
                    for (int nt = 0; nt < NIT; ++nt) stg[sS1[m][nt][e]] = dw1[nt][e];
                }
            }
        }
    };
    // parts: 1 = d W2 (+ d b2), 2 = d W1 (+ d b1), 3 = both; over ntiles (1 or 2) row tiles from tile t0, into stage `stage`
    auto wgrad_job = [&](int jn, int jl, int sl, int parts, int ntiles, int t0, int stage) {
        using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
        if (parts == 3) wgrad_tiles(jn, jl, sl, t0, stage, I3{}, I1{});
        else if (parts == 1) { if (ntiles > 1) wgrad_tiles(jn, jl, sl, t0, stage, I1{}, I2{}); else wgrad_tiles(jn, jl, sl, t0, stage, I1{}, I1{}); }
        else { if (ntiles > 1) wgrad_tiles(jn, jl, sl, t0, stage, I2{}, I2{}); else wgrad_tiles(jn, jl, sl, t0, stage, I2{}, I1{}); }
    };

    int meetings = 0;                                     // parity of the pair's exchange buffer
    Cursor now{0, 0}, ahead{0, 0};                        // the batch being stepped; the batch whose row indices are requested
    int64_t src_next = row_of(ahead);
    f4 nxq, ncq[NIT];
    load_rows(src_next, nxq, ncq);
    advance(ahead);
    src_next = row_of(ahead);
    for (int64_t kb = 0; kb < nb; ++kb, advance(now)) {
        const int rows = (int)rows_at(now);
        const float inv_B = 1.0f / (float)rows;
        const int nw = (rows + 15) >> 4;                  // row tiles of this batch
        const bool active = tile < nw;                    // wave-uniform
        f4 xq = nxq, cin[NIT];
#pragma unroll
        for (int nt = 0; nt < NIT; ++nt) cin[nt] = ncq[nt];
        load_rows(src_next, nxq, ncq);
        advance(ahead);
        src_next = row_of(ahead);
#ifdef RC_STAMP
        unsigned long long ts0 = __builtin_readcyclecounter(), ts1 = ts0, ts2 = ts0, ts3 = ts0, ts4 = ts0;
#endif
        if (wave == kNsWaves - 1 && lane == 0) *ADK = step_adam(lr, beta1, beta2, eps, wd, b1t, b2t);     // idle during the chain
        const bool valid = active && tile * 16 + r < rows;
        float ld = 0.f;
        // ---- forward: this wave's net of every layer; the coupling by both waves of the pair ----
        FwdW fw;
        if (active) load_fwd(0, fw);
        for (int l = 0; l < L; ++l) {
            const uint32_t mb = (uint32_t)(mbits >> (4 * l)) & 15u;
            float *fwl = FWt + (size_t)l * FW_LAYER;
            f4 o = f4{0.f, 0.f, 0.f, 0.f};
            if (active) {
                if (role == 0) {
#pragma unroll
                    for (int e = 0; e < DT; ++e) fwl[e * 64 + lane] = xq[e];           // layer input, for the backward
                }
                f4 in[NIT];
                in[0] = cin[0];
#pragma unroll
                for (int e = 0; e < DT; ++e) in[0][e] = ((mb >> e) & 1u) ? xq[e] + cin[0][e] : cin[0][e];
                if (NIT > 1) in[NIT - 1] = cin[NIT - 1];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = fw.b2[e];
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    f4 acc;
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] = fw.b1[m][e];
#pragma unroll
                    for (int k = 0; k < KIT; ++k) {
                        if (k == 1 && k1) continue;
                        acc = mfma16(fw.a1[m][k], in[k >> 2][k & 3], acc);
                    }
                    float *Hm = fwl + 2 * XF + (role * MT + m) * HT + tp;               // for the backward AND as a weight-gradient operand
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (e == 3 && h3) continue;                                     // (its rows of the tile stay at their initial zeros)
                        const float hv = actf<ACT>(acc[e]) * hm[m][e];
                        Hm[4 * e * TS] = hv;
                        o = mfma16(fw.a2[m][e], hv, o);
                    }
                }
                if (role == 1) {
#pragma unroll
                    for (int e = 0; e < DT; ++e) fwl[XF + e * 64 + lane] = o[e];
                }
                EX[(meetings & 1) * 128 + role * 64 + lane] = o;
                if (l + 1 < L) load_fwd(l + 1, fw);                                     // the next layer's fragments: behind the posting
            }
            __syncthreads();
            if (active) {
                const f4 po = EX[(meetings & 1) * 128 + (1 - role) * 64 + lane];
                const f4 tout = role ? po : o, sout = role ? o : po;
#pragma unroll
                for (int e = 0; e < DT; ++e) {
                    const bool mk = (mb >> e) & 1u;
                    const float xn = fmaf(xq[e], exp_acc(sout[e]), tout[e]);
                    xq[e] = mk ? xq[e] : xn;
                    ld += mk ? 0.f : sout[e];
                }
            }
            ++meetings;
        }
        f4 gy = f4{0.f, 0.f, 0.f, 0.f};
        if (active) {
            float ss = 0.f;
#pragma unroll
            for (int e = 0; e < DT; ++e) { ss = fmaf(xq[e], xq[e], ss); gy[e] = valid ? xq[e] * inv_B : 0.f; }
            ld += __shfl_xor(ld, 16); ld += __shfl_xor(ld, 32);
            ss += __shfl_xor(ss, 16); ss += __shfl_xor(ss, 32);
            float v = (valid && q == 0) ? ld + (-0.5f * ss - prior_c) : 0.f;
            v = row16_sum(v);
            if (lane == 0 && role == 0) RED[tile] = v;
        }
#ifdef RC_STAMP
        __builtin_amdgcn_sched_barrier(0); ts1 = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0);
#endif
        const float gld = valid ? -inv_B : 0.f;
        f4 cinT0;                                         // the condition part of the first input tile + the ones element behind it (d b1)
#pragma unroll
        for (int e = 0; e < 4; ++e) cinT0[e] = (4 * e + q == nin0) ? 1.f : cin[0][e];
        if (NIT > 1 && active && role == 0) {             // the second input tile does not change with the layer
            float *T2 = lds + pl.oT2 + tile * HT + tp;
#pragma unroll
            for (int e = 0; e < 4; ++e) T2[4 * e * TS] = (16 + 4 * e + q == nin0) ? 1.f : cin[NIT - 1][e];
        }
        // ---- backward: this wave's net; the gradient at the layer input by both waves.  What a layer needs besides the
        // incoming gradient is prepared a layer ahead (pre), behind the posting of the previous meeting ----
        struct Pre { f4 x, es, dh[MT]; };
        auto pre = [&](int l, Pre &p) {
            const float *fwl = FWt + (size_t)l * FW_LAYER;
            p.x = f4{0.f, 0.f, 0.f, 0.f}; p.es = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < DT; ++e) { p.x[e] = fwl[e * 64 + lane]; p.es[e] = exp_acc(fwl[XF + e * 64 + lane]); }
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const float *Hm = fwl + 2 * XF + (role * MT + m) * HT + tp;
#pragma unroll
                for (int e = 0; e < 4; ++e) p.dh[m][e] = (e == 3 && h3) ? 0.f : dactf<ACT>(Hm[4 * e * TS]) * hm[m][e];
            }
        };
        BwdW bw;
        Pre cur;
        if (active) { load_bwd(L - 1, bw); pre(L - 1, cur); }
        // The weight gradients of the layer the chain has just left.  Up to two row tiles: on waves 4-7, the SIMDs' second
        // waves -- two tiles: wave 4 + j takes net j & 1, d W2 (j < 2) or d W1 (j >= 2) over both tiles; one tile: waves 6 and 7
        // (next to the idle waves 2 and 3) take a net each, both contractions.  Three or four row tiles: all eight waves walk
        // the chain (waves 4-7: tiles 2 and 3) and every wave takes one job behind its own posting: net, contraction, tile pair
        // (the pairs' sums go to two stages, added in the Adam phase).
        const bool wide = nw > 2;
        const bool has_job = wide || (wave >= 4 && (nw > 1 || wave >= 6));
        const int jnet = wave & 1;
        const int jparts = wide ? 1 + ((wave >> 1) & 1) : (nw == 1 ? 3 : (wave >= 6 ? 2 : 1));
        const int jt0 = wide ? 2 * (wave >> 2) : 0, jstage = wide ? wave >> 2 : 0;
        const int jtiles = wide ? (nw - jt0 < 2 ? nw - jt0 : 2) : nw;
        int slot = 0, pslot = 0;                          // the record the chain fills / the one the helpers read
        for (int l = L - 1; l >= 0; --l) {
            const uint32_t mb = (uint32_t)(mbits >> (4 * l)) & 15u;
            const int nslot = slot == kNsSlots - 1 ? 0 : slot + 1;
            f4 gin = f4{0.f, 0.f, 0.f, 0.f};
            BwdW bn;
            Pre nxt;
            if (active) {
                float *db2 = lds + pl.oDB2 + (size_t)((tile * 2 * L) + 2 * l + role) * 16;
                float *bws = BWt + (size_t)slot * BW_SLOT;
                f4 go = f4{0.f, 0.f, 0.f, 0.f};                                 // d loss / d (this net's output)
#pragma unroll
                for (int e = 0; e < DT; ++e) {
                    const bool mk = (mb >> e) & 1u;
                    go[e] = mk ? 0.f : (role ? fmaf(gy[e] * cur.x[e], cur.es[e], gld) : gy[e]);
                }
#pragma unroll
                for (int e = 0; e < DT; ++e) bws[oGO + role * RGO * TS + tp + 4 * e * TS] = go[e];
                f4 gp[MT];
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    f4 gh = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int e = 0; e < KXT; ++e) gh = mfma16(bw.a2t[m][e], go[e], gh);
#pragma unroll
                    for (int e = 0; e < 4; ++e) gp[m][e] = gh[e] * cur.dh[m][e];
                }
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (e == 3 && h3) continue;
                        gin = mfma16(bw.a1t[m][e], gp[m][e], gin);
                    }
#pragma unroll
                for (int e = 0; e < DT; ++e) {                                  // d b2 (while the MFMA chain runs)
                    const float v = row16_sum(go[e]);
                    if (r == 0 && xok[e]) db2[4 * e + q] = v;
                }
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (e == 3 && h3) continue;
                        bws[oGP + (role * MT + m) * HT + tp + 4 * e * TS] = gp[m][e];
                    }
                EX[(meetings & 1) * 128 + role * 64 + lane] = gin;
                if (role == 0) {                          // the input tile of both nets' weight gradients (behind the posting)
#pragma unroll
                    for (int e = 0; e < KE; ++e) {
                        if (e == 1 && k1) continue;                                     // (zeros, as initialised)
                        bws[tp + 4 * e * TS] = (e < DT && ((mb >> e) & 1u)) ? cur.x[e < DT ? e : 0] + cinT0[e] : cinT0[e];
                    }
                }
                if (l > 0) { load_bwd(l - 1, bn); pre(l - 1, nxt); }
            }
            if (has_job && l < L - 1) wgrad_job(jnet, l + 1, pslot, jparts, jtiles, jt0, jstage);
            __syncthreads();
            if (active) {
                const f4 gother = EX[(meetings & 1) * 128 + (1 - role) * 64 + lane];
                const f4 gs = role ? gin : gother, gt = role ? gother : gin;
#pragma unroll
                for (int e = 0; e < DT; ++e) {
                    const bool mk = (mb >> e) & 1u;
                    gy[e] = xok[e] ? (mk ? gy[e] + (gs[e] + gt[e]) : gy[e] * cur.es[e]) : 0.f;
                }
                if (l > 0) { bw = bn; cur = nxt; }
            }
            ++meetings;
            pslot = slot;
            slot = nslot;
        }
        if (has_job) wgrad_job(jnet, 0, pslot, jparts, jtiles, jt0, jstage);
        __syncthreads();
#ifdef RC_STAMP
        __builtin_amdgcn_sched_barrier(0); ts2 = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0);
        ts3 = ts2;
#endif
        {
            const AdamK a = *ADK;
            adam_phase(lds + pl.oSTG, pl.stg_floats, pl.stg_net, npn, P, wide ? 2 : 1, PAR, MM, VV, pl.mv_lds != 0, exp_avg, exp_avg_sq, a, tid, nthreads);
            if (tid == 0) {
                float acc = 0.f;
                for (int w = 0; w < nw; ++w) acc += RED[w];
                loss_hist[kb] = -acc * inv_B;
            }
        }
#ifdef RC_STAMP
        __builtin_amdgcn_sched_barrier(0); ts4 = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0);
#endif
        __syncthreads();
#ifdef RC_STAMP
        if (kb == 20 && lane == 0)
            printf("NSSTAMP wave %d: fwd %llu bwd %llu adam %llu barrier2+top %llu\n", wave, ts1 - ts0, ts2 - ts1, ts4 - ts3,
                   (unsigned long long)__builtin_readcyclecounter() - ts4);
#endif
    }
    for (int p = tid; p < P; p += nthreads) {
        params[p] = PAR[p];
        if (pl.mv_lds) { exp_avg[p] = MM[p]; exp_avg_sq[p] = VV[p]; }
    }
}

// plan: four row tiles at most (batch <= 64), one hidden layer; false when the form does not apply or its records do not fit LDS
bool make_ns_plan(const KShape &k, int64_t batch, NsPlan *out) {
    if (k.nh != 1 || batch < 1 || batch > 64) return false;
    if (k.d > 16 || k.d + k.c > 31 || k.L > 16 || k.nout[0] > 32 || (k.nout[0] > 16 && rc_kit(k) > 4)) return false;
    NsPlan p;
    std::memset(&p, 0, sizeof(p));
    p.W = (int)((batch + 15) / 16);
    p.P = 2 * k.npn * k.L;
    p.stg_net = k.npn + kDump;
    p.stg_floats = 2 * k.L * p.stg_net;
    const int mt = k.nout[0] <= 16 ? 1 : 2, kit = rc_kit(k), dt = k.d <= 4 ? 1 : 4;
    const int fwl = ns_fw_layer(mt, dt), bws = ns_bw_slot(mt, kit, dt);
    for (int mv = 1; mv >= 0; --mv) {
        int f = 0;
        p.oPAR = f; f += p.P;
        p.oM = f; p.oV = f;
        if (mv) { p.oM = f; f += p.P; p.oV = f; f += p.P; }
        p.oSTG = f; f += (p.W > 2 ? 2 : 1) * p.stg_floats;      // one stage per PAIR of row tiles: a pair is summed inside the weight-gradient jobs
        p.oDB2 = f; f += p.W * 2 * k.L * 16;             // the chain's d b2 row sums per (tile, net, feature)
        p.oRED = f; f += kMaxWaves;
        p.oADK = f; f += 8;
        f = (f + 3) & ~3;
        p.oEX = f; f += p.W * 1024;                      // per tile [parity][role][lane] f4
        p.oT2 = f; f += kit > 4 ? p.W * 16 * TS : 0;
        p.oFW = f; f += p.W * k.L * fwl;
        p.oBW = f;
        f += p.W * kNsSlots * bws;
        p.total_floats = f;
        p.mv_lds = mv;
        if ((size_t)f * sizeof(float) <= kLdsMax) { *out = p; return true; }
    }
    return false;
}

template <int MT, int KIT, int ACT, int DT>
int launch_ns_k(hipStream_t st, const KShape &k, const NsPlan &p, const EpochArgs &a) {
    auto kern = k_fit_resident_ns<MT, KIT, ACT, DT>;
    static std::atomic<uint64_t> attr_done{0};
    const int rc = allow_big_lds(reinterpret_cast<const void *>(kern), (int)kLdsMax, attr_done);
    if (rc) return rc;
    note_dispatch(RNVP_PROFILE_TRAIN, "k_fit_resident_ns", RNVP_VARIANT_RESIDENT, 1, kNsThreads / 64, 1, RNVP_PREC_F32, a.n);
    note_launches(RNVP_PROFILE_TRAIN, 1);
    {
        KernelTimer timer(st, RNVP_PROFILE_TRAIN);
        hipLaunchKernelGGL(kern, dim3(1), dim3(kNsThreads), (size_t)p.total_floats * sizeof(float), st, k, p, a.params, a.masks, a.x, a.c, a.perm,
                           a.n, a.batch_size, a.n_epochs, a.loss_hist, a.exp_avg, a.exp_avg_sq, a.lr, a.beta1, a.beta2, a.eps, a.wd,
                           std::pow(a.beta1, (double)a.first_step), std::pow(a.beta2, (double)a.first_step));
    }
    RNVP_HIP_TRY(hipGetLastError());
    return RNVP_OK;
}

template <int MT, int KIT>
int launch_ns_d(hipStream_t st, const KShape &k, const NsPlan &p, const EpochArgs &a) {
    const bool th = k.act == RNVP_ACT_TANH;
    if (k.d <= 4) return th ? launch_ns_k<MT, KIT, RNVP_ACT_TANH, 1>(st, k, p, a) : launch_ns_k<MT, KIT, RNVP_ACT_RELU, 1>(st, k, p, a);
    return th ? launch_ns_k<MT, KIT, RNVP_ACT_TANH, 4>(st, k, p, a) : launch_ns_k<MT, KIT, RNVP_ACT_RELU, 4>(st, k, p, a);
}

template <int MT>
int launch_ns_m(hipStream_t st, const KShape &k, const NsPlan &p, const EpochArgs &a) {
    const int kit = rc_kit(k);
    if (kit == 2) return launch_ns_d<MT, 2>(st, k, p, a);
    if (kit == 4) return launch_ns_d<MT, 4>(st, k, p, a);
    if constexpr (MT == 1) return launch_ns_d<MT, 8>(st, k, p, a);
    return RNVP_EUNSUPPORTED;
}

}  // namespace

bool ns_applies(const KShape &k, int64_t batch) {
    NsPlan p;
    return make_ns_plan(k, batch, &p);
}

int launch_ns(hipStream_t st, const KShape &k, const EpochArgs &a) {
    NsPlan p;
    if (!make_ns_plan(k, a.batch_size, &p)) return RNVP_EUNSUPPORTED;
    return k.nout[0] <= 16 ? launch_ns_m<1>(st, k, p, a) : launch_ns_m<2>(st, k, p, a);
}

}  // namespace resident
}  // namespace rnvp
