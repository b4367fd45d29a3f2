// rnvp_resident_ns.hip -- resident fit (rnvp_resident.hip), net-split form for batches of at most 32 rows (the reference's default).
//
// A resident step is bound by the instruction stream of the wave that walks a 16-row tile (rnvp_resident.hip), and at
// batch 32 only two of a CU's four SIMDs have such a wave.  Here every row tile gets a PAIR of waves, one per net: wave
// 2t runs the t net of tile t, wave 2t+1 its s net -- both keep the tile's x, condition and gradient, each walks half of
// the GEMMs, tanh's and weight gradients.  Per layer the pair meets twice through LDS: the forward exchanges the nets'
// outputs (both then apply the coupling), the backward the two nets' input gradients (both then update the gradient at the
// layer input).  The exchange buffers alternate by parity, so one workgroup barrier per meeting is enough.  The forward
// keeps each wave's hidden activations (and s) in a lane-private LDS area; nothing is recomputed.  Same arithmetic per
// element as the one-wave form; the sum of the two input gradients is taken in the same order (s + t) by both waves.
#include "rnvp_resident_dev.h"

namespace rnvp {
namespace resident {
namespace {

// MT hidden tiles, KIT k-steps of the net input, DT slots of x that hold features; 4 waves = 2 row tiles x 2 nets
constexpr int kNsThreads = 512;          // waves 0-3 walk the rows, all of them share the Adam phase

template <int MT, int KIT, int ACT, int DT>
__global__ void __launch_bounds__(kNsThreads)
k_fit_resident_ns(KShape s, RcPlan pl, float *__restrict__ params, const uint8_t *__restrict__ masks, const float *__restrict__ x,
                  const float *__restrict__ c, const int64_t *__restrict__ perm, int64_t n, int64_t batch, int64_t n_epochs,
                  float *__restrict__ loss_hist, float *__restrict__ exp_avg, float *__restrict__ exp_avg_sq, double lr, double beta1,
                  double beta2, double eps, double wd, double b1t, double b2t) {
    constexpr int NIT = KIT > 4 ? 2 : 1;
    constexpr int KXT = KIT < DT ? KIT : DT;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, nthreads = blockDim.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, r = lane & 15, i = r;
    const int tile = wave >> 1, role = wave & 1;          // role 0: the t net, 1: the s net
    const int d = s.d, cd = s.c, L = s.L, P = pl.P, h = s.nout[0], nin0 = d + cd, npn = s.npn;
    const int w0 = s.woff[0], b0 = s.boff[0], w1 = s.woff[1], b1o = s.boff[1];
    const int pi = 4 * (i & 3) + (i >> 2);
    float *PAR = lds + pl.oPAR, *MM = lds + pl.oM, *VV = lds + pl.oV, *STG = lds + pl.oSTG + (size_t)tile * pl.stg_floats;
    float *RED = lds + pl.oRED;
    f4 *XS = reinterpret_cast<f4 *>(lds + pl.oXS + (size_t)wave * pl.xs_floats);       // [layer][x | hidden tiles | s][lane]
    f4 *EX = reinterpret_cast<f4 *>(lds + pl.oEX) + (size_t)tile * 256;                // [parity][role][lane]
    float *TT = lds + pl.oTT + (size_t)wave * pl.tt_floats;
    float *T_in = TT, *T_go = TT + NIT * 16 * TS, *T_h = T_go + 16 * TS, *T_gp = T_h + 16 * TS;
    for (int e = tid; e < pl.total_floats; e += nthreads) lds[e] = 0.f;
    __syncthreads();
    for (int p = tid; p < P; p += nthreads) {
        PAR[p] = params[p];
        if (pl.mv_lds) { MM[p] = exp_avg[p]; VV[p] = exp_avg_sq[p]; }
    }
    __syncthreads();
    const float prior_c = 0.5f * (float)d * kLog2Pi;
    const int64_t nb_e = (n + batch - 1) / batch, nb = nb_e * n_epochs;
    auto batch_at = [&](int64_t kb, int64_t &s0, int64_t &rows) {
        const int64_t ep = kb / nb_e, k = kb - ep * nb_e;
        s0 = ep * n + k * batch;
        rows = (n - k * batch < batch) ? n - k * batch : batch;
    };

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
            const bool mk = masks ? masks[l * d + j] != 0 : (((j + l + (s.alt == 2 ? 1 : 0)) & 1) != 0);
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

    auto row_of = [&](int64_t kb) -> int64_t {
        if (kb >= nb) return -1;
        int64_t s0, rows;
        batch_at(kb, s0, rows);
        const int64_t rr = (int64_t)tile * 16 + r;
        return rr < rows ? perm[s0 + rr] : -1;
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
    // the pair's meeting: my f4 out (post), the partner's back (fetch); the buffers alternate by the meeting's parity, so one
    // workgroup barrier per meeting is enough -- whatever is independent of the partner goes between post and fetch
    int meetings = 0;
    auto post = [&](const f4 &mine) { EX[(meetings & 1) * 128 + role * 64 + lane] = mine; };
    auto fetch = [&]() -> f4 {
        __syncthreads();
        const f4 v = EX[(meetings & 1) * 128 + (1 - role) * 64 + lane];
        ++meetings;
        return v;
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

    int64_t src_next = row_of(0);
    f4 nxq, ncq[NIT];
    load_rows(src_next, nxq, ncq);
    src_next = row_of(1);
    for (int64_t kb = 0; kb < nb; ++kb) {
        int64_t s0, rows64;
        batch_at(kb, s0, rows64);
        (void)s0;
        const int rows = (int)rows64;
        const float inv_B = 1.0f / (float)rows;
        const int nw = (rows + 15) >> 4;                  // row tiles of this batch
        f4 xq = nxq, cin[NIT];
#pragma unroll
        for (int nt = 0; nt < NIT; ++nt) cin[nt] = ncq[nt];
        load_rows(src_next, nxq, ncq);
        src_next = row_of(kb + 2);
#ifdef RC_STAMP
        unsigned long long ts0 = __builtin_readcyclecounter(), ts1 = ts0, ts2 = ts0, ts3 = ts0, ts4 = ts0;
#endif
        if (tile < nw) {
            const bool valid = tile * 16 + r < rows;
            float ld = 0.f;
            // ---- forward: this wave's net of every layer; the coupling by both waves of the pair ----
            FwdW fw;
            load_fwd(0, fw);
            for (int l = 0; l < L; ++l) {
                const uint32_t mb = (uint32_t)(mbits >> (4 * l)) & 15u;
                f4 *rec = XS + (size_t)l * (MT + 2) * 64 + lane;
                rec[0] = xq;                                               // layer input, for the backward
                f4 in[NIT];
                in[0] = cin[0];
#pragma unroll
                for (int e = 0; e < DT; ++e) in[0][e] = ((mb >> e) & 1u) ? xq[e] + cin[0][e] : cin[0][e];
                if (NIT > 1) in[NIT - 1] = cin[NIT - 1];
                f4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = fw.b2[e];
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    f4 acc;
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] = fw.b1[m][e];
#pragma unroll
                    for (int k = 0; k < KIT; ++k) acc = mfma16(fw.a1[m][k], in[k >> 2][k & 3], acc);
                    f4 hv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) hv[e] = actf<ACT>(acc[e]) * hm[m][e];
                    rec[(1 + m) * 64] = hv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o = mfma16(fw.a2[m][e], hv[e], o);
                }
                post(o);
                if (l + 1 < L) load_fwd(l + 1, fw);                        // the next layer's fragments: requested before the meeting
                const f4 po = fetch();
                const f4 tout = role ? po : o, sout = role ? o : po;
                rec[(1 + MT) * 64] = sout;
#pragma unroll
                for (int e = 0; e < DT; ++e) {
                    const bool mk = (mb >> e) & 1u;
                    const float xn = fmaf(xq[e], exp_acc(sout[e]), tout[e]);
                    xq[e] = mk ? xq[e] : xn;
                    ld += mk ? 0.f : sout[e];
                }
            }
            f4 gy;
            {
                float ss = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) gy[e] = 0.f;
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
            f4 cinT[NIT];
#pragma unroll
            for (int nt = 0; nt < NIT; ++nt)
#pragma unroll
                for (int e = 0; e < 4; ++e) cinT[nt][e] = (16 * nt + 4 * e + q == nin0) ? 1.f : cin[nt][e];
            if (NIT > 1) { wfence(); tile_put(T_in + (NIT - 1) * 16 * TS, cinT[NIT - 1], q, r); }
            // ---- backward: this wave's net; the gradient at the layer input by both waves ----
            BwdW bw;
            load_bwd(L - 1, bw);
            for (int l = L - 1; l >= 0; --l) {
                float *stg = STG + (size_t)(2 * l + role) * pl.stg_net;
                const uint32_t mb = (uint32_t)(mbits >> (4 * l)) & 15u;
                const f4 *rec = XS + (size_t)l * (MT + 2) * 64 + lane;
                xq = rec[0];
                const f4 sout = rec[(1 + MT) * 64];
                f4 in0T = cinT[0];
#pragma unroll
                for (int e = 0; e < DT; ++e) in0T[e] = ((mb >> e) & 1u) ? xq[e] + cinT[0][e] : cinT[0][e];
                wfence();
                tile_put(T_in, in0T, q, r);
                f4 es = f4{0.f, 0.f, 0.f, 0.f}, go = f4{0.f, 0.f, 0.f, 0.f};      // d loss / d (this net's output)
#pragma unroll
                for (int e = 0; e < DT; ++e) {
                    const bool mk = (mb >> e) & 1u;
                    es[e] = exp_acc(sout[e]);
                    go[e] = mk ? 0.f : (role ? fmaf(gy[e] * xq[e], es[e], gld) : gy[e]);
                }
#pragma unroll
                for (int e = 0; e < DT; ++e) T_go[(4 * e + q) * TS + r] = go[e];   // the other elements stay at their initial zeros
#pragma unroll
                for (int e = 0; e < DT; ++e) {                                     // d b2
                    const float v = row16_sum(go[e]);
                    if (r == 0 && xok[e]) stg[b1o + 4 * e + q] = v;
                }
                f4 hh[MT], gp[MT], gin = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    hh[m] = rec[(1 + m) * 64];
                    f4 gh = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int e = 0; e < KXT; ++e) gh = mfma16(bw.a2t[m][e], go[e], gh);
#pragma unroll
                    for (int e = 0; e < 4; ++e) gp[m][e] = gh[e] * dactf<ACT>(hh[m][e]) * hm[m][e];
                }
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int e = 0; e < 4; ++e) gin = mfma16(bw.a1t[m][e], gp[m][e], gin);
                post(gin);
                if (l > 0) load_bwd(l - 1, bw);
                // weight gradients of this net: contractions over the tile's 16 rows through the transposition tiles
                wfence();
                float inT[NIT][4], goT[4];
#pragma unroll
                for (int nt = 0; nt < NIT; ++nt) tile_get(T_in + nt * 16 * TS, q, i, inT[nt]);
                tile_get(T_go, q, i, goT);
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    wfence();
                    tile_put(T_h, hh[m], q, r);
                    tile_put(T_gp, gp[m], q, r);
                    wfence();
                    float hT[4], gpT[4];
                    tile_get(T_h, q, i, hT);
                    tile_get(T_gp, q, i, gpT);
                    f4 dw2 = f4{0.f, 0.f, 0.f, 0.f}, dw1[NIT];
#pragma unroll
                    for (int nt = 0; nt < NIT; ++nt) dw1[nt] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        dw2 = mfma16(goT[ks], hT[ks], dw2);
#pragma unroll
                        for (int nt = 0; nt < NIT; ++nt) dw1[nt] = mfma16(gpT[ks], inT[nt][ks], dw1[nt]);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        stg[sS2[m][e]] = dw2[e];
#pragma unroll
                        for (int nt = 0; nt < NIT; ++nt) stg[sS1[m][nt][e]] = dw1[nt][e];
                    }
                }
                const f4 gother = fetch();
                const f4 gs = role ? gin : gother, gt = role ? gother : gin;
#pragma unroll
                for (int e = 0; e < DT; ++e) {
                    const bool mk = (mb >> e) & 1u;
                    gy[e] = xok[e] ? (mk ? gy[e] + (gs[e] + gt[e]) : gy[e] * es[e]) : 0.f;
                }
            }
        } else {
            for (int t = 0; t < 2 * L; ++t) { __syncthreads(); ++meetings; }      // a ragged batch: keep the other pair's meetings company
        }
#ifdef RC_STAMP
        __builtin_amdgcn_sched_barrier(0); ts2 = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0);
#endif
        __syncthreads();
#ifdef RC_STAMP
        __builtin_amdgcn_sched_barrier(0); ts3 = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0);
#endif
        {
            const AdamK a = step_adam(lr, beta1, beta2, eps, wd, b1t, b2t);
            adam_phase(lds + pl.oSTG, pl.stg_floats, pl.stg_net, npn, P, nw, PAR, MM, VV, pl.mv_lds != 0, exp_avg, exp_avg_sq, a, tid, nthreads);
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
            printf("NSSTAMP wave %d: fwd %llu bwd %llu barrier %llu adam %llu barrier2+top %llu\n", wave, ts1 - ts0, ts2 - ts1, ts3 - ts2, ts4 - ts3,
                   (unsigned long long)__builtin_readcyclecounter() - ts4);
#endif
    }
    for (int p = tid; p < P; p += nthreads) {
        params[p] = PAR[p];
        if (pl.mv_lds) { exp_avg[p] = MM[p]; exp_avg_sq[p] = VV[p]; }
    }
}

template <int MT, int KIT, int ACT, int DT>
int launch_ns_k(hipStream_t st, const KShape &k, const RcPlan &p, const EpochArgs &a) {
    auto kern = k_fit_resident_ns<MT, KIT, ACT, DT>;
    static std::atomic<uint64_t> attr_done{0};
    const int rc = allow_big_lds(reinterpret_cast<const void *>(kern), (int)kLdsMax, attr_done);
    if (rc) return rc;
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
int launch_ns_d(hipStream_t st, const KShape &k, const RcPlan &p, const EpochArgs &a) {
    const bool th = k.act == RNVP_ACT_TANH;
    if (k.d <= 4) return th ? launch_ns_k<MT, KIT, RNVP_ACT_TANH, 1>(st, k, p, a) : launch_ns_k<MT, KIT, RNVP_ACT_RELU, 1>(st, k, p, a);
    return th ? launch_ns_k<MT, KIT, RNVP_ACT_TANH, 4>(st, k, p, a) : launch_ns_k<MT, KIT, RNVP_ACT_RELU, 4>(st, k, p, a);
}

template <int MT>
int launch_ns_m(hipStream_t st, const KShape &k, const RcPlan &p, const EpochArgs &a) {
    const int kit = rc_kit(k);
    if (kit == 2) return launch_ns_d<MT, 2>(st, k, p, a);
    if (kit == 4) return launch_ns_d<MT, 4>(st, k, p, a);
    if constexpr (MT == 1) return launch_ns_d<MT, 8>(st, k, p, a);
    return RNVP_EUNSUPPORTED;
}

}  // namespace

// plan of the net-split form: two row tiles at most (batch <= 32), one hidden layer; false when it does not apply / fit
bool make_ns_plan(const KShape &k, int64_t batch, RcPlan *out) {
    if (k.nh != 1 || batch < 1 || batch > 32) return false;
    if (k.d > 16 || k.d + k.c > 31 || k.L > 16 || k.nout[0] > 32 || (k.nout[0] > 16 && rc_kit(k) > 4)) return false;
    RcPlan p;
    std::memset(&p, 0, sizeof(p));
    p.W = (int)((batch + 15) / 16);
    p.P = 2 * k.npn * k.L;
    p.stg_net = k.npn + kDump;
    p.stg_floats = 2 * k.L * p.stg_net;
    const int mt = k.nout[0] <= 16 ? 1 : 2, nit = rc_kit(k) > 4 ? 2 : 1;
    p.save = 1;
    p.xs_floats = k.L * 64 * 4 * (mt + 2);               // per WAVE: layer input, hidden tiles, s
    p.tt_floats = (nit + 3) * 16 * TS;                   // per WAVE
    for (int mv = 1; mv >= 0; --mv) {
        int f = 0;
        p.oPAR = f; f += p.P;
        p.oM = f; p.oV = f;
        if (mv) { p.oM = f; f += p.P; p.oV = f; f += p.P; }
        p.oSTG = f; f += p.W * p.stg_floats;
        p.oRED = f; f += kMaxWaves;
        f = (f + 3) & ~3;
        p.oEX = f; f += 2 * 1024;                        // two tiles x [parity][role][lane] f4
        p.oXS = f; f += 4 * p.xs_floats;
        p.oTT = f; f += 4 * p.tt_floats;
        p.total_floats = f;
        p.mv_lds = mv;
        if ((size_t)f * sizeof(float) <= kLdsMax) { *out = p; return true; }
    }
    return false;
}

int launch_ns(hipStream_t st, const KShape &k, const RcPlan &p, const EpochArgs &a) {
    return k.nout[0] <= 16 ? launch_ns_m<1>(st, k, p, a) : launch_ns_m<2>(st, k, p, a);
}

}  // namespace resident
}  // namespace rnvp
