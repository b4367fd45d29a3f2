// cvae_resident.hip -- resident fit (rnvp_resident.hip) for the conditional VAE: cvae_fit_epoch as one persistent launch.
#include "rnvp_resident_dev.h"

namespace rnvp {
namespace resident {
namespace {

// ---- the same for the conditional VAE: encoder -> reparameterise -> decoder -> KL + MSE -> backward (cvae.py:186-203) -------
// One hidden layer per MLP (the reference's default hidden=(10,), latent 2, batch_size 32: cvae.py:145).  The two nets of a
// step run one after the other (the decoder eats z), nothing is recomputed or saved: a row's activations stay in registers
// from the encoder's first GEMM to the last weight gradient.  The encoder's head is gathered so that lane group q receives
// mu[q], mu[4+q] in slots 0, 1 and log_sigma[q], log_sigma[4+q] in slots 2, 3 (latent <= 8): z = mu + e^{ls/2} eps is
// lane-local and comes out in the element order the decoder's first GEMM wants.
struct CvPlan {
    int W, P, mv_lds;
    int oPAR, oM, oV, oSTG, oRED, oTT;                 // float offsets
    int tt_floats, stg_floats;                         // per wave
    int total_floats;
};

template <int MT, int KIT, int ACT, int WMAX>
__global__ void __launch_bounds__(64 * WMAX)
k_cvae_fit_resident(CvaeK s, CvPlan pl, float *__restrict__ params, const float *__restrict__ x, const float *__restrict__ c,
                    const int64_t *__restrict__ perm, const float *__restrict__ eps, int64_t n, int64_t batch, float klw,
                    float *__restrict__ loss_hist, float *__restrict__ exp_avg, float *__restrict__ exp_avg_sq, double lr, double beta1,
                    double beta2, double adam_eps, double wd, double b1t, double b2t) {
    constexpr int NIT = KIT > 4 ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, nthreads = blockDim.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, r = lane & 15, i = r;
    const int d = s.d, cd = s.c, lat = s.lat, P = pl.P, h = s.enc.nout[0], ne = d + cd, nd = lat + cd, pe = s.pe;
    const int w0e = s.enc.woff[0], b0e = s.enc.boff[0], w1e = s.enc.woff[1], b1e = s.enc.boff[1];
    const int w0d = pe + s.dec.woff[0], b0d = pe + s.dec.boff[0], w1d = pe + s.dec.woff[1], b1d = pe + s.dec.boff[1];
    const int pi = 4 * (i & 3) + (i >> 2);
    float *PAR = lds + pl.oPAR, *MM = lds + pl.oM, *VV = lds + pl.oV, *STG = lds + pl.oSTG + (size_t)wave * pl.stg_floats;
    float *RED = lds + pl.oRED;
    float *TT = lds + pl.oTT + (size_t)wave * pl.tt_floats;
    float *T_ine = TT, *T_ind = TT + NIT * 16 * TS, *T_go = T_ind + NIT * 16 * TS, *T_h = T_go + 2 * 16 * TS;      // [enc in][dec in][g head, g rec][h, g_pre]
    for (int e = tid; e < pl.total_floats; e += nthreads) lds[e] = 0.f;
    __syncthreads();
    for (int p = tid; p < P; p += nthreads) {
        PAR[p] = params[p];
        if (pl.mv_lds) { MM[p] = exp_avg[p]; VV[p] = exp_avg_sq[p]; }
    }
    __syncthreads();
    const int64_t nb = (n + batch - 1) / batch;

    // head position p (0..15) of the encoder's last Linear: p < 8: mu[p], else log_sigma[p - 8]; its row in W / b, or -1
    auto head_row = [&](int p) -> int { return p < 8 ? (p < lat ? p : -1) : (p - 8 < lat ? lat + p - 8 : -1); };
    // ---- per-lane constants ----
    const int gW1e = w0e + pi * ne + q, gW1d = w0d + pi * nd + q;                 // + m 16 nin + 4k
    const int gB1e = b0e + q, gB1d = b0d + q;                                     // + 16m + 4e
    const int hr_a = head_row(pi);
    const int gW2e = w1e + (hr_a < 0 ? 0 : hr_a) * h + q;                         // + 16m + 4e : W2e[row(pi)][16m + 4e + q]
    const int gW2d = w1d + pi * h + q;                                            // W2d[pi][16m + 4e + q]
    const int gB2d = b1d + q;                                                     // + 4e
    const int gW1td = w0d + q * nd + pi;                                          // + (16m + 4e) nd : W1d[16m + 4e + q][pi]   (g_z)
    int gB2e[4], gW2te[4], gW2td[4];
    float hv[4];                                                                  // 1 where head slot e of this lane group is a real unit
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int hr = head_row(4 * e + q);
        hv[e] = hr >= 0 ? 1.f : 0.f;
        gB2e[e] = b1e + (hr < 0 ? 0 : hr);
        gW2te[e] = w1e + (hr < 0 ? 0 : hr) * h + pi;                              // + 16m : W2e[row(4e + q)][16m + pi]      (g_h, encoder)
        gW2td[e] = w1d + (4 * e + q) * h + pi;                                    // + 16m : W2d[4e + q][16m + pi]           (g_h, decoder)
    }
    f4 hm[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) hm[m][e] = 16 * m + 4 * e + q < h ? 1.f : 0.f;
    bool xok[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) xok[e] = 4 * e + q < d;
    const int dump = P + (lane & 15);
    int sS2e[MT][4], sS2d[MT][4], sS1e[MT][NIT][4], sS1d[MT][NIT][4], sB2e[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { const int hr = head_row(4 * e + q); sB2e[e] = hr >= 0 ? b1e + hr : dump; }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int po = 4 * q + e, hid_n = 16 * m + i, hid_m = 16 * m + 4 * q + e, hr = head_row(po);
            sS2e[m][e] = (hr >= 0 && hid_n < h) ? w1e + hr * h + hid_n : dump;
            sS2d[m][e] = (po < d && hid_n < h) ? w1d + po * h + hid_n : dump;
#pragma unroll
            for (int nt = 0; nt < NIT; ++nt) {
                const int j = 16 * nt + i;
                sS1e[m][nt][e] = hid_m >= h ? dump : (j < ne ? w0e + hid_m * ne + j : (j == ne ? b0e + hid_m : dump));
                sS1d[m][nt][e] = hid_m >= h ? dump : (j < nd ? w0d + hid_m * nd + j : (j == nd ? b0d + hid_m : dump));
            }
        }

    auto row_of = [&](int64_t kb) -> int64_t {
        if (kb >= nb) return -1;
        const int64_t s0 = kb * batch;
        const int64_t rows = (n - s0 < batch) ? n - s0 : batch;
        const int64_t rr = (int64_t)wave * 16 + r;
        return rr < rows ? s0 + rr : -1;                                          // position in the epoch (eps is in that order)
    };
    // x; the condition behind x (encoder input) and behind z (decoder input); eps in the element order of z
    auto load_rows = [&](int64_t pos, f4 &xo, f4 (&ce)[NIT], f4 (&cdv)[NIT], f4 &eo) {
        const int64_t src = pos >= 0 ? perm[pos] : 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            xo[e] = (pos >= 0 && xok[e]) ? x[src * d + 4 * e + q] : 0.f;
            eo[e] = (pos >= 0 && e < 2 && 4 * e + q < lat) ? eps[pos * lat + 4 * e + q] : 0.f;
        }
#pragma unroll
        for (int nt = 0; nt < NIT; ++nt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = 16 * nt + 4 * e + q;
                ce[nt][e] = (pos >= 0 && j >= d && j < ne) ? c[src * cd + (j - d)] : 0.f;
                cdv[nt][e] = (pos >= 0 && j >= lat && j < nd) ? c[src * cd + (j - lat)] : 0.f;
            }
    };
    // hidden layer of one MLP: act(W1 in + b1), padding units zeroed
    auto hidden_fwd = [&](int gW1, int gB1, int nin, const f4 (&in)[NIT], f4 (&hh)[MT]) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            f4 acc;
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = PAR[gB1 + 16 * m + 4 * e];
#pragma unroll
            for (int k = 0; k < KIT; ++k) acc = mfma16(PAR[gW1 + m * 16 * nin + 4 * k], in[k >> 2][k & 3], acc);
#pragma unroll
            for (int e = 0; e < 4; ++e) hh[m][e] = actf<ACT>(acc[e]) * hm[m][e];
        }
    };

    f4 nx, nce[NIT], ncd[NIT], nep;
    load_rows(row_of(0), nx, nce, ncd, nep);
    for (int64_t kb = 0; kb < nb; ++kb) {
        const int64_t s0 = kb * batch;
        const int rows = (int)((n - s0 < batch) ? n - s0 : batch);
        const float inv_B = 1.0f / (float)rows;
        const int nw = (rows + 15) >> 4;
        const f4 xq = nx, epq = nep;
        f4 cinE[NIT], cinD[NIT];
#pragma unroll
        for (int nt = 0; nt < NIT; ++nt) { cinE[nt] = nce[nt]; cinD[nt] = ncd[nt]; }
        load_rows(row_of(kb + 1), nx, nce, ncd, nep);                             // the next batch's rows, a step ahead
        if (wave < nw) {
            const bool valid = wave * 16 + r < rows;
            const float sc = valid ? inv_B : 0.f;
            // ---- encoder (cvae.py:58-62) ----
            f4 inE[NIT];
            inE[0] = xq + cinE[0];
            if (NIT > 1) inE[NIT - 1] = cinE[NIT - 1];
            f4 hhe[MT];
            hidden_fwd(gW1e, gB1e, ne, inE, hhe);
            f4 oe;                                                                // slots 0, 1: mu; 2, 3: log_sigma
#pragma unroll
            for (int e = 0; e < 4; ++e) oe[e] = PAR[gB2e[e]];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int e = 0; e < 4; ++e) oe = mfma16(PAR[gW2e + 16 * m + 4 * e], hhe[m][e], oe);
            float mu[2], ls[2], el[2], els[2];
            f4 zq = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                mu[e] = oe[e] * hv[e]; ls[e] = oe[2 + e] * hv[2 + e];
                el[e] = exp_acc(0.5f * ls[e]); els[e] = exp_acc(ls[e]);
                zq[e] = hv[e] != 0.f ? fmaf(el[e], epq[e], mu[e]) : 0.f;           // sample_z, cvae.py:188
            }
            // ---- decoder (cvae.py:81-84) ----
            f4 inD[NIT];
            inD[0] = zq + cinD[0];
            if (NIT > 1) inD[NIT - 1] = cinD[NIT - 1];
            f4 hhd[MT];
            hidden_fwd(gW1d, gB1d, nd, inD, hhd);
            f4 xr;
#pragma unroll
            for (int e = 0; e < 4; ++e) xr[e] = PAR[gB2d + 4 * e];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int e = 0; e < 4; ++e) xr = mfma16(PAR[gW2d + 16 * m + 4 * e], hhd[m][e], xr);
            // ---- loss: KL_weight * KL + MSE (cvae.py:190-193) and its seeds ----
            f4 grec;
            {
                const float inv_d = 1.f / (float)d;
                float kl = 0.f, se = 0.f;
#pragma unroll
                for (int e = 0; e < 2; ++e) kl += hv[e] != 0.f ? 1.f + ls[e] - mu[e] * mu[e] - els[e] : 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float df = xok[e] ? xr[e] - xq[e] : 0.f;
                    se = fmaf(df, df, se);
                    grec[e] = (2.f * sc * inv_d) * df;                             // d MSE / d x_rec
                }
                float lrow = klw * (-0.5f * kl) + se * inv_d;
                lrow += __shfl_xor(lrow, 16); lrow += __shfl_xor(lrow, 32);
                float v = (valid && q == 0) ? lrow : 0.f;
                v = row16_sum(v);
                if (lane == 0) RED[wave] = v;
            }
            // ---- decoder backward ----
            f4 gpd[MT], gz = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                f4 gh = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) gh = mfma16(PAR[gW2td[e] + 16 * m], grec[e], gh);
#pragma unroll
                for (int e = 0; e < 4; ++e) gpd[m][e] = gh[e] * dactf<ACT>(hhd[m][e]) * hm[m][e];
            }
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int e = 0; e < 4; ++e) gz = mfma16(PAR[gW1td + (16 * m + 4 * e) * nd], gpd[m][e], gz);
            // ---- d loss / d mu, d log_sigma; encoder backward (no input gradient needed) ----
            f4 goe;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                goe[e] = hv[e] * fmaf(klw * sc, mu[e], gz[e]);
                goe[2 + e] = hv[2 + e] * (gz[e] * epq[e] * 0.5f * el[e] + klw * sc * (-0.5f) * (1.f - els[e]));
            }
            f4 gpe[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                f4 gh = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) gh = mfma16(PAR[gW2te[e] + 16 * m], goe[e], gh);
#pragma unroll
                for (int e = 0; e < 4; ++e) gpe[m][e] = gh[e] * dactf<ACT>(hhe[m][e]) * hm[m][e];
            }
            // ---- biases of the last Linears (row sums), then the weight gradients through the transposition tiles ----
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float ve = row16_sum(goe[e]), vd = row16_sum(grec[e]);
                if (r == 0) { STG[sB2e[e]] = ve; if (xok[e]) STG[b1d + 4 * e + q] = vd; }
            }
            wfence();
#pragma unroll
            for (int nt = 0; nt < NIT; ++nt) {
                f4 te = inE[nt], td = inD[nt];
#pragma unroll
                for (int e = 0; e < 4; ++e) {                                       // the ones element behind the inputs: d b1
                    if (16 * nt + 4 * e + q == ne) te[e] = 1.f;
                    if (16 * nt + 4 * e + q == nd) td[e] = 1.f;
                }
                tile_put(T_ine + nt * 16 * TS, te, q, r);
                tile_put(T_ind + nt * 16 * TS, td, q, r);
            }
            tile_put(T_go, goe, q, r);
            tile_put(T_go + 16 * TS, grec, q, r);
            wfence();
            float inTe[NIT][4], inTd[NIT][4], goT[2][4];
#pragma unroll
            for (int nt = 0; nt < NIT; ++nt) { tile_get(T_ine + nt * 16 * TS, q, i, inTe[nt]); tile_get(T_ind + nt * 16 * TS, q, i, inTd[nt]); }
            tile_get(T_go, q, i, goT[0]);
            tile_get(T_go + 16 * TS, q, i, goT[1]);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                wfence();
                tile_put(T_h, hhe[m], q, r);
                tile_put(T_h + 16 * TS, gpe[m], q, r);
                tile_put(T_h + 32 * TS, hhd[m], q, r);
                tile_put(T_h + 48 * TS, gpd[m], q, r);
                wfence();
                float hT[2][4], gpT[2][4];
                tile_get(T_h, q, i, hT[0]); tile_get(T_h + 16 * TS, q, i, gpT[0]);
                tile_get(T_h + 32 * TS, q, i, hT[1]); tile_get(T_h + 48 * TS, q, i, gpT[1]);
                f4 dw2[2], dw1[2][NIT];
#pragma unroll
                for (int net = 0; net < 2; ++net) {
                    dw2[net] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int nt = 0; nt < NIT; ++nt) dw1[net][nt] = f4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    dw2[0] = mfma16(goT[0][ks], hT[0][ks], dw2[0]);                 // [head position 4q+e][hidden 16m + i]
                    dw2[1] = mfma16(goT[1][ks], hT[1][ks], dw2[1]);                 // [out feature 4q+e][hidden 16m + i]
#pragma unroll
                    for (int nt = 0; nt < NIT; ++nt) {
                        dw1[0][nt] = mfma16(gpT[0][ks], inTe[nt][ks], dw1[0][nt]);  // [hidden 16m + 4q+e][encoder input 16nt + i]
                        dw1[1][nt] = mfma16(gpT[1][ks], inTd[nt][ks], dw1[1][nt]);  // [hidden 16m + 4q+e][decoder input 16nt + i]
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    STG[sS2e[m][e]] = dw2[0][e];
                    STG[sS2d[m][e]] = dw2[1][e];
#pragma unroll
                    for (int nt = 0; nt < NIT; ++nt) { STG[sS1e[m][nt][e]] = dw1[0][nt][e]; STG[sS1d[m][nt][e]] = dw1[1][nt][e]; }
                }
            }
        }
        __syncthreads();
        {
            const AdamK a = step_adam(lr, beta1, beta2, adam_eps, wd, b1t, b2t);
            adam_phase(lds + pl.oSTG, pl.stg_floats, P, P, P, nw, PAR, MM, VV, pl.mv_lds != 0, exp_avg, exp_avg_sq, a, tid, nthreads);
            if (tid == 0 && loss_hist) {
                float acc = 0.f;
                for (int w = 0; w < nw; ++w) acc += RED[w];
                loss_hist[kb] = acc * inv_B;
            }
        }
        __syncthreads();
    }
    for (int p = tid; p < P; p += nthreads) {
        params[p] = PAR[p];
        if (pl.mv_lds) { exp_avg[p] = MM[p]; exp_avg_sq[p] = VV[p]; }
    }
}

int cv_kit(const CvaeK &k) {
    const int a = k.d + k.c + 1, b = k.lat + k.c + 1;
    const int ki = ((a > b ? a : b) + 3) / 4;
    return ki <= 2 ? 2 : (ki <= 4 ? 4 : 8);
}

bool make_cv_plan(const CvaeK &k, int64_t batch, CvPlan *out) {
    if (k.enc.nh != 1 || k.d > 16 || k.d + k.c > 31 || k.lat > 8 || k.lat + k.c > 31 || k.enc.nout[0] > 32) return false;
    if (k.enc.nout[0] > 16 && cv_kit(k) > 4) return false;          // the policy measured for the flows
    if (batch < 1 || batch > 16 * kRcMaxWaves) return false;
    CvPlan p;
    std::memset(&p, 0, sizeof(p));
    p.W = (int)((batch + 15) / 16);
    p.P = k.enc.npn + k.dec.npn;
    p.stg_floats = p.P + kDump;
    p.tt_floats = (2 * (cv_kit(k) > 4 ? 2 : 1) + 6) * 16 * TS;
    for (int mv = 1; mv >= 0; --mv) {
        int f = 0;
        p.oPAR = f; f += p.P;
        p.oM = f; p.oV = f;
        if (mv) { p.oM = f; f += p.P; p.oV = f; f += p.P; }
        p.oSTG = f; f += p.W * p.stg_floats;
        p.oRED = f; f += kMaxWaves;
        p.oTT = f; f += p.W * p.tt_floats;
        p.total_floats = f;
        p.mv_lds = mv;
        if ((size_t)f * sizeof(float) <= kLdsMax) { *out = p; return true; }
    }
    return false;
}

struct CvArgs {
    float *params; const float *x, *c; const int64_t *perm; const float *eps; int64_t n, batch_size; float klw;
    float *loss_hist, *exp_avg, *exp_avg_sq; double lr, beta1, beta2, eps_adam, wd; int64_t first_step;
};

template <int MT, int KIT, int ACT, int WMAX>
int launch_cv_w(hipStream_t st, const CvaeK &k, const CvPlan &p, const CvArgs &a) {
    auto kern = k_cvae_fit_resident<MT, KIT, ACT, WMAX>;
    static std::atomic<uint64_t> attr_done{0};
    const int rc = allow_big_lds(reinterpret_cast<const void *>(kern), (int)kLdsMax, attr_done);
    if (rc) return rc;
    {
        KernelTimer timer(st, RNVP_PROFILE_TRAIN);
        hipLaunchKernelGGL(kern, dim3(1), dim3(64 * WMAX), (size_t)p.total_floats * sizeof(float), st, k, p, a.params, a.x, a.c, a.perm,
                           a.eps, a.n, a.batch_size, a.klw, a.loss_hist, a.exp_avg, a.exp_avg_sq, a.lr, a.beta1, a.beta2, a.eps_adam,
                           a.wd, std::pow(a.beta1, (double)a.first_step), std::pow(a.beta2, (double)a.first_step));
    }
    RNVP_HIP_TRY(hipGetLastError());
    return RNVP_OK;
}

template <int MT, int KIT>
int launch_cv(hipStream_t st, const CvaeK &k, const CvPlan &p, const CvArgs &a) {
    const bool th = k.enc.act == RNVP_ACT_TANH;
    if (p.W <= 4)
        return th ? launch_cv_w<MT, KIT, RNVP_ACT_TANH, 4>(st, k, p, a) : launch_cv_w<MT, KIT, RNVP_ACT_RELU, 4>(st, k, p, a);
    return th ? launch_cv_w<MT, KIT, RNVP_ACT_TANH, kRcMaxWaves>(st, k, p, a) : launch_cv_w<MT, KIT, RNVP_ACT_RELU, kRcMaxWaves>(st, k, p, a);
}

template <int MT>
int launch_cv_kit(hipStream_t st, const CvaeK &k, const CvPlan &p, const CvArgs &a) {
    const int kit = cv_kit(k);
    if (kit == 2) return launch_cv<MT, 2>(st, k, p, a);
    if (kit == 4) return launch_cv<MT, 4>(st, k, p, a);
    if constexpr (MT == 1) return launch_cv<MT, 8>(st, k, p, a);       // two hidden tiles: one input tile only (make_cv_plan)
    return RNVP_EUNSUPPORTED;
}

}  // namespace


bool cvae_fits(const CvaeK &k, int family, int64_t batch_size) {
    if (family == RNVP_FAMILY_VALU) return false;
    CvPlan p;
    return make_cv_plan(k, batch_size, &p);
}

int cvae_fit_epoch(hipStream_t st, const CvaeK &k, float *params, const float *x, const float *c, const int64_t *perm,
                   const float *eps, int64_t n, int64_t batch_size, float kl_weight, float *loss_hist, float *exp_avg,
                   float *exp_avg_sq, double lr, double beta1, double beta2, double adam_eps, double weight_decay, int64_t first_step) {
    if (n == 0) return RNVP_OK;
    CvPlan p;
    if (!make_cv_plan(k, batch_size, &p)) return RNVP_EUNSUPPORTED;
    const CvArgs a{params, x, c, perm, eps, n, batch_size, kl_weight, loss_hist, exp_avg, exp_avg_sq, lr, beta1, beta2, adam_eps,
                   weight_decay, first_step};
    if (k.enc.nout[0] <= 16) return launch_cv_kit<1>(st, k, p, a);
    return launch_cv_kit<2>(st, k, p, a);
}

}  // namespace resident
}  // namespace rnvp
