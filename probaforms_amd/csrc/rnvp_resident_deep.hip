// rnvp_resident_deep.hip -- resident fit (rnvp_resident.hip) for flows with two or three hidden layers of one or two tiles each.
#include "rnvp_resident_dev.h"

namespace rnvp {
namespace resident {
namespace {

// ---- two or three hidden layers of at most 16 units each (hidden=(10, 10), (16, 16, 16), ...): the same chain, one GEMM longer per
// hidden layer.  Without this form such a flow falls to the any-shape kernels' five launches per step (240 us at batch 32):
// a 14x cliff next to hidden=(10,).  A hidden -> hidden Linear is one 16x16 tile: its D operand is the next Linear's B
// operand exactly like the last Linear's, its weight gradient one more contraction over the rows, its bias gradient a DPP
// row sum.  Fragments are loaded layer by layer (no look-ahead: three Linears' worth per net would not fit 256 registers).
// MTH = 2: every hidden layer is carried as two tiles (up to 32 units; the docstring network hidden=(10, 20, 15) of
// realnvp.py:22-38): a hidden -> hidden Linear becomes 2 x 2 tile products, padding tiles multiply zeros.
template <int NH, int KIT, int ACT, int WMAX, int DT, int MTH>
__global__ void __launch_bounds__(64 * WMAX)
k_fit_resident_deep(KShape s, RcPlan pl, float *__restrict__ params, const uint8_t *__restrict__ masks, const float *__restrict__ x,
                    const float *__restrict__ c, const int64_t *__restrict__ perm, int64_t n, int64_t batch, int64_t n_epochs,
                    float *__restrict__ loss_hist, float *__restrict__ exp_avg, float *__restrict__ exp_avg_sq, double lr, double beta1,
                    double beta2, double eps, double wd, double b1t, double b2t) {
    constexpr int NIT = KIT > 4 ? 2 : 1;
    constexpr int KXT = KIT < DT ? KIT : DT;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, nthreads = blockDim.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, r = lane & 15, i = r;
    const int d = s.d, cd = s.c, L = s.L, P = pl.P, nin0 = d + cd, npn = s.npn;
    const int pi = 4 * (i & 3) + (i >> 2);
    float *PAR = lds + pl.oPAR, *MM = lds + pl.oM, *VV = lds + pl.oV, *STG = lds + pl.oSTG + (size_t)wave * pl.stg_floats;
    float *RED = lds + pl.oRED;
    f4 *XS = reinterpret_cast<f4 *>(lds + pl.oXS + (size_t)wave * pl.xs_floats);
    float *TT = lds + pl.oTT + (size_t)wave * pl.tt_floats;
    float *T_in = TT, *T_g = TT + NIT * 16 * TS, *T_h = T_g + 2 * MTH * 16 * TS;  // [input tiles][g: net, tile][h: net, tile]
    for (int e = tid; e < pl.total_floats; e += nthreads) lds[e] = 0.f;
    __syncthreads();
    for (int p = tid; p < P; p += nthreads) {
        PAR[p] = params[p];
        if (pl.mv_lds) { MM[p] = exp_avg[p]; VV[p] = exp_avg_sq[p]; }
    }
    __syncthreads();
    const float prior_c = 0.5f * (float)d * kLog2Pi;
    // perm holds n_epochs permutations of n rows back to back; every epoch is cut into the same batches (the last one ragged)
    const int64_t nb_e = (n + batch - 1) / batch, nb = nb_e * n_epochs;
    auto batch_at = [&](int64_t kb, int64_t &s0, int64_t &rows) {
        const int64_t ep = kb / nb_e, k = kb - ep * nb_e;
        s0 = ep * n + k * batch;
        rows = (n - k * batch < batch) ? n - k * batch : batch;
    };

    // ---- per-lane constants.  Linear k maps nin_k -> nout_k (k = 0: the net input; k = NH: the d outputs) ----
    int gF[NH + 1], gB[NH + 1], gT[NH + 1][4], sS[NH + 1][4], sS0[NIT][4], sBk[NH + 1][4];      // tile (0, 0); + tile strides below
    f4 hm[NH][MTH];
    const int dump = npn + (lane & 15);
#pragma unroll
    for (int k = 0; k <= NH; ++k) {
        const int nin = s.nin[k], nout = s.nout[k], wo = s.woff[k], bo = s.boff[k];
        gF[k] = wo + pi * nin + q;                       // + 4e : W_k[pi][4e + q]           (forward)
        gB[k] = bo + q;                                  // + 4e
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            gT[k][e] = wo + (4 * e + q) * nin + pi;      // W_k[4e + q][pi]                  (gradient of Linear k's input)
            sS[k][e] = (4 * q + e < nout && i < nin) ? wo + (4 * q + e) * nin + i : dump;        // d W_k[4q + e][i]
            sBk[k][e] = (4 * e + q < nout) ? bo + 4 * e + q : dump;                               // d b_k[4e + q] (lanes r == 0)
        }
        if (k < NH)
#pragma unroll
            for (int m = 0; m < MTH; ++m)
#pragma unroll
                for (int e = 0; e < 4; ++e) hm[k][m][e] = 16 * m + 4 * e + q < nout ? 1.f : 0.f;
    }
    // stage position of element (unit u of Linear k's outputs, column j) of d W_k; of d b_k[u]
    auto stage_w = [&](int k, int u, int j) -> int { return (u < s.nout[k] && j < s.nin[k]) ? s.woff[k] + u * s.nin[k] + j : dump; };
    auto stage_b = [&](int k, int u) -> int { return u < s.nout[k] ? s.boff[k] + u : dump; };
#pragma unroll
    for (int nt = 0; nt < NIT; ++nt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int u = 4 * q + e, j = 16 * nt + i;
            sS0[nt][e] = u >= s.nout[0] ? dump : (j < nin0 ? s.woff[0] + u * nin0 + j : (j == nin0 ? s.boff[0] + u : dump));
        }
    uint64_t mbits = 0;
    for (int l = 0; l < L; ++l)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = 4 * e + q;
            // masks == NULL: the caller declared the reference's alternating pattern (rnvp_shape::alt_masks 1 or 2)
            const bool mk = masks ? (j < d && masks[l * d + j] != 0) : (((j + l + (s.alt == 2 ? 1 : 0)) & 1) != 0);   // (no read past the [L, d] table)
            if (j >= d || mk) mbits |= 1ull << (4 * l + e);
        }
    bool xok[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) xok[e] = 4 * e + q < d;

    auto row_of = [&](int64_t kb) -> int64_t {
        if (kb >= nb) return -1;
        int64_t s0, rows;
        batch_at(kb, s0, rows);
        const int64_t rr = (int64_t)wave * 16 + r;
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
    // both nets of one layer, interleaved: hidden activations hh[net][k][tile]; outputs o[1] (s) and, if asked for, o[0] (t)
    auto nets_fwd = [&](const float *pl0, const f4 (&in)[NIT], f4 (&hh)[2][NH][MTH], f4 (&o)[2], auto need_t) {
        constexpr int N0 = decltype(need_t)::value ? 0 : 1;
#pragma unroll
        for (int m = 0; m < MTH; ++m) {
            f4 acc[2];
#pragma unroll
            for (int net = 0; net < 2; ++net)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[net][e] = pl0[net * npn + gB[0] + 16 * m + 4 * e];
#pragma unroll
            for (int k = 0; k < KIT; ++k)
#pragma unroll
                for (int net = 0; net < 2; ++net)
                    acc[net] = mfma16(pl0[net * npn + gF[0] + m * 16 * nin0 + 4 * k], in[k >> 2][k & 3], acc[net]);
#pragma unroll
            for (int net = 0; net < 2; ++net)
#pragma unroll
                for (int e = 0; e < 4; ++e) hh[net][0][m][e] = actf<ACT>(acc[net][e]) * hm[0][m][e];
        }
#pragma unroll
        for (int k = 1; k < NH; ++k) {
            const int nin = s.nin[k];
#pragma unroll
            for (int m = 0; m < MTH; ++m) {
                f4 acc[2];
#pragma unroll
                for (int net = 0; net < 2; ++net)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[net][e] = pl0[net * npn + gB[k] + 16 * m + 4 * e];
#pragma unroll
                for (int mi = 0; mi < MTH; ++mi)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int net = 0; net < 2; ++net)
                            acc[net] = mfma16(pl0[net * npn + gF[k] + m * 16 * nin + 16 * mi + 4 * e], hh[net][k - 1][mi][e], acc[net]);
#pragma unroll
                for (int net = 0; net < 2; ++net)
#pragma unroll
                    for (int e = 0; e < 4; ++e) hh[net][k][m][e] = actf<ACT>(acc[net][e]) * hm[k][m][e];
            }
        }
#pragma unroll
        for (int net = N0; net < 2; ++net)
#pragma unroll
            for (int e = 0; e < 4; ++e) o[net][e] = pl0[net * npn + gB[NH] + 4 * e];
#pragma unroll
        for (int mi = 0; mi < MTH; ++mi)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int net = N0; net < 2; ++net) o[net] = mfma16(pl0[net * npn + gF[NH] + 16 * mi + 4 * e], hh[net][NH - 1][mi][e], o[net]);
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
        const int nw = (rows + 15) >> 4;
        f4 xq = nxq, cin[NIT];
#pragma unroll
        for (int nt = 0; nt < NIT; ++nt) cin[nt] = ncq[nt];
        load_rows(src_next, nxq, ncq);
        src_next = row_of(kb + 2);
        if (wave < nw) {
            const bool valid = wave * 16 + r < rows;
            float ld = 0.f;
            for (int l = 0; l < L; ++l) {                                  // forward (realnvp.py:91-101, nflow.py:107-117)
                const uint32_t mb = (uint32_t)(mbits >> (4 * l)) & 15u;
                XS[l * 64 + lane] = xq;
                f4 in[NIT];
                in[0] = cin[0];
#pragma unroll
                for (int e = 0; e < DT; ++e) in[0][e] = ((mb >> e) & 1u) ? xq[e] + cin[0][e] : cin[0][e];
                if (NIT > 1) in[NIT - 1] = cin[NIT - 1];
                f4 hh[2][NH][MTH], o[2];
                nets_fwd(PAR + (size_t)l * 2 * npn, in, hh, o, std::true_type{});
#pragma unroll
                for (int e = 0; e < DT; ++e) {
                    const bool mk = (mb >> e) & 1u;
                    const float xn = fmaf(xq[e], exp_acc(o[1][e]), o[0][e]);
                    xq[e] = mk ? xq[e] : xn;
                    ld += mk ? 0.f : o[1][e];
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
                if (lane == 0) RED[wave] = v;
            }
            const float gld = valid ? -inv_B : 0.f;
            f4 cinT[NIT];
#pragma unroll
            for (int nt = 0; nt < NIT; ++nt)
#pragma unroll
                for (int e = 0; e < 4; ++e) cinT[nt][e] = (16 * nt + 4 * e + q == nin0) ? 1.f : cin[nt][e];
            if (NIT > 1) { wfence(); tile_put(T_in + (NIT - 1) * 16 * TS, cinT[NIT - 1], q, r); }
            for (int l = L - 1; l >= 0; --l) {                             // backward
                const float *pl0 = PAR + (size_t)l * 2 * npn;
                float *stg0 = STG + (size_t)l * 2 * pl.stg_net;
                const uint32_t mb = (uint32_t)(mbits >> (4 * l)) & 15u;
                xq = XS[l * 64 + lane];
                f4 in[NIT], in0T;
                in[0] = cin[0]; in0T = cinT[0];
#pragma unroll
                for (int e = 0; e < DT; ++e) {
                    const bool mk = (mb >> e) & 1u;
                    in[0][e] = mk ? xq[e] + cin[0][e] : cin[0][e];
                    in0T[e] = mk ? xq[e] + cinT[0][e] : cinT[0][e];
                }
                if (NIT > 1) in[NIT - 1] = cin[NIT - 1];
                wfence();
                tile_put(T_in, in0T, q, r);
                f4 hh[2][NH][MTH], o[2];
                nets_fwd(pl0, in, hh, o, std::false_type{});
                f4 es = f4{0.f, 0.f, 0.f, 0.f}, g[2][MTH];      // g: gradient at the outputs of the Linear in hand, per tile
#pragma unroll
                for (int net = 0; net < 2; ++net)
#pragma unroll
                    for (int m = 0; m < MTH; ++m) g[net][m] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < DT; ++e) {
                    const bool mk = (mb >> e) & 1u;
                    es[e] = exp_acc(o[1][e]);
                    g[1][0][e] = mk ? 0.f : fmaf(gy[e] * xq[e], es[e], gld);
                    g[0][0][e] = mk ? 0.f : gy[e];
                }
                // Linear k = NH .. 1: bias gradient (row sums), weight gradient g^T . h_{k-1}, gradient of h_{k-1}
#pragma unroll
                for (int k = NH; k >= 1; --k) {
                    const int MG = (k == NH) ? 1 : MTH;                    // tiles of g (the d outputs are one tile)
                    const int ke = (k == NH) ? KXT : 4;                    // slots of a g tile that hold units
                    const int nin = s.nin[k];
#pragma unroll
                    for (int net = 0; net < 2; ++net)
#pragma unroll
                        for (int mg = 0; mg < MTH; ++mg)
                            if (mg < MG)
#pragma unroll
                                for (int e = 0; e < 4; ++e)
                                    if (e < ke) {
                                        const float v = row16_sum(g[net][mg][e]);
                                        if (r == 0) stg0[net * pl.stg_net + (MTH == 1 ? sBk[k][e] : stage_b(k, 16 * mg + 4 * e + q))] = v;
                                    }
                    wfence();
#pragma unroll
                    for (int net = 0; net < 2; ++net)
#pragma unroll
                        for (int m = 0; m < MTH; ++m) {
                            if (m < MG) tile_put(T_g + (net * MTH + m) * 16 * TS, g[net][m], q, r);
                            tile_put(T_h + (net * MTH + m) * 16 * TS, hh[net][k - 1][m], q, r);
                        }
                    f4 gh[2][MTH];
#pragma unroll
                    for (int mi = 0; mi < MTH; ++mi) {
                        gh[0][mi] = f4{0.f, 0.f, 0.f, 0.f}; gh[1][mi] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int mg = 0; mg < MTH; ++mg)
                            if (mg < MG)
#pragma unroll
                                for (int e = 0; e < 4; ++e)
                                    if (e < ke)
#pragma unroll
                                        for (int net = 0; net < 2; ++net)
                                            gh[net][mi] = mfma16(pl0[net * npn + gT[k][e] + mg * 16 * nin + 16 * mi], g[net][mg][e], gh[net][mi]);
                    }
                    wfence();
#pragma unroll
                    for (int mg = 0; mg < MTH; ++mg) {
                        if (mg >= MG) continue;
                        float gT_[2][4];
                        tile_get(T_g + (0 * MTH + mg) * 16 * TS, q, i, gT_[0]);
                        tile_get(T_g + (1 * MTH + mg) * 16 * TS, q, i, gT_[1]);
#pragma unroll
                        for (int mi = 0; mi < MTH; ++mi) {
                            float hT_[2][4];
                            tile_get(T_h + (0 * MTH + mi) * 16 * TS, q, i, hT_[0]);
                            tile_get(T_h + (1 * MTH + mi) * 16 * TS, q, i, hT_[1]);
                            f4 dw[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                                for (int net = 0; net < 2; ++net) dw[net] = mfma16(gT_[net][ks], hT_[net][ks], dw[net]);      // [unit of k][unit of k-1]
#pragma unroll
                            for (int net = 0; net < 2; ++net)
#pragma unroll
                                for (int e = 0; e < 4; ++e)
                                    stg0[net * pl.stg_net + (MTH == 1 ? sS[k][e] : stage_w(k, 16 * mg + 4 * q + e, 16 * mi + i))] = dw[net][e];
                        }
                    }
#pragma unroll
                    for (int net = 0; net < 2; ++net)
#pragma unroll
                        for (int mi = 0; mi < MTH; ++mi)
#pragma unroll
                            for (int e = 0; e < 4; ++e) g[net][mi][e] = gh[net][mi][e] * dactf<ACT>(hh[net][k - 1][mi][e]) * hm[k - 1][mi][e];
                }
                // Linear 0: weight + bias gradient against the input tile(s); input gradient for the x part
                f4 gin[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int m = 0; m < MTH; ++m)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int net = 0; net < 2; ++net) gin[net] = mfma16(pl0[net * npn + gT[0][e] + m * 16 * nin0], g[net][m][e], gin[net]);
                wfence();
#pragma unroll
                for (int net = 0; net < 2; ++net)
#pragma unroll
                    for (int m = 0; m < MTH; ++m) tile_put(T_g + (net * MTH + m) * 16 * TS, g[net][m], q, r);
                wfence();
                float inT[NIT][4];
#pragma unroll
                for (int nt = 0; nt < NIT; ++nt) tile_get(T_in + nt * 16 * TS, q, i, inT[nt]);
#pragma unroll
                for (int m = 0; m < MTH; ++m) {
                    float g0T[2][4];
                    tile_get(T_g + (0 * MTH + m) * 16 * TS, q, i, g0T[0]);
                    tile_get(T_g + (1 * MTH + m) * 16 * TS, q, i, g0T[1]);
                    f4 dw0[2][NIT];
#pragma unroll
                    for (int net = 0; net < 2; ++net)
#pragma unroll
                        for (int nt = 0; nt < NIT; ++nt) dw0[net][nt] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                        for (int net = 0; net < 2; ++net)
#pragma unroll
                            for (int nt = 0; nt < NIT; ++nt) dw0[net][nt] = mfma16(g0T[net][ks], inT[nt][ks], dw0[net][nt]);
#pragma unroll
                    for (int net = 0; net < 2; ++net)
#pragma unroll
                        for (int nt = 0; nt < NIT; ++nt)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                int pos;
                                if (MTH == 1) pos = sS0[nt][e];
                                else {
                                    const int u = 16 * m + 4 * q + e, j = 16 * nt + i;
                                    pos = j == nin0 ? stage_b(0, u) : stage_w(0, u, j);
                                }
                                stg0[net * pl.stg_net + pos] = dw0[net][nt][e];
                            }
                }
#pragma unroll
                for (int e = 0; e < DT; ++e) {
                    const bool mk = (mb >> e) & 1u;
                    gy[e] = xok[e] ? (mk ? gy[e] + (gin[1][e] + gin[0][e]) : gy[e] * es[e]) : 0.f;
                }
            }
        }
        __syncthreads();
        {
            const AdamK a = step_adam(lr, beta1, beta2, eps, wd, b1t, b2t);
            adam_phase(lds + pl.oSTG, pl.stg_floats, pl.stg_net, npn, P, nw, PAR, MM, VV, pl.mv_lds != 0, exp_avg, exp_avg_sq, a, tid, nthreads);
            if (tid == 0) {
                float acc = 0.f;
                for (int w = 0; w < nw; ++w) acc += RED[w];
                loss_hist[kb] = -acc * inv_B;
            }
        }
        __syncthreads();
    }
    for (int p = tid; p < P; p += nthreads) {
        params[p] = PAR[p];
        if (pl.mv_lds) { exp_avg[p] = MM[p]; exp_avg_sq[p] = VV[p]; }
    }
}

template <int NH, int KIT, int ACT, int WMAX, int DT, int MTH>
int launch_deep_m(hipStream_t st, const KShape &k, const RcPlan &p, const EpochArgs &a) {
    auto kern = k_fit_resident_deep<NH, KIT, ACT, WMAX, DT, MTH>;
    static std::atomic<uint64_t> attr_done{0};
    const int rc = allow_big_lds(reinterpret_cast<const void *>(kern), (int)kLdsMax, attr_done);
    if (rc) return rc;
    note_dispatch(RNVP_PROFILE_TRAIN, "k_fit_resident_deep", RNVP_VARIANT_RESIDENT, 1, WMAX, 1, RNVP_PREC_F32, a.n);
    note_launches(RNVP_PROFILE_TRAIN, 1);
    {
        KernelTimer timer(st, RNVP_PROFILE_TRAIN);
        hipLaunchKernelGGL(kern, dim3(1), dim3(64 * WMAX), (size_t)p.total_floats * sizeof(float), st, k, p, a.params, a.masks, a.x, a.c,
                           a.perm, a.n, a.batch_size, a.n_epochs, a.loss_hist, a.exp_avg, a.exp_avg_sq, a.lr, a.beta1, a.beta2, a.eps, a.wd,
                           std::pow(a.beta1, (double)a.first_step), std::pow(a.beta2, (double)a.first_step));
    }
    RNVP_HIP_TRY(hipGetLastError());
    return RNVP_OK;
}

template <int NH, int KIT, int ACT, int WMAX, int DT>
int launch_deep_w(hipStream_t st, const KShape &k, const RcPlan &p, const EpochArgs &a) {
    bool wide = false;                       // a hidden layer of more than one tile: every hidden layer runs as two
    for (int i = 0; i < k.nh; ++i) wide = wide || k.nout[i] > 16;
    return wide ? launch_deep_m<NH, KIT, ACT, WMAX, DT, 2>(st, k, p, a) : launch_deep_m<NH, KIT, ACT, WMAX, DT, 1>(st, k, p, a);
}

template <int NH, int KIT, int DT>
int launch_deep_d(hipStream_t st, const KShape &k, const RcPlan &p, const EpochArgs &a) {
    const bool th = k.act == RNVP_ACT_TANH;
    if (p.W <= 4)
        return th ? launch_deep_w<NH, KIT, RNVP_ACT_TANH, 4, DT>(st, k, p, a) : launch_deep_w<NH, KIT, RNVP_ACT_RELU, 4, DT>(st, k, p, a);
    return th ? launch_deep_w<NH, KIT, RNVP_ACT_TANH, kRcMaxWaves, DT>(st, k, p, a)
              : launch_deep_w<NH, KIT, RNVP_ACT_RELU, kRcMaxWaves, DT>(st, k, p, a);
}

template <int NH>
int launch_deep_n(hipStream_t st, const KShape &k, const RcPlan &p, const EpochArgs &a) {
    const int kit = rc_kit(k);
    if (kit == 2) return launch_deep_d<NH, 2, 4>(st, k, p, a);
    if (kit == 4) return launch_deep_d<NH, 4, 4>(st, k, p, a);
    return launch_deep_d<NH, 8, 4>(st, k, p, a);
}

}  // namespace

int launch_deep(hipStream_t st, const KShape &k, const RcPlan &p, const EpochArgs &a) {
    return k.nh == 2 ? launch_deep_n<2>(st, k, p, a) : launch_deep_n<3>(st, k, p, a);
}

}  // namespace resident
}  // namespace rnvp
