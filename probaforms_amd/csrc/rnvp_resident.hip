// rnvp_resident.hip -- "resident" fit: the batch loop of a SMALL flow at a SMALL batch size in ONE persistent workgroup (gfx950).
//
// The reference's defaults -- hidden=(10,), 8 layers, batch_size=32 (/root/reference/probaforms/models/realnvp.py:161-176)
// -- are a flow of ~1000 parameters stepped on a few dozen rows: a step is a chain of ~50 tiny dependent GEMMs, and the
// three launches of the general path (weight re-pack, loss + gradient, reduce + Adam) cost more than the arithmetic
// (41 us per step for d = 2, h = 10).  A CU's 160 KB of LDS holds such a model whole, so the batch loop of RealNVP.fit
// (realnvp.py:235-254) runs here as ONE launch per epoch -- or per fit: rnvp_fit_epochs hands over several epochs'
// permutations at once:
//   * the flat parameters (reference order) live in LDS for the whole launch, and so do Adam's moments when they fit;
//   * wave w owns rows 16w .. 16w+15 of every batch and runs its forward / backward chain register to register on
//     v_mfma_f32_16x16x4_f32 (the per-lane layout is described in rnvp_resident_dev.h): no workgroup barrier inside a step;
//   * the weight gradients of the wave's 16 rows are a third kind of MFMA contraction (over the rows, operands
//     transposed through wave-private LDS tiles), written to the wave's stage in the reference's flat order;
//   * one barrier, then all threads add the stages in wave order (deterministic), apply Adam in place in LDS and write
//     the batch loss; one more barrier and the next batch starts (its rows were requested a step ahead).  Parameters and
//     moments go back to HBM once, at the end of the launch.
// A single wave issues one VALU instruction per 4 cycles, so what bounds a step here is its INSTRUCTION COUNT: everything
// that does not depend on the data -- gather offsets, padding masks, stage offsets, mask bits -- is computed once per
// launch, the next layer's weight fragments are requested while the current layer computes, and (up to four waves) the
// forward keeps the hidden activations for the backward.  Measured (scripts/resident_time.py, profiles/r03_resident_time.txt):
// 15 us per step for the defaults against 41 -- and 10.8 in the net-split form that takes batches of at most 64 rows
// (rnvp_resident_ns.hip: a wave per row tile AND net, weight gradients on helper waves).
// This file: ONE hidden layer (one or two tiles), one wave per row tile -- batches of 65 to 128 rows, and smaller ones whose
// records do not fit the net-split form's LDS plan.  Two or three hidden layers: rnvp_resident_deep.hip; the conditional
// VAE: cvae_resident.hip.
// Same arithmetic per element as the other kernel families (tanh through exp2 / rcp, torch.optim.Adam as separately
// rounded operations); the summation ORDER over rows and hidden units differs, so results agree with them to rounding,
// not bit for bit -- run to run this path is bit-reproducible.
#include "rnvp_resident_dev.h"

namespace rnvp {
namespace resident {
namespace {

// MT hidden tiles, KIT k-steps of the net input (4 KIT >= d + cdim + 1), WMAX waves, DT slots of x that hold features
// (4 DT >= d: 2-d data needs one of the four); SAVE: the forward keeps every layer's hidden activations and s in a
// lane-private LDS area and the backward reads them back instead of recomputing both nets (when that area fits)
// threads of a launch: a one-tile-per-layer kernel of up to four row waves gets four more waves that only share the Adam phase
// (where the register budget allows it; they wait at the two barriers of a step meanwhile)
template <int MT, int KIT, int WMAX, bool SAVE> constexpr int rc_threads() { return MT == 1 && (KIT <= 4 || SAVE) && WMAX == 4 ? 512 : 64 * WMAX; }

template <int MT, int KIT, int ACT, int WMAX, int DT, bool SAVE>
__global__ void __launch_bounds__((rc_threads<MT, KIT, WMAX, SAVE>()))
k_fit_resident_rc(KShape s, RcPlan pl, float *__restrict__ params, const uint8_t *__restrict__ masks, const float *__restrict__ x,
                  const float *__restrict__ c, const int64_t *__restrict__ perm, int64_t n, int64_t batch, int64_t n_epochs,
                  float *__restrict__ loss_hist, float *__restrict__ exp_avg, float *__restrict__ exp_avg_sq, double lr, double beta1,
                  double beta2, double eps, double wd, double b1t, double b2t) {
    constexpr int NIT = KIT > 4 ? 2 : 1;               // 16-element tiles of the net input
    constexpr int KXT = KIT < DT ? KIT : DT;           // k-steps over the features of x
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, nthreads = blockDim.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, r = lane & 15, i = r;
    const int d = s.d, cd = s.c, L = s.L, P = pl.P, h = s.nout[0], nin0 = d + cd, npn = s.npn;
    const int w0 = s.woff[0], b0 = s.boff[0], w1 = s.woff[1], b1o = s.boff[1];
    const int pi = 4 * (i & 3) + (i >> 2);                // the element a fragment's row i stands for
    float *PAR = lds + pl.oPAR, *MM = lds + pl.oM, *VV = lds + pl.oV, *STG = lds + pl.oSTG + (size_t)wave * pl.stg_floats;
    float *RED = lds + pl.oRED;
    f4 *XS = reinterpret_cast<f4 *>(lds + pl.oXS + (size_t)wave * pl.xs_floats);
    f4 *HS = XS + (size_t)L * 64;                         // SAVE: [layer][t tiles | s tiles | s out][lane]
    float *TT = lds + pl.oTT + (size_t)wave * pl.tt_floats;
    float *T_in = TT, *T_go = TT + NIT * 16 * TS, *T_h = T_go + 2 * 16 * TS;      // [input tiles][g_out t, s][h, g_pre of t; of s]
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

    // ---- per-lane constants: gather offsets (floats, relative to a net's parameter block), padding masks, stage offsets ----
    const int gW1 = w0 + pi * nin0 + q;                   // + m 16 nin0 + 4k : W1[16m + pi][4k + q]          (GEMM1)
    const int gW2 = w1 + pi * h + q;                      // + 16m + 4e      : W2[pi][16m + 4e + q]          (GEMM2)
    const int gW1t = w0 + q * nin0 + pi;                  // + (16m + 4e) nin0 : W1[16m + 4e + q][pi]        (g_in)
    const int gB1 = b0 + q, gB2 = b1o + q;                // + 16m + 4e / + 4e
    int gW2t[4];                                          // + 16m           : W2[4e + q][16m + pi]          (g_h)
#pragma unroll
    for (int e = 0; e < 4; ++e) gW2t[e] = w1 + (4 * e + q) * h + pi;
    f4 hm[MT];                                            // 1 for the real hidden units of a tile, 0 for its padding
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) hm[m][e] = 16 * m + 4 * e + q < h ? 1.f : 0.f;
    uint64_t mbits = 0;                                   // bit 4l + e: mask[l][4e + q]; padding features count as masked (passed through)
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
    // stage offsets (relative to a net's stage block) of the weight-gradient tiles' D registers
    int sS2[MT][4], sS1[MT][NIT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int fo = 4 * q + e, hid_n = 16 * m + i, hid_m = 16 * m + 4 * q + e;
            sS2[m][e] = (fo < d && hid_n < h) ? w1 + fo * h + hid_n : npn + (lane & 15);                    // d W2[fo][hid]
#pragma unroll
            for (int nt = 0; nt < NIT; ++nt) {
                const int j = 16 * nt + i;
                sS1[m][nt][e] = hid_m >= h ? npn + (lane & 15) : (j < nin0 ? w0 + hid_m * nin0 + j : (j == nin0 ? b0 + hid_m : npn + (lane & 15)));
            }
        }

    // rows of a batch for this lane (zeros past d / cdim / the batch): x, and the condition placed behind x in the net input
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
    // A fragments of one layer, both nets: forward (biases, W1, W2) and backward (W2^T for g_h, W1^T for g_in)
    struct FwdW { float b1[2][MT][4], a1[2][MT][KIT], b2[2][4], a2[2][MT][4]; };
    struct BwdW { float a2t[2][MT][KXT], a1t[2][MT][4]; };
    auto load_fwd = [&](int l, FwdW &w) {
#pragma unroll
        for (int net = 0; net < 2; ++net) {
            const float *pn = PAR + (size_t)(2 * l + net) * npn;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { w.b1[net][m][e] = pn[gB1 + 16 * m + 4 * e]; w.a2[net][m][e] = pn[gW2 + 16 * m + 4 * e]; }
#pragma unroll
                for (int k = 0; k < KIT; ++k) w.a1[net][m][k] = pn[gW1 + m * 16 * nin0 + 4 * k];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) w.b2[net][e] = pn[gB2 + 4 * e];
        }
    };
    auto load_bwd = [&](int l, BwdW &w) {
#pragma unroll
        for (int net = 0; net < 2; ++net) {
            const float *pn = PAR + (size_t)(2 * l + net) * npn;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
#pragma unroll
                for (int e = 0; e < KXT; ++e) w.a2t[net][m][e] = pn[gW2t[e] + 16 * m];
#pragma unroll
                for (int e = 0; e < 4; ++e) w.a1t[net][m][e] = pn[gW1t + (16 * m + 4 * e) * nin0];
            }
        }
    };
    // both nets of one layer, their two independent chains interleaved instruction by instruction: hidden activations
    // hh[net][m]; outputs o[1] of the s net and (need_t) o[0] of the t net
    auto nets_fwd = [&](const FwdW &w, const f4 (&in)[NIT], f4 (&hh)[2][MT], f4 (&o)[2], auto need_t) {
        constexpr int N0 = decltype(need_t)::value ? 0 : 1;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            f4 acc[2];
#pragma unroll
            for (int net = 0; net < 2; ++net)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[net][e] = w.b1[net][m][e];
#pragma unroll
            for (int k = 0; k < KIT; ++k)
#pragma unroll
                for (int net = 0; net < 2; ++net) acc[net] = mfma16(w.a1[net][m][k], in[k >> 2][k & 3], acc[net]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int net = 0; net < 2; ++net) hh[net][m][e] = actf<ACT>(acc[net][e]) * hm[m][e];
        }
#pragma unroll
        for (int net = N0; net < 2; ++net)
#pragma unroll
            for (int e = 0; e < 4; ++e) o[net][e] = w.b2[net][e];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int net = N0; net < 2; ++net) o[net] = mfma16(w.a2[net][m][e], hh[net][m][e], o[net]);
    };

    // two batches ahead: row indices; one batch ahead: the rows themselves
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
#ifdef RC_STAMP
        unsigned long long ts0 = __builtin_readcyclecounter(), ts1 = ts0, ts2 = ts0, ts3 = ts0, ts4 = ts0;
#endif
        if (wave < nw) {
            const bool valid = wave * 16 + r < rows;
            float ld = 0.f;
            // ---- forward (realnvp.py:91-101, nflow.py:107-117); the next layer's fragments are requested a layer ahead ----
            FwdW fw;
            load_fwd(0, fw);
            for (int l = 0; l < L; ++l) {
                const uint32_t mb = (uint32_t)(mbits >> (4 * l)) & 15u;
                XS[l * 64 + lane] = xq;                                    // layer input, for the backward (lane-private)
                f4 in[NIT];
                in[0] = cin[0];
#pragma unroll
                for (int e = 0; e < DT; ++e) in[0][e] = ((mb >> e) & 1u) ? xq[e] + cin[0][e] : cin[0][e];     // [x * mask | c]
                if (NIT > 1) in[NIT - 1] = cin[NIT - 1];
                FwdW fn;
                if (MT == 1) load_fwd(l + 1 < L ? l + 1 : l, fn);
                f4 hh[2][MT], o[2];
                nets_fwd(fw, in, hh, o, std::true_type{});
                if (SAVE) {
                    f4 *hs = HS + (size_t)l * (2 * MT + 1) * 64 + lane;
#pragma unroll
                    for (int net = 0; net < 2; ++net)
#pragma unroll
                        for (int m = 0; m < MT; ++m) hs[(net * MT + m) * 64] = hh[net][m];
                    hs[2 * MT * 64] = o[1];
                }
#pragma unroll
                for (int e = 0; e < DT; ++e) {
                    const bool mk = (mb >> e) & 1u;
                    const float xn = fmaf(xq[e], exp_acc(o[1][e]), o[0][e]);
                    xq[e] = mk ? xq[e] : xn;
                    ld += mk ? 0.f : o[1][e];
                }
                if (MT == 1) fw = fn;
                else if (l + 1 < L) load_fwd(l + 1, fw);
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
#ifdef RC_STAMP
            __builtin_amdgcn_sched_barrier(0); ts1 = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0);
#endif
            const float gld = valid ? -inv_B : 0.f;
            f4 cinT[NIT];                                  // the condition part of the input TILE: + the ones element behind it (d b1)
#pragma unroll
            for (int nt = 0; nt < NIT; ++nt)
#pragma unroll
                for (int e = 0; e < 4; ++e) cinT[nt][e] = (16 * nt + 4 * e + q == nin0) ? 1.f : cin[nt][e];
            if (NIT > 1) { wfence(); tile_put(T_in + (NIT - 1) * 16 * TS, cinT[NIT - 1], q, r); }
            // ---- backward, the two nets of a layer side by side ----
            FwdW bf;
            BwdW bb;
            if (!SAVE) load_fwd(L - 1, bf);
            load_bwd(L - 1, bb);
            for (int l = L - 1; l >= 0; --l) {
                float *stg0 = STG + (size_t)l * 2 * pl.stg_net;
                const uint32_t mb = (uint32_t)(mbits >> (4 * l)) & 15u;
                xq = XS[l * 64 + lane];
                FwdW nf;
                BwdW nbw;
                if (MT == 1) { const int lp = l > 0 ? l - 1 : 0; if (!SAVE) load_fwd(lp, nf); load_bwd(lp, nbw); }
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
                f4 hh[2][MT], o[2];
                if (SAVE) {
                    const f4 *hs = HS + (size_t)l * (2 * MT + 1) * 64 + lane;
#pragma unroll
                    for (int net = 0; net < 2; ++net)
#pragma unroll
                        for (int m = 0; m < MT; ++m) hh[net][m] = hs[(net * MT + m) * 64];
                    o[1] = hs[2 * MT * 64];
                } else {
                    nets_fwd(bf, in, hh, o, std::false_type{});
                }
                f4 es, go[2];                               // d loss / d (net output): s: (1-m)(gy x e^s + gld), t: (1-m) gy
                go[0] = f4{0.f, 0.f, 0.f, 0.f}; go[1] = f4{0.f, 0.f, 0.f, 0.f}; es = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < DT; ++e) {
                    const bool mk = (mb >> e) & 1u;
                    es[e] = exp_acc(o[1][e]);
                    go[1][e] = mk ? 0.f : fmaf(gy[e] * xq[e], es[e], gld);
                    go[0][e] = mk ? 0.f : gy[e];
                }
#pragma unroll
                for (int e = 0; e < DT; ++e) {              // the other elements of the two tiles stay at their initial zeros
                    T_go[(4 * e + q) * TS + r] = go[0][e];
                    T_go[16 * TS + (4 * e + q) * TS + r] = go[1][e];
                }
                f4 gp[2][MT], gin[2];
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    f4 gh[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                    for (int e = 0; e < KXT; ++e)
#pragma unroll
                        for (int net = 0; net < 2; ++net) gh[net] = mfma16(bb.a2t[net][m][e], go[net][e], gh[net]);
#pragma unroll
                    for (int net = 0; net < 2; ++net)
#pragma unroll
                        for (int e = 0; e < 4; ++e) gp[net][m][e] = gh[net][e] * dactf<ACT>(hh[net][m][e]) * hm[m][e];
                }
                gin[0] = f4{0.f, 0.f, 0.f, 0.f}; gin[1] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int net = 0; net < 2; ++net) gin[net] = mfma16(bb.a1t[net][m][e], gp[net][m][e], gin[net]);
#pragma unroll
                for (int net = 0; net < 2; ++net)
#pragma unroll
                    for (int e = 0; e < DT; ++e) {          // d b2
                        const float v = row16_sum(go[net][e]);
                        if (r == 0 && xok[e]) stg0[net * pl.stg_net + b1o + 4 * e + q] = v;
                    }
                // weight gradients: contractions over the tile's 16 rows through the transposition tiles
                wfence();
                float inT[NIT][4], goT[2][4];
#pragma unroll
                for (int nt = 0; nt < NIT; ++nt) tile_get(T_in + nt * 16 * TS, q, i, inT[nt]);
                tile_get(T_go, q, i, goT[0]);
                tile_get(T_go + 16 * TS, q, i, goT[1]);
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    wfence();
#pragma unroll
                    for (int net = 0; net < 2; ++net) {
                        tile_put(T_h + net * 32 * TS, hh[net][m], q, r);
                        tile_put(T_h + net * 32 * TS + 16 * TS, gp[net][m], q, r);
                    }
                    wfence();
                    float hT[2][4], gpT[2][4];
#pragma unroll
                    for (int net = 0; net < 2; ++net) {
                        tile_get(T_h + net * 32 * TS, q, i, hT[net]);
                        tile_get(T_h + net * 32 * TS + 16 * TS, q, i, gpT[net]);
                    }
                    f4 dw2[2], dw1[2][NIT];
#pragma unroll
                    for (int net = 0; net < 2; ++net) {
                        dw2[net] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int nt = 0; nt < NIT; ++nt) dw1[net][nt] = f4{0.f, 0.f, 0.f, 0.f};
                    }
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                        for (int net = 0; net < 2; ++net) {
                            dw2[net] = mfma16(goT[net][ks], hT[net][ks], dw2[net]);                        // [out feature 4q+e][hidden 16m + i]
#pragma unroll
                            for (int nt = 0; nt < NIT; ++nt)
                                dw1[net][nt] = mfma16(gpT[net][ks], inT[nt][ks], dw1[net][nt]);            // [hidden 16m + 4q+e][input 16nt + i]
                        }
#pragma unroll
                    for (int net = 0; net < 2; ++net) {
                        float *stg = stg0 + net * pl.stg_net;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            stg[sS2[m][e]] = dw2[net][e];
#pragma unroll
                            for (int nt = 0; nt < NIT; ++nt) stg[sS1[m][nt][e]] = dw1[net][nt][e];
                        }
                    }
                }
#pragma unroll
                for (int e = 0; e < DT; ++e) {
                    const bool mk = (mb >> e) & 1u;
                    gy[e] = xok[e] ? (mk ? gy[e] + (gin[1][e] + gin[0][e]) : gy[e] * es[e]) : 0.f;          // the nets see x * mask
                }
                if (MT == 1) { if (!SAVE) bf = nf; bb = nbw; }
                else if (l > 0) { if (!SAVE) load_fwd(l - 1, bf); load_bwd(l - 1, bb); }
            }
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
            printf("RCSTAMP wave %d: fwd %llu bwd %llu barrier %llu adam %llu barrier2+top %llu\n", wave, ts1 - ts0, ts2 - ts1, ts3 - ts2, ts4 - ts3,
                   (unsigned long long)__builtin_readcyclecounter() - ts4);
#endif
    }
    for (int p = tid; p < P; p += nthreads) {
        params[p] = PAR[p];
        if (pl.mv_lds) { exp_avg[p] = MM[p]; exp_avg_sq[p] = VV[p]; }
    }
}

bool make_rc_plan(const KShape &k, int64_t batch, RcPlan *out) {
    // measured against the batch-by-batch loop (scripts/resident_time.py, profiles/): one hidden tile wins everywhere
    // (21-37 vs 41-53 us per step), two win while the net input fits one tile (38 vs 49 us), beyond that the loop's
    // multi-workgroup kernels are faster (h = 64: 75 vs 37 us)
    if (k.nh < 1 || k.nh > 3 || k.d > 16 || k.d + k.c > 31 || k.L > 16) return false;
    if (k.nh == 1) {
        if (k.nout[0] > 32 || (k.nout[0] > 16 && rc_kit(k) > 4)) return false;
    } else {
        for (int i = 0; i < k.nh; ++i)          // two or three hidden layers: one or two tiles each (k_fit_resident_deep)
            if (k.nout[i] > 32) return false;
    }
    if (batch < 1 || batch > 16 * kRcMaxWaves) return false;       // up to 8 waves: two per SIMD keep 256 registers each
    RcPlan p;
    std::memset(&p, 0, sizeof(p));
    p.W = (int)((batch + 15) / 16);
    p.P = 2 * k.npn * k.L;
    p.stg_net = k.npn + kDump;
    p.stg_floats = 2 * k.L * p.stg_net;
    bool wide = false;
    for (int i = 0; i < k.nh; ++i) wide = wide || k.nout[i] > 16;
    p.tt_floats = ((rc_kit(k) > 4 ? 2 : 1) + (k.nh > 1 && wide ? 8 : 6)) * 16 * TS;
    const int mt = k.nout[0] <= 16 ? 1 : 2;
    for (int sv = (k.nh == 1 && p.W <= 4 ? 1 : 0); sv >= 0; --sv)
    for (int mv = 1; mv >= 0; --mv) {
        p.save = sv;
        p.xs_floats = k.L * 64 * 4 * (1 + (sv ? 2 * mt + 1 : 0));      // saved layer inputs (+ hidden activations and s)
        int f = 0;
        p.oPAR = f; f += p.P;
        p.oM = f; p.oV = f;
        if (mv) { p.oM = f; f += p.P; p.oV = f; f += p.P; }
        p.oSTG = f; f += p.W * p.stg_floats;
        p.oRED = f; f += kMaxWaves;
        f = (f + 3) & ~3;                       // the saved layer inputs are read and written as float4
        p.oXS = f; f += p.W * p.xs_floats;
        p.oTT = f; f += p.W * p.tt_floats;
        // (the unguarded gathers of padding lanes reach at most ~1 000 floats past a net's block: into the stages / tiles, never
        // past the allocation -- the transposition tiles alone are 1 900 floats per wave)
        p.total_floats = f;
        p.mv_lds = mv;
        if ((size_t)f * sizeof(float) <= kLdsMax) { *out = p; return true; }
    }
    return false;
}

}  // namespace

bool fits(const KShape &k, int64_t batch_size) {
    if (k.family == RNVP_FAMILY_VALU) return false;
    RcPlan rc;
    if (make_rc_plan(k, batch_size, &rc)) return true;
    return false;
}

namespace {

template <int MT, int KIT, int ACT, int WMAX, int DT, bool SAVE>
int launch_rc_s(hipStream_t st, const KShape &k, const RcPlan &p, const EpochArgs &a) {
    auto kern = k_fit_resident_rc<MT, KIT, ACT, WMAX, DT, SAVE>;
    static std::atomic<uint64_t> attr_done{0};
    const int rc = allow_big_lds(reinterpret_cast<const void *>(kern), (int)kLdsMax, attr_done);
    if (rc) return rc;
    note_dispatch(RNVP_PROFILE_TRAIN, "k_fit_resident_rc", RNVP_VARIANT_RESIDENT, 1, rc_threads<MT, KIT, WMAX, SAVE>() / 64, 1, RNVP_PREC_F32, a.n);
    note_launches(RNVP_PROFILE_TRAIN, 1);
    {
        KernelTimer timer(st, RNVP_PROFILE_TRAIN);
        hipLaunchKernelGGL(kern, dim3(1), dim3(rc_threads<MT, KIT, WMAX, SAVE>()), (size_t)p.total_floats * sizeof(float), st, k, p, a.params, a.masks, a.x, a.c,
                           a.perm, a.n, a.batch_size, a.n_epochs, a.loss_hist, a.exp_avg, a.exp_avg_sq, a.lr, a.beta1, a.beta2, a.eps, a.wd,
                           std::pow(a.beta1, (double)a.first_step), std::pow(a.beta2, (double)a.first_step));
    }
    RNVP_HIP_TRY(hipGetLastError());
    return RNVP_OK;
}

template <int MT, int KIT, int ACT, int WMAX, int DT>
int launch_rc_w(hipStream_t st, const KShape &k, const RcPlan &p, const EpochArgs &a) {
    if constexpr (WMAX <= 4) {          // make_rc_plan keeps the hidden activations for up to four waves only
        if (p.save) return launch_rc_s<MT, KIT, ACT, WMAX, DT, true>(st, k, p, a);
    }
    return launch_rc_s<MT, KIT, ACT, WMAX, DT, false>(st, k, p, a);
}

template <int MT, int KIT, int DT>
int launch_rc_d(hipStream_t st, const KShape &k, const RcPlan &p, const EpochArgs &a) {
    const bool th = k.act == RNVP_ACT_TANH;
    if (p.W <= 4)       // one wave per SIMD: 512 registers
        return th ? launch_rc_w<MT, KIT, RNVP_ACT_TANH, 4, DT>(st, k, p, a) : launch_rc_w<MT, KIT, RNVP_ACT_RELU, 4, DT>(st, k, p, a);
    return th ? launch_rc_w<MT, KIT, RNVP_ACT_TANH, kRcMaxWaves, DT>(st, k, p, a)
              : launch_rc_w<MT, KIT, RNVP_ACT_RELU, kRcMaxWaves, DT>(st, k, p, a);
}

template <int MT, int KIT>
int launch_rc(hipStream_t st, const KShape &k, const RcPlan &p, const EpochArgs &a) {
    if (k.d <= 4) return launch_rc_d<MT, KIT, 1>(st, k, p, a);      // 2-d toy data, the reference's examples: one x slot
    return launch_rc_d<MT, KIT, 4>(st, k, p, a);
}

template <int MT>
int launch_rc_kit(hipStream_t st, const KShape &k, const RcPlan &p, const EpochArgs &a) {
    const int kit = rc_kit(k);
    if (kit == 2) return launch_rc<MT, 2>(st, k, p, a);
    if (kit == 4) return launch_rc<MT, 4>(st, k, p, a);
    if constexpr (MT == 1) return launch_rc<MT, 8>(st, k, p, a);       // two hidden tiles: one input tile only (make_rc_plan)
    return RNVP_EUNSUPPORTED;
}
}  // namespace

int fit_epoch(hipStream_t st, const KShape &k, float *params, const uint8_t *masks, const float *x, const float *c,
              const int64_t *perm, int64_t n, int64_t batch_size, int64_t n_epochs, float *loss_hist, float *exp_avg, float *exp_avg_sq,
              double lr, double beta1, double beta2, double eps, double weight_decay, int64_t first_step) {
    if (n == 0 || n_epochs == 0) return RNVP_OK;
    RcPlan rcp;
    const EpochArgs a{params, masks, x, c, perm, n, batch_size, n_epochs, loss_hist, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay,
                      first_step};
#ifndef RNVP_NO_NS
    if (ns_applies(k, batch_size)) return launch_ns(st, k, a);                    // <= 64 rows: a wave per (row tile, net)
#endif
    if (make_rc_plan(k, batch_size, &rcp)) {
        if (k.nh > 1) return launch_deep(st, k, rcp, a);
        if (k.nout[0] <= 16) return launch_rc_kit<1>(st, k, rcp, a);
        return launch_rc_kit<2>(st, k, rcp, a);
    }
    return RNVP_EUNSUPPORTED;
}

}  // namespace resident
}  // namespace rnvp
