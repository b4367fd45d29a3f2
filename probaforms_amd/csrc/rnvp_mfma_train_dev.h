// rnvp_mfma_train_dev.h -- fused forward + backward of the RealNVP coupling stack on f32 MFMA
// (gfx950).  Replaces `loss = -nf.log_prob(X, C); loss.backward()`
// (/root/reference/probaforms/models/realnvp.py:246-250; the backward the reference gets from
// autograd is hand-derived in SURVEY.md 3.3).  Geometry: rnvp_mfma.h.
//
// Per wave: R tiles of 16 rows stay in registers for the whole step.
//   forward  : as rnvp_mfma.hip, additionally spilling, per layer, the pre-transform value of the
//              transformed features and exp(s) to a wave-private scratch (2*NF floats per lane).
//   backward : per layer, per hidden tile (16 units of one net):
//       GEMM1 recompute -> h = tanh(.)                       [hid x rows]   (accumulator layout)
//       g_h   = W2^T . g_out          (B operand = g_out registers, k order permuted)
//       g_pre = g_h * (1 - h^2)
//       g_in += W1^T . g_pre          (accumulates over all hidden tiles; lands on the lanes that
//                                      keep the conditioning features)
//       dW2  += h . g_out^T,  dW1|db1 += g_pre . [in | 1]^T   (contraction over ROWS: h and g_pre
//                                      are transposed through a wave-private LDS tile; the ones
//                                      column makes db1 a by-product of the same MFMA)
//     The weight-gradient accumulators live in registers across the wave's R row tiles, are then
//     stored (plain ds_write_b128; LDS float atomics measured ~1 lane/clk on gfx950) into the
//     wave's own LDS slot, and every FT hidden tiles the workgroup adds the four slots in wave
//     order into its private partial in global memory.  A second pass sums the partials over
//     workgroups in a fixed order and scatters them into the reference's flat parameter order
//     (optionally applying Adam in the same kernel: rnvp_train_step).
//     No float atomics anywhere: results are bitwise reproducible.
//   The tile loops are hand-scheduled in phases (rnvp_mfma_layer.h explains why); for d == 16 the
//   input-gradient product, like GEMM2 in the forward, runs as 4x4x1 MFMA blocks.
//   Per launch the host picks the row tiles per wave from the batch size (pick_rows) and, while a batch gives
//   at most one workgroup per CU, the net-split wave mode (layer_bwd): 8 waves, each pair sharing its row tiles.
// This header holds the device code and the launch templates; one translation unit per tile geometry instantiates them
// (rnvp_mfma_train_nf2.hip, _nf4.hip, _nf8.hip: the three compile in parallel), rnvp_mfma_train.hip holds the host side.
#pragma once
#include <atomic>

#include "rnvp_mfma_layer.h"

// amdgpu_waves_per_eu(RNVP_WPE, RNVP_WPE): see rnvp_mfma.hip (VGPR-form MFMAs, no AGPR copies).
#ifndef RNVP_WPE
#define RNVP_WPE 2
#endif

namespace rnvp {
namespace mfma {

// (shared by the translation units of the training step)
struct TrainPlan {
    int glayer_floats;      // per layer: 2 net blocks + db2
    size_t lds_bytes;       // at RMAX (the largest)
    int RMAX;
    size_t scratch_per_wave;   // floats, at RMAX
};
// layout of the partials a launch wrote (the net-split launches of d <= 16 use the compact dW2 records)
struct PartialLayout { int glayer_floats, w2c; };

namespace {

// RNVP_ABLATE: developer-only timing experiments (results are WRONG when set); never defined in the
// product build.  bit0 no LDS transposes, bit1 no LDS accumulation, bit2 no barrier/global flush,
// bit3 cheap activation, bit4 forward only, bit5 no weight-gradient MFMAs.
constexpr int kAblate = RNVP_ABLATE;
// RNVP_STAMP: diagnostic build that accumulates s_memtime deltas per phase and printf()s them for
// workgroup 0 (read the SHARES, not the absolute time: the stamps serialise the wave).
#ifdef RNVP_STAMP
#define STAMP(var) do { __builtin_amdgcn_sched_barrier(0); var = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define STAMP_ADD(acc, t0) do { unsigned long long t1__; STAMP(t1__); acc += t1__ - t0; t0 = t1__; } while (0)
#else
#define STAMP(var) do { } while (0)
#define STAMP_ADD(acc, t0) do { } while (0)
#endif
#define BWD_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
struct Stamps { unsigned long long fwd, bsetup, bloop, bflush, btail, ld, p1, p2, p3, p4, p5, fb1, fsum; };

#ifndef RNVP_TRAIN_WAVES
#define RNVP_TRAIN_WAVES 4
#endif
// the compact 4x4x1 form of dW2 (Dims::w2c) also where a wave runs both nets one after the other
constexpr bool kW2cNoNs = true;
// RNVP_TRAIN_WIDE: k_mfma_train_wide for d in (16, 32] when a batch needs more than 256 four-wave workgroups
#ifndef RNVP_TRAIN_WIDE
#define RNVP_TRAIN_WIDE 1
#endif
constexpr bool kTrainWide = RNVP_TRAIN_WIDE != 0;
#ifndef RNVP_TRAIN_BXF
#define RNVP_TRAIN_BXF 1
#endif
constexpr bool kTrainBxF = RNVP_TRAIN_BXF != 0;
constexpr bool kTrainSplit = kTrainBxF;                 // the packed block carries the split fragments
constexpr int kWaves = RNVP_TRAIN_WAVES;    // waves per workgroup
#ifndef RNVP_MAX_GRID_TRAIN
#define RNVP_MAX_GRID_TRAIN 512
#endif
constexpr int kMaxGridTrain = RNVP_MAX_GRID_TRAIN;
// RNVP_NS_TFLUSH (net-split launches, FT >= 2): the flush of the waves' LDS gradient slots into the workgroup's partial WITHOUT
// workgroup barriers, done by the t-net waves for BOTH nets.  On every SIMD the t wave (older: it wins the issue arbitration)
// runs ahead of its s partner and used to idle at the flush barriers while the s waves -- the kernel's critical path -- paid
// the arrival skew of both barriers and the sum itself (profiles/r05_ns_prio_ab.txt: barrier1 41-48k + sum 15-21k + barrier2
// 1-15k of 661k cycles on the s waves; 131-138k of barrier wait on the t waves).  Now the slots are two buffers of FT / 2 hidden
// tiles used in turn: a wave that has written a window counts itself in (one LDS counter per net) and goes on; the t waves,
// after their own window, wait for the window's eight arrivals, add the slots of both nets in slot order (the order of
// arrival plays no part: bitwise as before) and write the partial, then count the window done; a wave checks that count
// before it overwrites a buffer, two windows later.  The s waves never wait for a flush.
#ifndef RNVP_NS_TFLUSH
#define RNVP_NS_TFLUSH 1
#endif
constexpr bool kNsTFlush = RNVP_NS_TFLUSH != 0;
// RNVP_WIDE_TFLUSH: the same for the eight-wave wide form (d in (16, 32], more than 256 workgroups' worth of rows: C3 at 65 536): there
// the four older waves wait 24 % of the kernel at the flush barriers for the four younger ones (stamps in profiles/r05_wide_tflush_ab.txt)
#ifndef RNVP_WIDE_TFLUSH
#define RNVP_WIDE_TFLUSH 1
#endif
constexpr bool kWideTFlush = RNVP_WIDE_TFLUSH != 0;
constexpr size_t kSyncBytes = 128;                       // LDS behind a workgroup's buffers: the flush counters (FlushSync) and the error flag of the bounded spins
// the step's error word: one int behind the kMaxGridTrain x kWaves loss partials of the workspace (spin_nap, k_train_finish)
__host__ __device__ __forceinline__ int *error_word(float *losspart) { return reinterpret_cast<int *>(losspart + (size_t)kMaxGridTrain * kWaves); }
struct FlushSync { int *arr; int *done; int win; Poison poison; };      // arr[2 g + p]: arrivals of net group g (t / s waves) at windows of parity p; done: window shares summed; win: windows this wave finished
// The (h, g_pre) transposition tiles of the backward: lane (q, r) writes row r, columns 4q .. 4q + 3 as one b128 and reads back
// row 4 ks + q, column r as b32.  With the row stride kTS = 20 the b128 writes are conflict free, but the lanes of q and q + 1 of
// one 32-lane read group meet on 4 of the 32 banks (banks r and 20 + r overlap for r >= 12): every transposed read is 2-way
// conflicted.  That is 8.4 M of the kernel's 9.2 M SQ_LDS_BANK_CONFLICT cycles per 65 536-row launch (profiles/r06_lds_conflicts.txt:
// product 9.175 M, without the tiles 0.786 M).  A conflict-free layout exists -- row stride 16, the four f4 blocks of row R rotated
// by (R >> 1) & 3 -- and was built, parity green: conflicts 0.786 M, LDS-active cycles 39.7 M -> 31.3 M, launch time UNCHANGED
// (0.3052-0.3077 against 0.3066-0.3070 ms, profiles/r06_tt_swizzle_ab.txt): the LDS pipe is not what the issue-bound SIMDs wait for.
// Not kept ("only keep a change that shortens the launch").
__device__ __forceinline__ void lds_drain() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

template <int NF, int CQ> struct Dims {
    static constexpr int KS1 = NF + CQ;
    static constexpr int K4 = (KS1 + 3) / 4;
    static constexpr int OTL = NF >= 4 ? NF / 4 : 1;
    static constexpr int NT2 = NF >= 4 ? 2 * OTL : 1;
    static constexpr int KP4 = (KS1 + 1 + 3) / 4;     // f4 groups of [inputs | 1 | pad] per lane
    static constexpr int NTI = KP4;                   // N tiles of the W1 gradient
    static constexpr int MTI = OTL;                   // M tiles of the input gradient
    static constexpr int KSP = 4 * KP4;               // input columns per lane group
    static constexpr int SIN = 16 * NTI + 4;          // row stride of the input transposition tile
    // LDS scratch per wave: g_out^T staging (NT2 tiles), [in|1]^T staging, and 2 tiles (h, g_pre) per row tile of a sub-pass
    template <int R> static constexpr int tb() { return NT2 * 16 * kTS + 16 * SIN + 2 * (R >= 4 ? 2 : R) * 16 * kTS; }
#ifdef RNVP_TRAIN_FT
    static constexpr int FT = RNVP_TRAIN_FT;
#else
    // hidden tiles accumulated in LDS between two flushes: sized so that TWO workgroups fit in a CU's 160 KB
    // (2 waves per SIMD once the batch gives every CU two workgroups).  Measured against the previous 8/8/4:
    // C2 +9 % from 131 072 rows up (neutral at 65 536), C3 +7 % at 65 536 / +18 % at 262 144 rows, C4 +8.5 %.
    // (d <= 16 with windows of ONE tile, FT = 2 -- which would free 24.5 KB of LDS --: +0.8 % at 65 536 rows, +8 % at 16 960,
    // profiles/r06_ft2_ab.txt)
    static constexpr int FT = NF == 2 ? 4 : (NF == 4 ? 2 : 1);
#endif
    static constexpr int SLOT = FT * (NTI + OTL) * 256 + NT2 * 16;      // floats of one wave's slot
    // W2C (d <= 16 in net-split mode: every wave owns ONE net): dW2 = h^T g_out runs as 16 independent 4x4x1 blocks per
    // instruction instead of a 16x16x4 tile whose other net's eight columns are structural zeros -- block (q, hb) of lane
    // 16q + 4hb + j contracts hidden units 4hb..4hb+3 with out columns 4cb + j over the rows 4ks + q, the four row classes
    // q are added by a permlane reduce-scatter once per hidden tile, and the tile's dW2 record shrinks from 256 to 128
    // floats (only this net's 16 x 8 entries).  g_out^T is read from a wave-private LDS image [row tile][row][GS]
    // (position 2j + cb holds column 4cb + j, so one ds_read_b64 feeds both column blocks) instead of registers.
    // NS == 0 (a wave runs net t, then net s): the same form with one g_out^T image per net (RNVP_W2C_NO_NS).
    template <int NS> static constexpr bool w2c() { return NF == 2 && (NS == 1 || (NS == 0 && kW2cNoNs)) && kUseX4; }
    template <int NS> static constexpr int gimg(int R) { return (NS == 0 ? 2 : 1) * R * 16 * GS; }      // floats of the image(s)
    static constexpr int GS = 10;
    template <int NS> static constexpr int tblk() { return w2c<NS>() ? NTI * 256 + 128 : (NTI + OTL) * 256; }
    template <int NS> static constexpr int slot() { return FT * tblk<NS>() + NT2 * 16; }
    template <int R, int NS> static constexpr int tbn() {
        return (w2c<NS>() ? gimg<NS>(R) : NT2 * 16 * kTS) + 16 * SIN + 2 * (R >= 4 ? 2 : R) * 16 * kTS;
    }
};

// ---- backward of one layer ---------------------------------------------------------------------------
// NS (net split): the workgroup has 8 waves; waves w and w + 4 hold the SAME row tiles and take one net each
// (role 0: t, role 1: s).  Each runs its own net's hidden tiles and accumulates its own net's weight
// gradients; the two exchange only the net outputs (forward) and the input-gradient partial sums (here)
// through xown / xother in LDS, once per layer.
// What the backward of a layer opens with (tile-split kernel only): the saved layer input / exp(s) records and the first
// hidden tile's fragments, requested by the PREVIOUS call (the layer above) before its rendezvous -- the same latency
// argument as TilePre (rnvp_mfma_layer.h).
template <int NF, int CQ, int R> struct BwdPre {
    static constexpr int K4 = Dims<NF, CQ>::K4, OTL = Dims<NF, CQ>::OTL, NGI = (NF == 2 && kUseX4) ? 2 : Dims<NF, CQ>::MTI;
    float sx[R][NF], se[R][NF];
    f4 a1[K4], b1, a2t[OTL], a1t[NGI];
};
template <int NF, int CQ, int R>
__device__ __forceinline__ void load_bwd_pre(const float *__restrict__ W, const Geo &g, int lane, const float *__restrict__ scr,
                                             int net, int ht_lo, BwdPre<NF, CQ, R> &p) {
    using P = BwdPre<NF, CQ, R>;
    constexpr int K4 = P::K4, OTL = P::OTL, NGI = P::NGI;
    constexpr bool X4 = (NF == 2) && kUseX4;
    const int q = lane >> 4, HT = g.HT;
#pragma unroll
    for (int rt = 0; rt < R; ++rt)
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            p.sx[rt][f] = scr[((rt * 2 * NF) + f) * 64 + lane];
            p.se[rt][f] = scr[((rt * 2 * NF) + NF + f) * 64 + lane];
        }
    const float *pA1 = W + g.oA1 + ((size_t)net * HT * K4 * 64 + lane) * 4;
    const float *pB1 = W + g.oB1 + ((size_t)net * HT * 4 + q) * 4;
    const float *pA2T = W + g.oA2T + ((size_t)net * HT * OTL * 64 + lane) * 4;
    const float *pA1T = W + (X4 ? g.oA1X : g.oA1T) + ((size_t)net * HT * NGI * 64 + lane) * 4;
#pragma unroll
    for (int k4 = 0; k4 < K4; ++k4) p.a1[k4] = *opaque(pA1 + ((size_t)ht_lo * K4 + k4) * 256);
    p.b1 = *opaque(pB1 + ht_lo * 16);
#pragma unroll
    for (int o = 0; o < OTL; ++o) p.a2t[o] = *opaque(pA2T + ((size_t)ht_lo * OTL + o) * 256);
#pragma unroll
    for (int m = 0; m < NGI; ++m) p.a1t[m] = *opaque(pA1T + ((size_t)ht_lo * NGI + m) * 256);
}

template <int NF, int CQ, int R, int PC, int NS, int ACT, int WV = kWaves>
__device__ __forceinline__ void layer_bwd(const float *__restrict__ W, const Geo &g, int lane, int wave,
                                          float (&xr)[R][2 * NF], const float (&cr)[R][CQ > 0 ? CQ : 1],
                                          float (&gy)[R][2 * NF], const float (&gld)[R],
                                          const float *__restrict__ scr, float *lds, float *tb,
                                          float *gp_layer, bool first, Stamps &stp, float *xown,
                                          const float *xother, int tile_lo, int tile_hi,
                                          BwdPre<NF, CQ, R> &pre_ref, bool use_pre, const float *__restrict__ Wprev = nullptr,
                                          const float *__restrict__ scr_prev = nullptr, FlushSync *fs = nullptr) {
    BwdPre<NF, CQ, R> *const pre = &pre_ref;       // (a reference + flag, not a nullable pointer: the record must stay in registers)
    // pre (tile split only): this layer's opening loads, made by the caller / the layer above; Wprev, scr_prev (nullable):
    // the layer below, whose opening loads are requested here before the input-gradient rendezvous
    // NS == 2 (tile split, k_mfma_train_ts): every wave of the workgroup holds the same row tiles; this wave owns net
    // wave >> 2 and the hidden tiles [tile_lo, tile_hi) of it, writes their weight gradients straight to gp_layer (no
    // other wave has them) and adds its input-gradient share to those of all kTsWaves waves (xother = the record base).
    constexpr bool TS = NS == 2;
    unsigned long long t0 = 0; (void)t0;
    STAMP(t0);
    using D = Dims<NF, CQ>;
    constexpr int KS1 = D::KS1, K4 = D::K4, OTL = D::OTL, NT2 = D::NT2, KP4 = D::KP4, NTI = D::NTI,
                  MTI = D::MTI, KSP = D::KSP, SIN = D::SIN;
    const int q = lane >> 4, r = lane & 15, tid = wave * 64 + lane;
    const int role = NS ? (wave >> 2) : 0;
    const int HT = g.HT;
    const int ht_lo = TS ? tile_lo : 0, ht_hi = TS ? tile_hi : HT;
    constexpr bool W2C = D::template w2c<NS>();
    constexpr int FT = D::FT, SLOT = D::template slot<NS>(), TBLK = D::template tblk<NS>(), GS = D::GS;
    // barrier-free flush by the waves that run ahead: the t waves of a net-split launch (both nets), or -- RNVP_WIDE_TFLUSH -- waves
    // 0..3 of the eight-wave wide form, the older wave of every SIMD (the net the pass is on; the younger four never wait)
    constexpr bool TFW = kWideTFlush && NS == 0 && WV == 8 && FT >= 2 && FT % 2 == 0;
    constexpr bool TF = (kNsTFlush && NS == 1 && FT >= 2 && FT % 2 == 0 && WV == 4) || TFW;
    constexpr int FT2 = FT / 2 > 0 ? FT / 2 : 1;
    const int netblock = HT * TBLK;                       // floats of one net's gradient block
    float *slot = lds + wave * SLOT;
    float *bufG = tb;                                     // NT2 tiles of 16 x kTS (g_out^T staging); W2C: R x 16 x GS
    float *bufI = tb + (W2C ? D::template gimg<NS>(R) : NT2 * 16 * kTS);   // 16 x SIN
    float *bufH = bufI + 16 * SIN;                        // 2R tiles of 16 x kTS: (h, g_pre) per row tile
    // 1. restore the layer input, form g_out = [g_t | g_s] and the gradient of the pass-through part
    f4 go[R][NT2];
#pragma unroll
    for (int rt = 0; rt < R; ++rt) {
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const float sx = (TS && use_pre) ? pre->sx[rt][f] : scr[((rt * 2 * NF) + f) * 64 + lane];
            const float se = (TS && use_pre) ? pre->se[rt][f] : scr[((rt * 2 * NF) + NF + f) * 64 + lane];
            const int e = 2 * f + 1 - PC;
            const float gyv = gy[rt][e];
            const float gt = gyv;                                  // (1-m) * gy
            const float gs = fmaf(gyv * sx, se, gld[rt]);          // (1-m) * (gy * x * e^s + gld)
            gy[rt][e] = gyv * se;                                  // gy * (1-m) * e^s
            xr[rt][e] = sx;
            if (NF >= 4) { go[rt][f >> 2][f & 3] = gt; go[rt][(NF >= 4 ? OTL : 0) + (f >> 2)][f & 3] = gs; }
            else { go[rt][0][f & 1] = gt; go[rt][0][2 + (f & 1)] = gs; }
        }
    }
    f4 gb2[NT2];
#pragma unroll
    for (int ot = 0; ot < NT2; ++ot) {
        gb2[ot] = go[0][ot];
#pragma unroll
        for (int rt = 1; rt < R; ++rt) gb2[ot] += go[rt][ot];
    }
    // 2. row-contraction operands: g_out^T and [in | 1]^T through the wave's LDS tiles
    float goT[R][NT2][4], inT[R][NTI][4];
#pragma unroll
    for (int rt = 0; rt < R; ++rt) {
        if constexpr (W2C && NS == 0) {      // one image per net
            float *gp = bufG + (rt * 16 + r) * GS + 4 * (q & 1) + (q >> 1);
            wave_lds_fence();
            gp[0] = go[rt][0][0]; gp[2] = go[rt][0][1];
            gp[R * 16 * GS] = go[rt][0][2]; gp[R * 16 * GS + 2] = go[rt][0][3];
        } else if constexpr (W2C) {        // this wave's net only: columns 2q, 2q+1 of row r, at positions 2j + cb (Dims)
            const float v0 = role ? go[rt][0][2] : go[rt][0][0], v1 = role ? go[rt][0][3] : go[rt][0][1];
            float *gp = bufG + (rt * 16 + r) * GS + 4 * (q & 1) + (q >> 1);
            wave_lds_fence();
            gp[0] = v0; gp[2] = v1;
        } else {
#pragma unroll
            for (int ot = 0; ot < NT2; ++ot) transpose16(bufG + ot * 16 * kTS, go[rt][ot], lane, goT[rt][ot]);
        }
        wave_lds_fence();
#pragma unroll
        for (int k4 = 0; k4 < KP4; ++k4) {
            f4 v;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int kk = 4 * k4 + u;
                v[u] = (kk < KS1) ? in_op<NF, CQ, PC, R>(xr, cr, rt, kk < KS1 ? kk : 0)
                                  : (kk == KS1 ? (q == 0 ? 1.0f : 0.0f) : 0.0f);
            }
            *reinterpret_cast<f4 *>(bufI + r * SIN + q * KSP + 4 * k4) = v;
        }
        wave_lds_fence();
#pragma unroll
        for (int nt = 0; nt < NTI; ++nt)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) inT[rt][nt][ks] = bufI[(4 * ks + q) * SIN + 16 * nt + r];
    }
    // d == 16: the input gradient runs as 4x4x1 blocks (see run_tiles_x4): 2 x 4 partial outputs per lane
    constexpr bool X4 = (NF == 2) && kUseX4;
    constexpr int NGI = X4 ? 2 : MTI;
    f4 gin[R][NGI];
#pragma unroll
    for (int rt = 0; rt < R; ++rt)
#pragma unroll
        for (int mt = 0; mt < NGI; ++mt) gin[rt][mt] = f4{0.f, 0.f, 0.f, 0.f};
    STAMP_ADD(stp.bsetup, t0);

    // 3. the two nets, hidden tile by hidden tile.  Hand-scheduled like the forward layer
    //    (rnvp_mfma_layer.h): every phase runs over all R row tiles so that dependent MFMA chains
    //    interleave, LDS round trips are covered by the input-gradient MFMAs, and the weight
    //    fragments of the next tile are in flight for a whole iteration.
    auto net_pass = [&](auto net_c) {
        constexpr int net = decltype(net_c)::value;
        const float *pA1 = W + g.oA1 + ((size_t)net * HT * K4 * 64 + lane) * 4;
        const float *pB1 = W + g.oB1 + ((size_t)net * HT * 4 + q) * 4;
        const float *pA2T = W + g.oA2T + ((size_t)net * HT * OTL * 64 + lane) * 4;
        const float *pA1T = W + (X4 ? g.oA1X : g.oA1T) + ((size_t)net * HT * NGI * 64 + lane) * 4;
        f4 a1[K4], a2t[OTL], a1t[NGI], b1;
        bool have = false;
        if constexpr (TS) {
            if (use_pre) {          // loaded ahead (BwdPre)
                have = true;
#pragma unroll
                for (int k4 = 0; k4 < K4; ++k4) a1[k4] = pre->a1[k4];
                b1 = pre->b1;
#pragma unroll
                for (int o = 0; o < OTL; ++o) a2t[o] = pre->a2t[o];
#pragma unroll
                for (int m = 0; m < NGI; ++m) a1t[m] = pre->a1t[m];
            }
        }
        if (!have) {
#pragma unroll
        for (int k4 = 0; k4 < K4; ++k4) a1[k4] = *reinterpret_cast<const f4 *>(pA1 + ((size_t)ht_lo * K4 + k4) * 256);
        b1 = *reinterpret_cast<const f4 *>(pB1 + ht_lo * 16);
#pragma unroll
        for (int o = 0; o < OTL; ++o) a2t[o] = *reinterpret_cast<const f4 *>(pA2T + ((size_t)ht_lo * OTL + o) * 256);
#pragma unroll
        for (int m = 0; m < NGI; ++m) a1t[m] = *reinterpret_cast<const f4 *>(pA1T + ((size_t)ht_lo * NGI + m) * 256);
        }
        for (int ht = ht_lo; ht < ht_hi; ++ht) {
            const int nx = (kAblate & 64) ? 0 : ((ht + 1 < ht_hi) ? ht + 1 : ht);
            // The next tile's weight fragments.  d <= 16: a second register set, requested here, a whole iteration ahead, and
            // copied over at the loop's end.  Wider rows (INPL): loaded IN PLACE right after their last use in this tile (a1 / b1 /
            // a2t after phase 1 of the last row-tile half, a1t after its phase 3) -- half an iteration of cover is enough for the
            // L2-resident fragments, and the 20-36 registers and 10-18 copies per tile buy more than the distance: round 6, same
            // box, 65 536 rows: C3 1.188 -> 1.170 ms, C4 1.067 -> 1.032; C2 0.2674 -> 0.2681 (hence not there)
            // (profiles/r06_bwd_inplace_ab.txt)
            constexpr bool INPL = NF >= 4;
            f4 na1[K4], na2t[OTL], na1t[NGI], nb1;
            if constexpr (!INPL) {
#pragma unroll
                for (int k4 = 0; k4 < K4; ++k4) na1[k4] = *opaque(pA1 + ((size_t)nx * K4 + k4) * 256);
                nb1 = *opaque(pB1 + nx * 16);
#pragma unroll
                for (int o = 0; o < OTL; ++o) na2t[o] = *opaque(pA2T + ((size_t)nx * OTL + o) * 256);
#pragma unroll
                for (int m = 0; m < NGI; ++m) na1t[m] = *opaque(pA1T + ((size_t)nx * NGI + m) * 256);
            }
            BWD_SCHED_BARRIER();

            f4 gW2[W2C ? 2 : OTL], gW1[NTI];           // W2C: one accumulator per column block cb
#pragma unroll
            for (int o = 0; o < (W2C ? 2 : OTL); ++o) gW2[o] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int nt = 0; nt < NTI; ++nt) gW1[nt] = f4{0.f, 0.f, 0.f, 0.f};
            // The four phases run over RH row tiles at a time (two interleaved MFMA chains are enough for
            // the 16x16x4 issue rate; holding the transients of all R tiles at once spills at 256 VGPRs).
            constexpr int RH = (R >= 4) ? 2 : R;
#pragma unroll
            for (int r0 = 0; r0 < R; r0 += RH) {
                // phase 1 (MFMA): GEMM1 recompute and g_h = W2^T g_out, chains interleaved over row tiles
                f4 acc[RH], gh[RH];
#pragma unroll
                for (int u = 0; u < RH; ++u) { acc[u] = b1; gh[u] = f4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
                for (int kk = 0; kk < KS1; ++kk)
#pragma unroll
                    for (int u = 0; u < RH; ++u)
                        acc[u] = mfma16(a1[kk >> 2][kk & 3], in_op<NF, CQ, PC, R>(xr, cr, r0 + u, kk), acc[u]);
                if (NF >= 4) {
#pragma unroll
                    for (int o = 0; o < OTL; ++o)
#pragma unroll
                        for (int rho = 0; rho < 4; ++rho)
#pragma unroll
                            for (int u = 0; u < RH; ++u)
                                gh[u] = mfma16(a2t[o][rho], go[r0 + u][(NF >= 4 ? net * OTL : 0) + o][rho], gh[u]);
                } else {
#pragma unroll
                    for (int v = 0; v < 2; ++v)
#pragma unroll
                        for (int u = 0; u < RH; ++u)
                            gh[u] = mfma16(a2t[0][2 * net + v], go[r0 + u][0][2 * net + v], gh[u]);
                }
                if constexpr (INPL) {
                    if (r0 + RH >= R) {
#pragma unroll
                        for (int k4 = 0; k4 < K4; ++k4) a1[k4] = *opaque(pA1 + ((size_t)nx * K4 + k4) * 256);
                        b1 = *opaque(pB1 + nx * 16);
#pragma unroll
                        for (int o = 0; o < OTL; ++o) a2t[o] = *opaque(pA2T + ((size_t)nx * OTL + o) * 256);
                    }
                }
                BWD_SCHED_BARRIER();
                STAMP_ADD(stp.p1, t0);

                // phase 2 (VALU + LDS writes): h = tanh, g_pre = g_h * (1 - h^2); both go to this wave's
                // per-row-tile transposition tiles
                f4 gpv[RH];
                wave_lds_fence();
#pragma unroll
                for (int u = 0; u < RH; ++u) {
                    const f4 hv = act4<ACT>(acc[u]);
                    gpv[u] = gh[u] * dact4<ACT>(hv);                                     // activation'
                    if (!(kAblate & 1)) {
                        *reinterpret_cast<f4 *>(bufH + (2 * u) * 16 * kTS + r * kTS + 4 * q) = hv;
                        *reinterpret_cast<f4 *>(bufH + (2 * u + 1) * 16 * kTS + r * kTS + 4 * q) = gpv[u];
                    } else {
                        asm volatile("" ::"v"(hv));
                    }
                }
                wave_lds_fence();
                BWD_SCHED_BARRIER();
                STAMP_ADD(stp.p2, t0);

                // phase 3 (MFMA): g_in += W1^T g_pre  -- covers the LDS round trip of the transposed reads that follow it in
                // program order (requesting them AHEAD of the products measured +2 %: profiles/r05_reads_first_ab.txt)
                float hT[RH][4], pT[RH][4];
                float2 gB[W2C ? RH : 1][4];
                auto read_transposed = [&]() {
#pragma unroll
                    for (int u = 0; u < RH; ++u)
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) {
                            if (!(kAblate & 1)) {
                                hT[u][ks] = bufH[(2 * u) * 16 * kTS + (4 * ks + q) * kTS + r];
                                pT[u][ks] = bufH[(2 * u + 1) * 16 * kTS + (4 * ks + q) * kTS + r];
                            } else {
                                hT[u][ks] = gpv[u][ks]; pT[u][ks] = gpv[u][ks];
                            }
                            if constexpr (W2C)
                                gB[u][ks] = *reinterpret_cast<const float2 *>(bufG + (NS == 0 ? net * R * 16 * GS : 0) +
                                                                              ((r0 + u) * 16 + 4 * ks + q) * GS + 2 * (r & 3));
                        }
                };
                if constexpr (X4) {
#pragma unroll
                    for (int rho = 0; rho < 4; ++rho)
#pragma unroll
                        for (int m = 0; m < 2; ++m)
#pragma unroll
                            for (int u = 0; u < RH; ++u) gin[r0 + u][m] = mfma4(a1t[m][rho], gpv[u][rho], gin[r0 + u][m]);
                } else {
#pragma unroll
                    for (int m = 0; m < MTI; ++m)
#pragma unroll
                        for (int rho = 0; rho < 4; ++rho)
#pragma unroll
                            for (int u = 0; u < RH; ++u)
                                gin[r0 + u][m] = mfma16(a1t[m][rho], gpv[u][rho], gin[r0 + u][m]);
                }
                if constexpr (INPL) {
                    if (r0 + RH >= R) {
#pragma unroll
                        for (int m = 0; m < NGI; ++m) a1t[m] = *opaque(pA1T + ((size_t)nx * NGI + m) * 256);
                    }
                }
                read_transposed();
                BWD_SCHED_BARRIER();
                STAMP_ADD(stp.p3, t0);

                // phase 4 (MFMA): dW2 += h g_out^T, dW1|db1 += g_pre [in|1]^T; independent chains alternate
                if (!(kAblate & 32)) {
#pragma unroll
                    for (int u = 0; u < RH; ++u)
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) {
                            if constexpr (W2C) {
                                gW2[0] = mfma4(hT[u][ks], gB[u][ks].x, gW2[0]);
                                gW2[1] = mfma4(hT[u][ks], gB[u][ks].y, gW2[1]);
                            } else {
#pragma unroll
                                for (int o = 0; o < OTL; ++o)
                                    gW2[o] = mfma16(hT[u][ks], goT[r0 + u][(NF >= 4 ? net * OTL : 0) + o][ks], gW2[o]);
                            }
#pragma unroll
                            for (int nt = 0; nt < NTI; ++nt) gW1[nt] = mfma16(pT[u][ks], inT[r0 + u][nt][ks], gW1[nt]);
                        }
                } else {
#pragma unroll
                    for (int u = 0; u < RH; ++u)
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) { gW2[0][ks] += hT[u][ks]; gW1[0][ks] += pT[u][ks]; }
                }
                BWD_SCHED_BARRIER();
                STAMP_ADD(stp.p4, t0);
            }
            // this wave's share of dW1|db1 and dW2 for hidden tile ht -> its own LDS slot
            if constexpr (TS) {      // the only share there is: straight to the gradient record
                float *gd = gp_layer + (size_t)net * netblock + (size_t)ht * TBLK + lane * 4;
#pragma unroll
                for (int nt = 0; nt < NTI; ++nt) *reinterpret_cast<f4 *>(gd + nt * 256) = gW1[nt];
#pragma unroll
                for (int o = 0; o < OTL; ++o) *reinterpret_cast<f4 *>(gd + (NTI + o) * 256) = gW2[o];
            } else if (!(kAblate & 2)) {
                int spos = ht % FT;
                if constexpr (TF) {
                    // window = FT / 2 tiles in buffer (window number & 1); its first tile: the sum of the window two back must
                    // have read this buffer (four t-wave shares per window)
                    if (ht % FT2 == 0 && fs->win >= 2) {
                        int spins = 0;
                        while (__hip_atomic_load(fs->done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < 4 * (fs->win - 1)) spin_nap(spins, fs->poison);
                        asm volatile("" ::: "memory");
                    }
                    spos = (fs->win & 1) * FT2 + ht % FT2;
                }
                float *sb = slot + (size_t)spos * TBLK + lane * 4;
#pragma unroll
                for (int nt = 0; nt < NTI; ++nt) *reinterpret_cast<f4 *>(sb + nt * 256) = gW1[nt];
                if constexpr (W2C) {
                    // add the four row classes q; lane group q keeps column block q >> 1, hidden units 4hb + 2(q & 1) + {0, 1}
                    float s4[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) s4[i] = swap_add32(gW2[0][i], gW2[1][i]);
                    float2 kept;
                    kept.x = swap_add16(s4[0], s4[2]); kept.y = swap_add16(s4[1], s4[3]);
                    *reinterpret_cast<float2 *>(slot + (size_t)spos * TBLK + NTI * 256 + lane * 2) = kept;
                } else {
#pragma unroll
                    for (int o = 0; o < OTL; ++o) *reinterpret_cast<f4 *>(sb + (NTI + o) * 256) = gW2[o];
                }
            } else {
#pragma unroll
                for (int nt = 0; nt < NTI; ++nt) asm volatile("" ::"v"(gW1[nt]));
#pragma unroll
                for (int o = 0; o < (W2C ? 2 : OTL); ++o) asm volatile("" ::"v"(gW2[o]));
            }
            const bool last_tile = (ht + 1 == HT);
            STAMP_ADD(stp.p5, t0);
            if (!TS && ((ht + 1) % (TF ? FT2 : FT) == 0 || last_tile) && !(kAblate & 4)) {
                if (last_tile && net == 1) {
                    // db2: sum g_out over the 16 rows of the tile(s); lanes r == 0 hold (q, reg) sums
#pragma unroll
                    for (int ot = 0; ot < NT2; ++ot)
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            float v = gb2[ot][u];
                            v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
                            if (r == 0) slot[FT * TBLK + (ot * 4 + q) * 4 + u] = v;
                        }
                }
                // A workgroup's second and later row groups ADD to its partial: the old values are requested here, ahead of the
                // barrier, so that their memory latency (an L2 miss: the partials of all workgroups exceed the L2) overlaps the wait
                // for the slowest wave instead of following it -- C4 spent 17 % of the kernel in this read-modify-write
                // (profiles/r04_stamp_c4.txt)
                if constexpr (TF) {
                    const int w_t0 = (ht / FT2) * FT2, w_n4 = (ht + 1 - w_t0) * TBLK / 4;      // the window's first tile, its f4 count per net
                    const int bufoff = (fs->win & 1) * FT2 * TBLK;
                    lds_drain();                                               // this wave's slot (and db2) writes have landed
                    // arrivals are counted per net group AND window parity: a wave may run a whole window ahead of a slower one, so a
                    // single running count could reach "everybody arrived" on the fast waves' NEXT window; windows of one parity are
                    // strictly ordered by the completion check above, which makes the per-parity count exact
                    const int wpar = fs->win & 1, wneed = (fs->win >> 1) + 1;
#ifdef RNVP_SPIN_TEST      // developer build: wave 5 of workgroup 0 "forgets" one arrival -> the flushing waves' bounded wait must end the kernel
                    if (!(blockIdx.x == 0 && wave == 5 && fs->win == 3))
#endif
                    if (lane == 0) __hip_atomic_fetch_add(fs->arr + 2 * role + wpar, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    ++fs->win;
                    STAMP_ADD(stp.fb1, t0);
                    if (TFW ? wave < 4 : role == 0) {
                        // a workgroup's second and later row groups ADD to its partial: the old values are requested ahead of the wait
                        f4 oldv[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
                        if (!first && tid < w_n4) {
#pragma unroll
                            for (int n2 = 0; n2 < (TFW ? 1 : 2); ++n2)
                                oldv[n2] = *(reinterpret_cast<const f4 *>(gp_layer + (size_t)(TFW ? net : n2) * netblock + (size_t)w_t0 * TBLK) + tid);
                        }
                        // the flushing waves wait for the window's arrivals (the others are behind: this is the wait the flush barrier
                        // used to be, minus the late waves' share of it)
                        int spins = 0;
                        if constexpr (TFW) {
                            while (__hip_atomic_load(fs->arr + wpar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < WV * wneed) spin_nap(spins, fs->poison);
                        } else {
                            while (__hip_atomic_load(fs->arr + 2 + wpar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < 4 * wneed ||
                                   __hip_atomic_load(fs->arr + wpar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < 4 * wneed) spin_nap(spins, fs->poison);
                        }
                        asm volatile("" ::: "memory");
                        STAMP_ADD(stp.bflush, t0);
                        static_assert(FT2 * TBLK / 4 <= 256, "one f4 per flushing thread and net");
                        if (tid < w_n4) {
#pragma unroll
                            for (int n2 = 0; n2 < (TFW ? 1 : 2); ++n2) {
                                f4 *dst = reinterpret_cast<f4 *>(gp_layer + (size_t)(TFW ? net : n2) * netblock + (size_t)w_t0 * TBLK) + tid;
                                const f4 *s0 = reinterpret_cast<const f4 *>(lds + (size_t)(TFW ? 0 : n2 * WV) * SLOT + bufoff) + tid;
                                f4 v = s0[0];
#pragma unroll
                                for (int w = 1; w < WV; ++w) v += s0[w * (SLOT / 4)];          // slot order: deterministic
                                *dst = first ? v : v + oldv[n2];
                            }
                        }
                        if (last_tile && (!TFW || net == 1) && tid < NT2 * 16) {       // db2: in the slots of the waves that ran net s
                            const int i = (TFW ? 0 : WV * SLOT) + FT * TBLK + tid;
                            float v = lds[i];
#pragma unroll
                            for (int w = 1; w < WV; ++w) v += lds[w * SLOT + i];
                            float *p = gp_layer + 2 * (size_t)netblock + tid;
                            *p = first ? v : *p + v;
                        }
                        lds_drain();                                           // the slot reads are done: the buffer may be rewritten
                        if (lane == 0) __hip_atomic_fetch_add(fs->done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    STAMP_ADD(stp.fsum, t0);
                } else {
                constexpr int FTH = NS ? 256 : WV * 64;                  // threads that add one net's slots
                constexpr int KIT = (FT * TBLK / 4 + FTH - 1) / FTH;
                const int fl_t0 = (ht / FT) * FT, fl_n4 = (ht + 1 - fl_t0) * TBLK / 4;
                const int fl_i0 = NS ? (tid & 255) : tid;
                f4 *fl_dst = reinterpret_cast<f4 *>(gp_layer + (size_t)(NS ? (tid >> 8) : net) * netblock + (size_t)fl_t0 * TBLK);
                f4 oldv[KIT];
#pragma unroll
                for (int u = 0; u < KIT; ++u) {
                    oldv[u] = f4{0.f, 0.f, 0.f, 0.f};
                    if (!first && fl_i0 + u * FTH < fl_n4) oldv[u] = fl_dst[fl_i0 + u * FTH];
                }
                __syncthreads();
                STAMP_ADD(stp.fb1, t0);
                if constexpr (NS) {
                    // both nets flush together: threads 0..255 add the slots of waves 0..3 (t net), threads
                    // 256..511 those of waves 4..7 (s net), each in wave order
                    const int fr = tid >> 8;
                    const f4 *s0 = reinterpret_cast<const f4 *>(lds + (size_t)fr * WV * SLOT);
#pragma unroll
                    for (int u = 0; u < KIT; ++u) {
                        const int i = fl_i0 + u * FTH;
                        if (i < fl_n4) {
                            f4 v = s0[i];
#pragma unroll
                            for (int w = 1; w < WV; ++w) v += s0[w * (SLOT / 4) + i];
                            fl_dst[i] = first ? v : v + oldv[u];
                        }
                    }
                    if (last_tile && tid < NT2 * 16) {
                        const int i = WV * SLOT + FT * TBLK + tid;                  // db2 lives in the s waves' slots
                        float v = lds[i];
#pragma unroll
                        for (int w = 1; w < WV; ++w) v += lds[w * SLOT + i];
                        float *p = gp_layer + 2 * (size_t)netblock + tid;
                        *p = first ? v : *p + v;
                    }
                } else
                {   // slot0 + slot1 + slot2 + slot3 (wave order) -> the workgroup's partial in global memory
                    const f4 *s0 = reinterpret_cast<const f4 *>(lds);
#pragma unroll
                    for (int u = 0; u < KIT; ++u) {
                        const int i = fl_i0 + u * FTH;
                        if (i < fl_n4) {
                            f4 v = s0[i];
#pragma unroll
                            for (int w = 1; w < WV; ++w) v += s0[w * (SLOT / 4) + i];      // wave order: deterministic
                            fl_dst[i] = first ? v : v + oldv[u];
                        }
                    }
                    if (last_tile && net == 1 && tid < NT2 * 16) {
                        const int i = FT * TBLK + tid;
                        float v = lds[i];
#pragma unroll
                        for (int w = 1; w < WV; ++w) v += lds[w * SLOT + i];
                        float *p = gp_layer + 2 * (size_t)netblock + tid;
                        *p = first ? v : *p + v;
                    }
                }
                STAMP_ADD(stp.fsum, t0);
                if (!(kAblate & 128)) __syncthreads();
                STAMP_ADD(stp.bflush, t0);
                }
            }
            if constexpr (!INPL) {
#pragma unroll
                for (int k4 = 0; k4 < K4; ++k4) a1[k4] = na1[k4];
                b1 = nb1;
#pragma unroll
                for (int o = 0; o < OTL; ++o) a2t[o] = na2t[o];
#pragma unroll
                for (int m = 0; m < NGI; ++m) a1t[m] = na1t[m];
            }
        }
    };
    if constexpr (TS) {
        if (ht_lo < ht_hi) {
            if (role == 0) net_pass(std::integral_constant<int, 0>{});
            else net_pass(std::integral_constant<int, 1>{});
        }
        // the layer below: request what its backward opens with, now, ahead of the rendezvous at the end of this one
        if (use_pre && Wprev) load_bwd_pre<NF, CQ, R>(Wprev, g, lane, scr_prev, role, ht_lo, *pre);
        if (wave == 0) {         // db2: sum g_out over the 16 rows of the tile(s); lanes r == 0 hold (q, reg) sums
#pragma unroll
            for (int ot = 0; ot < NT2; ++ot)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float v = gb2[ot][u];
                    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
                    if (r == 0) gp_layer[2 * (size_t)netblock + (ot * 4 + q) * 4 + u] = v;
                }
        }
    } else if constexpr (NS) {
        if (role == 0) net_pass(std::integral_constant<int, 0>{});
        else net_pass(std::integral_constant<int, 1>{});
    } else {
        net_pass(std::integral_constant<int, 0>{});
        net_pass(std::integral_constant<int, 1>{});
    }
    // 4. gradient reaching the conditioning features through the nets
    float gi[R][NF];
#pragma unroll
    for (int rt = 0; rt < R; ++rt) {
        if constexpr (X4) {      // partial sums over the lane groups -> the owner of each conditioning feature
            float s4[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) s4[i] = swap_add32(gin[rt][0][i], gin[rt][1][i]);
            gi[rt][0] = swap_add16(s4[0], s4[2]);
            gi[rt][1] = swap_add16(s4[1], s4[3]);
        } else {
#pragma unroll
            for (int f = 0; f < NF; ++f) gi[rt][f] = (NF >= 4) ? gin[rt][f >> 2][f & 3] : gin[rt][0][f & 1];
        }
    }
    if constexpr (TS) {          // every wave's share of the input gradient, added in wave order
#pragma unroll
        for (int rt = 0; rt < R; ++rt)
#pragma unroll
            for (int f = 0; f < NF; ++f) xown[(rt * NF + f) * 64 + lane] = gi[rt][f];
        __syncthreads();
#pragma unroll
        for (int rt = 0; rt < R; ++rt)
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                float a = 0.f;
#pragma unroll
                for (int w = 0; w < kTsWaves; ++w) a += xother[(w * R * NF + rt * NF + f) * 64 + lane];
                gi[rt][f] = a;
            }
    } else if constexpr (NS) {   // this wave summed its own net only: add the partner's share (t + s, both waves alike)
#pragma unroll
        for (int rt = 0; rt < R; ++rt)
#pragma unroll
            for (int f = 0; f < NF; ++f) xown[(rt * NF + f) * 64 + lane] = gi[rt][f];
        __syncthreads();
#pragma unroll
        for (int rt = 0; rt < R; ++rt)
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const float o = xother[(rt * NF + f) * 64 + lane];
                gi[rt][f] = role == 0 ? gi[rt][f] + o : o + gi[rt][f];
            }
    }
#pragma unroll
    for (int rt = 0; rt < R; ++rt) {
#pragma unroll
        for (int f = 0; f < NF; ++f) gy[rt][2 * f + PC] += gi[rt][f];
    }
    STAMP_ADD(stp.btail, t0);
}

// WV: waves of the workgroup that own row tiles (4; 8 in the wide form k_mfma_train_wide, NS == 0 only)
template <int NF, int CQ, int R, int NS, int ACT, bool BXF, int WV = kWaves>
__device__ __forceinline__ void train_body(const float *__restrict__ wp, const Geo &g, int L, int alt, const float *__restrict__ x,
             const float *__restrict__ c, const int64_t *__restrict__ row_index, int64_t n, float inv_B,
             float *gpart, float *losspart, float *scratch, int glayer_floats, Seeds sd) {
    // sd.gz != nullptr (a prior other than N(0, I), rnvp_loss_grad_zseed): the backward is seeded with the caller's
    // d loss / d z rows and the loss partial carries the log-det term only
    using DM = Dims<NF, CQ>;
    constexpr int D = 8 * NF, CD = 4 * CQ;
    constexpr int NW = WV * (1 + NS);                 // waves in the workgroup
    constexpr int XW = R * NF * 64;                       // floats one wave exchanges per layer (NS)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pw = wave & (WV - 1), role = NS ? wave >> 2 : 0;     // row owner index; net of this wave (NS)
    const int q = lane >> 4, r = lane & 15;
    constexpr int SLOTN = DM::template slot<NS>(), TBN = DM::template tbn<R, NS>();
    float *tb = lds + NW * SLOTN + wave * TBN;
    float *xbuf = lds + NW * SLOTN + NW * TBN;                         // NS: 2 x NW x XW, double buffered by layer parity
    // the flush counters (FlushSync) and the error flag of the bounded spins (rnvp_mfma_layer.h spin_nap), behind the exchange records
    int *sync = reinterpret_cast<int *>(xbuf + (NS ? 2 * NW * XW : 0));
    const Poison poison{losspart + (size_t)blockIdx.x * kWaves, sync + 13, error_word(losspart)};
    FlushSync fsync{sync + 8, sync + 12, 0, poison};
    if constexpr (NS == 1 || (kWideTFlush && NS == 0 && WV == 8)) {
        if (threadIdx.x < 6) sync[8 + threadIdx.x] = 0;
        __syncthreads();
    }
    const int64_t rows_per_wg = (int64_t)WV * R * 16;
    const int64_t ngroups = (n + rows_per_wg - 1) / rows_per_wg;
    const float prior_c = 0.5f * (float)g.d * kLog2Pi;
    const bool full = (g.d == D) && (g.c == CD) && ((uintptr_t)x & 15) == 0;     // else: guarded scalar row loads
    float *gp = gpart + (size_t)blockIdx.x * glayer_floats * L;
    float *scr_wave = scratch + ((size_t)blockIdx.x * WV + pw) * L * R * 2 * NF * 64;
    float wave_sum = 0.f;
    bool first = true;
    Stamps stp = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t0 = 0, tk0 = 0; (void)t0; (void)tk0;
    STAMP(tk0);
    for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        STAMP(t0);
        const int64_t base = grp * rows_per_wg + (int64_t)pw * R * 16;
        float xr[R][2 * NF], cr[R][CQ > 0 ? CQ : 1], ld[R], gy[R][2 * NF], gld[R];
        bool valid[R];
#pragma unroll
        for (int rt = 0; rt < R; ++rt) {
            const int64_t row = base + rt * 16 + r;
            valid[rt] = row < n;
            const int64_t src = valid[rt] ? (row_index ? row_index[row] : row) : 0;
            load_row<NF, CQ>(x, c, src, g.d, g.c, full, q, xr[rt], cr[rt]);
            ld[rt] = 0.f;
        }
        STAMP_ADD(stp.ld, t0);
        for (int l = 0; l < L; ++l) {
            const float *W = wp + (size_t)l * g.layer_floats;
            float *scr = scr_wave + (size_t)l * R * 2 * NF * 64;
            if constexpr (NS) {
                float *xb = xbuf + (size_t)(l & 1) * NW * XW;
                if ((l + alt) & 1) layer_forward_ns<NF, CQ, R, 1, 2, ACT, BXF>(W, g, lane, role, xb + wave * XW, xb + (wave ^ WV) * XW, xr, cr, ld, scr);
                else layer_forward_ns<NF, CQ, R, 0, 2, ACT, BXF>(W, g, lane, role, xb + wave * XW, xb + (wave ^ WV) * XW, xr, cr, ld, scr);
            } else {
                if ((l + alt) & 1) layer_forward<NF, CQ, R, 1, 2, ACT, BXF>(W, g, lane, xr, cr, ld, scr);
                else layer_forward<NF, CQ, R, 0, 2, ACT, BXF>(W, g, lane, xr, cr, ld, scr);
            }
        }
        if constexpr (NS) __syncthreads();      // the pair's scratch records (written half by each wave) and the exchange buffers
#pragma unroll
        for (int rt = 0; rt < R; ++rt) {
            float ss = 0.f;
#pragma unroll
            for (int v = 0; v < 2 * NF; ++v) ss = fmaf(xr[rt][v], xr[rt][v], ss);
            float l1 = ld[rt];
            l1 += __shfl_xor(l1, 16); l1 += __shfl_xor(l1, 32);
            ss += __shfl_xor(ss, 16); ss += __shfl_xor(ss, 32);
            const float lp = sd.gz ? l1 : l1 + (-0.5f * ss - prior_c);
            float v = (valid[rt] && q == 0) ? lp : 0.f;
            v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
            wave_sum += v;
            // seed of the backward: d(-mean logp)/dz = z / B; padding rows contribute nothing
            const float sc = valid[rt] ? inv_B : 0.f;
            if (sd.gz) {
                const int64_t row = base + rt * 16 + r;
#pragma unroll
                for (int u = 0; u < 2 * NF; ++u) {
                    const int j = q * 2 * NF + u;
                    gy[rt][u] = (valid[rt] && j < g.d) ? sd.gz[row * g.d + j] : 0.f;
                }
            } else {
#pragma unroll
                for (int u = 0; u < 2 * NF; ++u) gy[rt][u] = xr[rt][u] * sc;
            }
            gld[rt] = -sc;                      // (per-row seeds, rnvp_backward: the tile-split kernel below; see backward_rows_ok)
        }
        STAMP_ADD(stp.fwd, t0);
        for (int l = L - 1; l >= 0 && !(kAblate & 16); --l) {
            const float *W = wp + (size_t)l * g.layer_floats;
            const float *scr = scr_wave + (size_t)l * R * 2 * NF * 64;
            float *gpl = gp + (size_t)l * glayer_floats;
            float *xb = xbuf + (size_t)(l & 1) * NW * XW;
            float *xo = NS ? xb + wave * XW : nullptr;
            const float *xp = NS ? xb + (wave ^ WV) * XW : nullptr;
            BwdPre<NF, CQ, R> nopre;            // (tile-split kernel only)
            if ((l + alt) & 1) layer_bwd<NF, CQ, R, 1, NS, ACT, WV>(W, g, lane, wave, xr, cr, gy, gld, scr, lds, tb, gpl, first, stp, xo, xp, 0, -1, nopre, false, nullptr, nullptr, &fsync);
            else layer_bwd<NF, CQ, R, 0, NS, ACT, WV>(W, g, lane, wave, xr, cr, gy, gld, scr, lds, tb, gpl, first, stp, xo, xp, 0, -1, nopre, false, nullptr, nullptr, &fsync);
        }
        if constexpr (NS) __syncthreads();      // exchange buffers are reused by the next group's first layer
        first = false;
    }
    if constexpr (NS == 1 || (kWideTFlush && NS == 0 && WV == 8)) {       // a wave of this workgroup gave up a bounded wait: the loss says so
        if (__hip_atomic_load(sync + 13, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) wave_sum = __builtin_nanf("");
    }
    if constexpr (WV > kWaves) {        // k_mfma_reduce adds kWaves loss partials per workgroup: fold the owners' sums, fixed order
        __syncthreads();
        if (lane == 0) lds[wave] = wave_sum;
        __syncthreads();
        if (wave == 0 && lane < kWaves) {
            float a = lds[lane];
#pragma unroll
            for (int w = kWaves; w < WV; w += kWaves) a += lds[lane + w];
            losspart[blockIdx.x * kWaves + lane] = a;
        }
    } else {
        if (lane == 0 && role == 0) losspart[blockIdx.x * WV + pw] = wave_sum;
    }
#ifdef RNVP_STAMP
    {
        unsigned long long tk1; STAMP(tk1);
        if (lane == 0 && (blockIdx.x == 0 || blockIdx.x == 100))
            printf("STAMP wg %d wave %d total %llu load %llu fwd %llu bsetup %llu p1 %llu p2 %llu p3 %llu p4 %llu p5 %llu flush: barrier1 %llu sum+store %llu barrier2 %llu btail %llu\n",
                   (int)blockIdx.x, wave, tk1 - tk0, stp.ld, stp.fwd, stp.bsetup, stp.p1, stp.p2, stp.p3, stp.p4, stp.p5, stp.fb1, stp.fsum, stp.bflush, stp.btail);
    }
#endif
}

// BXF (rnvp_shape.precision = RNVP_PREC_BX3, or AUTO where it resolves to BX3): GEMM1 of the FORWARD phase on split-bf16 MFMA
// (rnvp_split.h); the backward keeps f32 -- its split operands do not fit in 256 registers next to the gradient state, and
// the one-wave 512-register form that has the room was slower in every A/B of round 3 (k_mfma_train_bx, removed from the tree
// in round 6: profiles/r03_train_bx_ab.txt, r03_train_bx_experiments.txt and git history keep the record).
// Measured (whole call, 65536 rows): C2 -1.5 %, C3 -2 %, C4 (NF = 8) +2 %: the wide geometry keeps f32 (train_bxf()).
template <int NF, int CQ, int R, int NS, int ACT, bool BXF = false>
__global__ void __launch_bounds__(kWaves * 64 * (1 + NS)) __attribute__((amdgpu_waves_per_eu(RNVP_WPE, RNVP_WPE)))
k_mfma_train(const float *__restrict__ wp, Geo g, int L, int alt, const float *__restrict__ x,
             const float *__restrict__ c, const int64_t *__restrict__ row_index, int64_t n, float inv_B,
             float *gpart, float *losspart, float *scratch, int glayer_floats, Seeds sd) {
    train_body<NF, CQ, R, NS, ACT, BXF>(wp, g, L, alt, x, c, row_index, n, inv_B, gpart, losspart, scratch, glayer_floats, sd);
}

// Wide form: EIGHT row-owning waves per workgroup (no net split), one workgroup per CU.  Where a 65 536-row batch would
// otherwise need two 4-wave workgroups per CU (d > 16: 128 rows per workgroup), this halves the number of per-workgroup
// partial gradients -- the bytes k_mfma_train writes and k_sum_segments reads (C3: 512 x 1.18 MB -> 256 x 1.18 MB) -- at the
// same two waves per SIMD.  Eight LDS slots are added per flush instead of four.
constexpr int kWideWaves = 2 * kWaves;
template <int NF, int CQ, int R, int ACT, bool BXF = false>
__global__ void __launch_bounds__(kWideWaves * 64) __attribute__((amdgpu_waves_per_eu(RNVP_WPE, RNVP_WPE)))
k_mfma_train_wide(const float *__restrict__ wp, Geo g, int L, int alt, const float *__restrict__ x,
                  const float *__restrict__ c, const int64_t *__restrict__ row_index, int64_t n, float inv_B,
                  float *gpart, float *losspart, float *scratch, int glayer_floats, Seeds sd) {
    train_body<NF, CQ, R, 0, ACT, BXF, kWideWaves>(wp, g, L, alt, x, c, row_index, n, inv_B, gpart, losspart, scratch,
                                                          glayer_floats, sd);
}


// ---- tile-split step: batches of up to RNVP_TS_MAX_ROWS rows (d <= 16; half of that for wider rows) ------------------
// A workgroup of kTsWaves waves takes 16 R rows (R = 1 / 2), ALL its waves holding those same row tiles; wave w runs a
// quarter of the hidden tiles of net w >> 2 (layer_forward_ts / layer_bwd with NS == 2).  Small and medium batches are
// latency chains: the row-parallel kernel above gives such a batch one wave pair per 16 rows, each walking 2 * HT
// dependent tile steps per layer; here a layer is HT / 4 tile steps plus one LDS rendezvous, and while the batch needs
// at most one workgroup per CU the step time is that of one workgroup.  C2 flow, fused step (us, tile split / row
// parallel): 32 rows 47 / 103, 256: 51 / 106, 1024: 59 / 107, 4096: 66 / 113, 8192: 82 / 116, 16384: 119 / 120;
// C3 flow (d 32, h 256, L 12): 32 rows 134 / 402, 1024: 161 / 419, 4096: 202 / 435; C4 flow (d 64): 114 / 286, 135 / 294,
// 162 / 305.
// Every workgroup writes the same gradient record and loss partials as k_mfma_train: k_sum_segments / k_mfma_reduce
// follow unchanged.
template <int NF, int CQ, int R, int ACT>
__global__ void __launch_bounds__(kTsWaves * 64) __attribute__((amdgpu_waves_per_eu(RNVP_WPE, RNVP_WPE)))
k_mfma_train_ts(const float *__restrict__ wp, Geo g, int L, int alt, const float *__restrict__ x,
                const float *__restrict__ c, const int64_t *__restrict__ row_index, int64_t n, float inv_B,
                float *gpart, float *losspart, float *scratch, int glayer_floats, Seeds sd) {
    using DM = Dims<NF, CQ>;
    constexpr int D = 8 * NF, CD = 4 * CQ;
    constexpr int XW = R * NF * 64, TBN = DM::template tbn<R, 0>();
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // the wave number as an SGPR where that pays (scalar address arithmetic instead of per-lane: the C2 kernel's static VALU count
    // 1 689 -> 1 506): round 6, same box, per call: C3 at 32 / 1 024 / 8 192 rows -2.3 / -2.0 / -3.0 %, C2 at 8 192 rows -2.0 %, at
    // 1 024 0, at 32 rows +1.0 % -- hence not for d <= 16 with one row tile (profiles/r06_ts_wave_sgpr_ab.txt)
    const int lane = threadIdx.x & 63;
    const int wave = (NF >= 4 || R >= 2) ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : (int)(threadIdx.x >> 6);
    const int q = lane >> 4, r = lane & 15;
    float *tb = lds + wave * TBN;
    float *red = lds + kTsWaves * TBN;                    // 2 x kTsWaves x XW, double buffered by layer parity
    const int tps = (g.HT + kTsSlices - 1) / kTsSlices, slice = wave & (kTsSlices - 1);
    const int tile_lo = slice * tps < g.HT ? slice * tps : g.HT;
    const int tile_hi = tile_lo + tps < g.HT ? tile_lo + tps : g.HT;
    const float prior_c = 0.5f * (float)g.d * kLog2Pi;
    const bool full = (g.d == D) && (g.c == CD) && ((uintptr_t)x & 15) == 0;
    Stamps stp = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // workgroup b: rows [16 R b, 16 R (b + 1)), its own gradient record, scratch records and loss partials
    const int64_t row0 = (int64_t)blockIdx.x * R * 16;
    gpart += (size_t)blockIdx.x * glayer_floats * L;
    scratch += (size_t)blockIdx.x * L * R * 2 * NF * 64;
    float xr[R][2 * NF], cr[R][CQ > 0 ? CQ : 1], ld[R], gy[R][2 * NF], gld[R];
    bool valid[R];
#pragma unroll
    for (int rt = 0; rt < R; ++rt) {
        const int64_t row = row0 + rt * 16 + r;
        valid[rt] = row < n;
        const int64_t src = valid[rt] ? (row_index ? row_index[row] : row) : 0;
        load_row<NF, CQ>(x, c, src, g.d, g.c, full, q, xr[rt], cr[rt]);
        ld[rt] = 0.f;
    }
    // a layer's opening loads are requested one layer ahead (TilePre / BwdPre); d > 32 has no registers for that (spills)
    constexpr bool PRE = NF <= 4;
    TilePre<NF, CQ> pre;
    if (PRE && tile_hi > tile_lo) load_tile_pre<NF, CQ>(wp, g, lane, (wave >> 2) * g.HT + tile_lo, tile_hi - tile_lo, pre);
    for (int l = 0; l < L; ++l) {
        const float *W = wp + (size_t)l * g.layer_floats;
        const float *Wn = l + 1 < L ? W + g.layer_floats : nullptr;
        float *scr = scratch + (size_t)l * R * 2 * NF * 64;
        float *rb = red + (size_t)(l & 1) * kTsWaves * XW;
        if ((l + alt) & 1) layer_forward_ts<NF, CQ, R, 1, 2, ACT>(W, g, lane, wave, tile_lo, tile_hi - tile_lo, rb, xr, cr, ld, scr, Wn, pre, PRE);
        else layer_forward_ts<NF, CQ, R, 0, 2, ACT>(W, g, lane, wave, tile_lo, tile_hi - tile_lo, rb, xr, cr, ld, scr, Wn, pre, PRE);
    }
    __syncthreads();        // wave 0's scratch records; the rendezvous buffers change hands
    float wave_sum = 0.f;
#pragma unroll
    for (int rt = 0; rt < R; ++rt) {
        float ss = 0.f;
#pragma unroll
        for (int v = 0; v < 2 * NF; ++v) ss = fmaf(xr[rt][v], xr[rt][v], ss);
        float l1 = ld[rt];
        l1 += __shfl_xor(l1, 16); l1 += __shfl_xor(l1, 32);
        ss += __shfl_xor(ss, 16); ss += __shfl_xor(ss, 32);
        const float lp = sd.gz ? l1 : l1 + (-0.5f * ss - prior_c);
        float v = (valid[rt] && q == 0) ? lp : 0.f;
        v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
        wave_sum += v;
        const float sc = valid[rt] ? inv_B : 0.f;
        if (sd.gz) {
            const int64_t row = row0 + rt * 16 + r;
#pragma unroll
            for (int u = 0; u < 2 * NF; ++u) {
                const int j = q * 2 * NF + u;
                gy[rt][u] = (valid[rt] && j < g.d) ? sd.gz[row * g.d + j] : 0.f;
            }
        } else {
#pragma unroll
            for (int u = 0; u < 2 * NF; ++u) gy[rt][u] = xr[rt][u] * sc;
        }
        gld[rt] = sd.gld ? (valid[rt] ? sd.gld[row0 + rt * 16 + r] : 0.f) : -sc;
    }
    BwdPre<NF, CQ, R> bpre;          // a layer's opening loads, requested one layer ahead
    const bool have_tiles = PRE && tile_hi > tile_lo;
    if (have_tiles)
        load_bwd_pre<NF, CQ, R>(wp + (size_t)(L - 1) * g.layer_floats, g, lane, scratch + (size_t)(L - 1) * R * 2 * NF * 64, wave >> 2,
                                tile_lo, bpre);
    for (int l = L - 1; l >= 0; --l) {
        const float *W = wp + (size_t)l * g.layer_floats;
        const float *scr = scratch + (size_t)l * R * 2 * NF * 64;
        float *gpl = gpart + (size_t)l * glayer_floats;
        float *rb = red + (size_t)(l & 1) * kTsWaves * XW;
        const float *Wp = l > 0 ? W - g.layer_floats : nullptr;
        const float *sp = l > 0 ? scr - (size_t)R * 2 * NF * 64 : nullptr;
        if ((l + alt) & 1) layer_bwd<NF, CQ, R, 1, 2, ACT>(W, g, lane, wave, xr, cr, gy, gld, scr, lds, tb, gpl, true, stp, rb + wave * XW, rb, tile_lo, tile_hi, bpre, have_tiles, Wp, sp);
        else layer_bwd<NF, CQ, R, 0, 2, ACT>(W, g, lane, wave, xr, cr, gy, gld, scr, lds, tb, gpl, true, stp, rb + wave * XW, rb, tile_lo, tile_hi, bpre, have_tiles, Wp, sp);
    }
    if (sd.gx && wave == 0) {                   // rnvp_backward: d loss / d x (every wave holds the same sums)
        const bool fullg = (g.d == D) && ((uintptr_t)sd.gx & 15) == 0;
#pragma unroll
        for (int rt = 0; rt < R; ++rt)
            if (valid[rt]) store_row<NF>(sd.gx, row0 + rt * 16 + r, g.d, fullg, q, gy[rt]);
    }
    // the loss: every wave computed the same sum; k_mfma_reduce adds kWaves partials per workgroup
    if (lane == 0 && wave < kWaves) losspart[blockIdx.x * kWaves + wave] = wave == 0 ? wave_sum : 0.f;
}
#ifndef RNVP_TRAIN_R2
#define RNVP_TRAIN_R2 4
#endif
// Row tiles per wave.  RMAX (4 / 2 / 1 for NF = 2 / 4 / 8, what the registers allow) is fastest once every CU has a
// workgroup; smaller batches spread over the chip with fewer tiles per wave instead of running long chains on
// a few CUs.  Measured on C2 (one workgroup resident per CU): a workgroup's run time is ~1 : 1.4 : 2.15 for
// R = 1 : 2 : 4 and the step takes ceil(workgroups / 256) such rounds -- pick the cheapest
// (8192 rows: 0.300 -> 0.162 ms, 32768 rows: 0.319 -> 0.218 ms, 65536 rows and up: R = RMAX as before).
// A row's gradient contribution does not depend on R; the summation order over rows does (still
// deterministic for a given batch size).
// The ragged last batch of the C2 epoch (16 960 rows) takes R = 2 on 133 workgroups, half the chip; round 6 measured the two ways
// to give every CU rows with R = 1 (profiles/r06_ragged_wt_ab.txt, rnvp_loss_grad, ms): row-parallel on 265 four-wave workgroups,
// two per CU, 0.197; net split on 256 workgroups of which nine take a second group 0.239; shipped 0.170 -- 1 060 row tiles over
// 1 024 SIMDs need either a second round or a wave that runs both nets of its tile alone.
template <int NF, int CQ> struct TrainRows { static constexpr int value = NF == 2 ? RNVP_TRAIN_R2 : (NF == 4 ? 2 : 1); };

static int pick_rows(int rmax, int64_t n) {
    int best = rmax;
    double best_cost = 0.0;
    for (int R = rmax; R >= 1; R >>= 1) {
        const int64_t rows_per_wg = (int64_t)kWaves * R * 16;
        const int64_t rounds = ((n + rows_per_wg - 1) / rows_per_wg + 255) / 256;
        const double cost = (double)rounds * (R == 1 ? 1.0 : R == 2 ? 1.4 : 2.15);
        if (R == rmax || cost < best_cost) { best = R; best_cost = cost; }
    }
    return best;
}


// forward GEMM1 of the training kernels on split-bf16 MFMA: the caller asked for it (precision bx3, or auto on a shape
// where auto means bx3) and the geometry gains from it
static bool train_bxf(const KShape &k, const Geo &g) { return kTrainBxF && k.prec == RNVP_PREC_BX3 && g.NF <= 4 && g.NI1 > 0; }

template <int NF, int CQ, int R, int NS, int ACT, bool BXF = false>
int launch_train_act(hipStream_t st, const KShape &k, const Geo &g, const TrainPlan &pl, const float *packed,
                    const float *x, const float *c, const int64_t *row_index, int64_t n, float inv_B, float *gpart,
                    float *losspart, float *scratch, int grid, size_t lds_bytes, Seeds sd) {
    if constexpr (!BXF && NF <= 4) {
        if (train_bxf(k, g))
            return launch_train_act<NF, CQ, R, NS, ACT, true>(st, k, g, pl, packed, x, c, row_index, n, inv_B, gpart, losspart,
                                                              scratch, grid, lds_bytes, sd);
    }
    auto kern = k_mfma_train<NF, CQ, R, NS, ACT, BXF>;
    static std::atomic<uint64_t> attr_done{0};          // per kernel instance; one bit per device
    const int arc = allow_big_lds(reinterpret_cast<const void *>(kern), 160 * 1024, attr_done);
    if (arc) return arc;
    note_dispatch(RNVP_PROFILE_TRAIN, "k_mfma_train", NS ? RNVP_VARIANT_NETSPLIT : RNVP_VARIANT_ROWPAR, R, kWaves * (1 + NS), grid,
                  BXF ? RNVP_PREC_BX3 : RNVP_PREC_F32, n);
    {
        const KernelEvents ev(RNVP_PROFILE_TRAIN);      // rnvp_profile_*: this launch's own start / stop stamps when enabled
        hipExtLaunchKernelGGL(kern, dim3(grid), dim3(kWaves * 64 * (1 + NS)), lds_bytes, st, ev.start, ev.stop, 0, packed, g,
                              k.L, k.alt, x, c, row_index, n, inv_B, gpart, losspart, scratch, pl.glayer_floats, sd);
    }
    RNVP_HIP_TRY(hipGetLastError());
    return RNVP_OK;
}

template <int NF, int CQ, int R, int ACT, bool BXF = false>
int launch_train_wide(hipStream_t st, const KShape &k, const Geo &g, const TrainPlan &pl, const float *packed, const float *x,
                      const float *c, const int64_t *row_index, int64_t n, float inv_B, float *gpart, float *losspart,
                      float *scratch, int grid, size_t lds_bytes, Seeds sd) {
    if constexpr (!BXF && NF <= 4) {
        if (train_bxf(k, g))
            return launch_train_wide<NF, CQ, R, ACT, true>(st, k, g, pl, packed, x, c, row_index, n, inv_B, gpart, losspart, scratch,
                                                           grid, lds_bytes, sd);
    }
    auto kern = k_mfma_train_wide<NF, CQ, R, ACT, BXF>;
    static std::atomic<uint64_t> attr_done{0};
    const int arc = allow_big_lds(reinterpret_cast<const void *>(kern), 160 * 1024, attr_done);
    if (arc) return arc;
    note_dispatch(RNVP_PROFILE_TRAIN, "k_mfma_train_wide", RNVP_VARIANT_WIDE, R, kWideWaves, grid, BXF ? RNVP_PREC_BX3 : RNVP_PREC_F32, n);
    {
        const KernelEvents ev(RNVP_PROFILE_TRAIN);
        hipExtLaunchKernelGGL(kern, dim3(grid), dim3(kWideWaves * 64), lds_bytes, st, ev.start, ev.stop, 0, packed, g, k.L, k.alt,
                              x, c, row_index, n, inv_B, gpart, losspart, scratch, pl.glayer_floats, sd);
    }
    RNVP_HIP_TRY(hipGetLastError());
    return RNVP_OK;
}

template <int NF, int CQ, int R, int NS>
int launch_train_ns(hipStream_t st, const KShape &k, const Geo &g, const TrainPlan &pl, const float *packed,
                    const float *x, const float *c, const int64_t *row_index, int64_t n, float inv_B, float *gpart,
                    float *losspart, float *scratch, int grid, size_t lds_bytes, Seeds sd) {
    if (k.act == RNVP_ACT_TANH)
        return launch_train_act<NF, CQ, R, NS, 0>(st, k, g, pl, packed, x, c, row_index, n, inv_B, gpart, losspart, scratch,
                                                  grid, lds_bytes, sd);
    return launch_train_act<NF, CQ, R, NS, 1>(st, k, g, pl, packed, x, c, row_index, n, inv_B, gpart, losspart, scratch, grid,
                                              lds_bytes, sd);
}

#ifndef RNVP_NET_SPLIT
#define RNVP_NET_SPLIT 1
#endif

#ifndef RNVP_TILE_SPLIT
#define RNVP_TILE_SPLIT 1
#endif
#ifndef RNVP_TS_MAX_ROWS
#define RNVP_TS_MAX_ROWS 8192        // 256 workgroups x 32 rows
#endif
template <int NF, int CQ, int R>
int launch_train_ts(hipStream_t st, const KShape &k, const Geo &g, const TrainPlan &pl, const float *packed, const float *x,
                    const float *c, const int64_t *row_index, int64_t n, float inv_B, float *gpart, float *losspart,
                    float *scratch, Seeds sd, int grid) {
    using DM = Dims<NF, CQ>;
    const size_t lds_bytes = ((size_t)kTsWaves * DM::template tbn<R, 0>() + 2 * (size_t)kTsWaves * R * NF * 64) * sizeof(float);
    static std::atomic<uint64_t> attr_done[2] = {{0}, {0}};
    const int arc = k.act == RNVP_ACT_TANH
                        ? allow_big_lds(reinterpret_cast<const void *>(k_mfma_train_ts<NF, CQ, R, 0>), 160 * 1024, attr_done[0])
                        : allow_big_lds(reinterpret_cast<const void *>(k_mfma_train_ts<NF, CQ, R, 1>), 160 * 1024, attr_done[1]);
    if (arc) return arc;
    note_dispatch(RNVP_PROFILE_TRAIN, "k_mfma_train_ts", RNVP_VARIANT_TILESPLIT, R, kTsWaves, grid, RNVP_PREC_F32, n);
    const KernelEvents ev(RNVP_PROFILE_TRAIN);
    if (k.act == RNVP_ACT_TANH)
        hipExtLaunchKernelGGL((k_mfma_train_ts<NF, CQ, R, 0>), dim3(grid), dim3(kTsWaves * 64), lds_bytes, st, ev.start, ev.stop, 0,
                              packed, g, k.L, k.alt, x, c, row_index, n, inv_B, gpart, losspart, scratch, pl.glayer_floats, sd);
    else
        hipExtLaunchKernelGGL((k_mfma_train_ts<NF, CQ, R, 1>), dim3(grid), dim3(kTsWaves * 64), lds_bytes, st, ev.start, ev.stop, 0,
                              packed, g, k.L, k.alt, x, c, row_index, n, inv_B, gpart, losspart, scratch, pl.glayer_floats, sd);
    RNVP_HIP_TRY(hipGetLastError());
    return RNVP_OK;
}


template <int NF, int CQ, int R>
int launch_train_r(hipStream_t st, const KShape &k, const Geo &g, const TrainPlan &pl, const float *packed,
                   const float *x, const float *c, const int64_t *row_index, int64_t n, float inv_B, float *gpart,
                   float *losspart, float *scratch, int *grid_out, Seeds sd, PartialLayout *lay) {
    using DM = Dims<NF, CQ>;
    const int64_t rows_per_wg = (int64_t)kWaves * R * 16;
    const int64_t ngroups = (n + rows_per_wg - 1) / rows_per_wg;
    const int grid = (int)(ngroups < kMaxGridTrain ? ngroups : kMaxGridTrain);
    *grid_out = grid;
    const size_t per_wave = (size_t)DM::template slot<0>() + DM::template tbn<R, 0>();
    TrainPlan p0 = pl;                                   // partial layout of the launches without net split
    p0.glayer_floats = 2 * g.HT * DM::template tblk<0>() + DM::NT2 * 16;
    const size_t per_wave_ns = (size_t)DM::template slot<1>() + DM::template tbn<R, 1>();
    // Net split: while there is at most one workgroup per CU (one wave per SIMD), give every row tile to a PAIR of
    // waves, one per net -- two waves per SIMD without loading any weight fragment twice.
    const size_t lds_ns = (2 * kWaves * per_wave_ns + 2 * 2 * kWaves * (size_t)R * NF * 64) * sizeof(float) + kSyncBytes;
    if (RNVP_NET_SPLIT && ngroups <= 256 && lds_ns <= 160 * 1024) {
        lay->w2c = DM::template w2c<1>() ? 1 : 0;
        lay->glayer_floats = 2 * g.HT * DM::template tblk<1>() + DM::NT2 * 16;
        TrainPlan pn = pl;
        pn.glayer_floats = lay->glayer_floats;
        return launch_train_ns<NF, CQ, R, 1>(st, k, g, pn, packed, x, c, row_index, n, inv_B, gpart, losspart, scratch,
                                             grid, lds_ns, sd);
    }
    lay->w2c = DM::template w2c<0>() ? 1 : 0;
    lay->glayer_floats = p0.glayer_floats;
    if constexpr (kTrainWide && NF == 4) {       // d in (16, 32]: two 4-wave workgroups per CU become one 8-wave workgroup (d = 64: built, 147 KB of LDS,
                                                 // measured 1.206-1.214 against 1.212-1.222 ms at 65 536 rows and slower at 262 144: not taken)
        const size_t lds_wide = kWideWaves * per_wave * sizeof(float) + kSyncBytes;
        if (ngroups > 256 && lds_wide <= 160 * 1024) {
            const int64_t rows_wide = (int64_t)kWideWaves * R * 16;
            const int64_t gw = (n + rows_wide - 1) / rows_wide;
            const int gridw = (int)(gw < kMaxGridTrain / 2 ? gw : kMaxGridTrain / 2);
            *grid_out = gridw;
            if (k.act == RNVP_ACT_TANH)
                return launch_train_wide<NF, CQ, R, 0>(st, k, g, p0, packed, x, c, row_index, n, inv_B, gpart, losspart, scratch,
                                                       gridw, lds_wide, sd);
            return launch_train_wide<NF, CQ, R, 1>(st, k, g, p0, packed, x, c, row_index, n, inv_B, gpart, losspart, scratch, gridw,
                                                   lds_wide, sd);
        }
    }
    return launch_train_ns<NF, CQ, R, 0>(st, k, g, p0, packed, x, c, row_index, n, inv_B, gpart, losspart, scratch, grid,
                                         kWaves * per_wave * sizeof(float) + kSyncBytes, sd);
}

// rows the tile-split kernel takes for a geometry (launch_train below)
// RNVP_TS_R2_NF4: d in (16, 32] takes two row tiles per workgroup from 4097 to 8192 rows (one round of 256 workgroups) -- the
// per-rank batch of C3 under data parallelism over 8 GPUs (global batch 65 536); the row-parallel kernel gave
// that batch 128 workgroups, half the chip
#ifndef RNVP_TS_R2_NF4
#define RNVP_TS_R2_NF4 1
#endif
static int64_t ts_max_rows(const Geo &g) {
    if (!RNVP_TILE_SPLIT || g.HT < 3) return 0;
    return (g.NF == 2 || (g.NF == 4 && RNVP_TS_R2_NF4)) ? RNVP_TS_MAX_ROWS : RNVP_TS_MAX_ROWS / 2;
}

template <int NF, int CQ>
int launch_train(hipStream_t st, const KShape &k, const Geo &g, const TrainPlan &pl, const float *packed,
                 const float *x, const float *c, const int64_t *row_index, int64_t n, float inv_B, float *gpart,
                 float *losspart, float *scratch, int *grid_out, Seeds sd, PartialLayout *lay) {
    constexpr int RMAX = TrainRows<NF, CQ>::value;
    if constexpr (RNVP_TILE_SPLIT) {
        // d <= 16: one workgroup per 16 rows up to 4096 rows, per 32 rows up to 8192 (numbers above k_mfma_train_ts); wider
        // rows: 16 rows per workgroup, up to 4096.  With two hidden tiles per net or fewer there is nothing to split
        // (h = 32 measured 47 vs 48 us).
        constexpr bool kR2 = NF == 2 || (NF == 4 && RNVP_TS_R2_NF4);
        constexpr int64_t kMaxRows = kR2 ? RNVP_TS_MAX_ROWS : RNVP_TS_MAX_ROWS / 2;
        if (n <= kMaxRows && g.HT >= 3) {
            const int R = (kR2 && n > RNVP_TS_MAX_ROWS / 2) ? 2 : 1;
            const int grid = (int)((n + 16 * R - 1) / (16 * R));
            lay->w2c = 0;
            lay->glayer_floats = pl.glayer_floats;
            *grid_out = grid;
            if constexpr (kR2) {
                if (R == 2) return launch_train_ts<NF, CQ, 2>(st, k, g, pl, packed, x, c, row_index, n, inv_B, gpart, losspart, scratch, sd, grid);
            }
            return launch_train_ts<NF, CQ, 1>(st, k, g, pl, packed, x, c, row_index, n, inv_B, gpart, losspart, scratch, sd, grid);
        }
    }
    const int R = pick_rows(RMAX, n);
#define RNVP_ROWS(r)                                                                                              \
    if constexpr (RMAX >= r) {                                                                                    \
        if (R == r)                                                                                               \
            return launch_train_r<NF, CQ, r>(st, k, g, pl, packed, x, c, row_index, n, inv_B, gpart, losspart, scratch, \
                                             grid_out, sd, lay);                                                  \
    }
    RNVP_ROWS(4) RNVP_ROWS(2) RNVP_ROWS(1)
#undef RNVP_ROWS
    return RNVP_EUNSUPPORTED;
}
}  // namespace
}  // namespace mfma
}  // namespace rnvp
