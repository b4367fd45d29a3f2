// rnvp_mfma_pack.h -- the packed (MFMA fragment order) copy of one layer's parameters: value of packed slot `idx`.
// Shared by k_pack_weights (rnvp_mfma.hip: the whole block from the flat parameters in global memory) and by the training
// step's finish kernel (rnvp_mfma_train.hip: a workgroup re-packs the slots of the hidden tile whose parameters it has just
// updated, from its LDS copy).  Slot orders: rnvp_mfma.h.
#pragma once
#include "rnvp_mfma_layer.h"

namespace rnvp {
namespace mfma {

// parameter source: layer l of the flat reference-order buffer (include/rnvp_hip.h "params")
struct FlatParams {
    const float *pl;       // the layer's block: net t, then net s
    int npn, nin, h, w1o, bo1, w2o, bo2;
    __device__ FlatParams(const KShape &k, const float *params, int l)
        : pl(params + (size_t)l * 2 * k.npn), npn(k.npn), nin(k.d + k.c), h(k.nout[0]), w1o(k.woff[0]), bo1(k.boff[0]),
          w2o(k.woff[1]), bo2(k.boff[1]) {}
    __device__ float w1(int net, int hid, int col) const { return pl[net * npn + w1o + hid * nin + col]; }
    __device__ float w2(int net, int feat, int hid) const { return pl[net * npn + w2o + feat * h + hid]; }
    __device__ float b1(int net, int hid) const { return pl[net * npn + bo1 + hid]; }
    __device__ float b2(int net, int feat) const { return pl[net * npn + bo2 + feat]; }
};

template <class Src>
__device__ float pack_value(const KShape &k, const Geo &g, int l, int idx, const Src &src) {
    const int pc = (l + k.alt) & 1;
    const int h = k.nout[0];                                  // REAL sizes: flat indexing; padded slots -> 0
#define RNVP_W1(net, hid, col) (((hid) < h && (col) >= 0) ? src.w1(net, hid, col) : 0.f)
#define RNVP_W2(net, feat, hid) (((hid) < h && (feat) < k.d) ? src.w2(net, feat, hid) : 0.f)
    // real input column of a padded feature / condition slot, or -1
    auto xcol = [&](int feat) { return feat < k.d ? feat : -1; };
    auto ccol = [&](int ci) { return ci < k.c ? k.d + ci : -1; };
    if (idx < g.oB1) {                                     // A1 [tile][k4][lane][4]
        const int e = idx & 3, lane = (idx >> 2) & 63, rest = idx >> 8;
        const int k4 = rest % g.K4, tile = rest / g.K4;
        const int kk = 4 * k4 + e, q = lane >> 4, i = lane & 15;
        const int net = tile / g.HT, hid = 16 * (tile % g.HT) + i;
        int col;
        if (kk < g.NF) col = xcol(feat_cond(g.NF, q, kk, pc));
        else if (kk < g.KS1) col = ccol(q * g.CQ + (kk - g.NF));
        else return 0.f;
        return (k.act == RNVP_ACT_TANH ? kTanhScale : 1.0f) * RNVP_W1(net, hid, col);   // tanh: pre-scaled, see tanh4
    }
    if (idx < g.oA2) {                                     // bias1 [tile][q][4]
        const int j = idx - g.oB1;
        const int e = j & 3, q = (j >> 2) & 3, tile = j >> 4;
        const int net = tile / g.HT, hid = 16 * (tile % g.HT) + 4 * q + e;
        return hid < h ? (k.act == RNVP_ACT_TANH ? kTanhScale : 1.0f) * src.b1(net, hid) : 0.f;
    }
    if (idx < g.oB2) {                                     // A2 [tile][otl][lane][4 rho]
        const int j = idx - g.oA2;
        const int rho = j & 3, lane = (j >> 2) & 63, rest = j >> 8;
        const int otl = rest % g.OTL, tile = rest / g.OTL;
        const int q = lane >> 4, i = lane & 15, net = tile / g.HT;
        const int hid = 16 * (tile % g.HT) + 4 * q + rho;
        const int qo = i >> 2, ro = i & 3;
        int f, net_out;
        if (g.NF >= 4) { f = 4 * otl + ro; net_out = net; }
        else { f = ro & 1; net_out = ro >> 1; }
        if (net_out != net) return 0.f;
        return RNVP_W2(net, feat_trans(g.NF, qo, f, pc), hid);
    }
    if (idx < g.oA2T) {                                    // bias2 [ot][q][4]
        const int j = idx - g.oB2;
        const int ro = j & 3, qo = (j >> 2) & 3, ot = j >> 4;
        int f, net;
        if (g.NF >= 4) { net = ot / g.OTL; f = 4 * (ot % g.OTL) + ro; }
        else { net = ro >> 1; f = ro & 1; }
        const int feat = feat_trans(g.NF, qo, f, pc);
        return feat < k.d ? src.b2(net, feat) : 0.f;
    }
    if (idx >= g.oA2TS) {                                  // A2TS [tile][NI2][lane][4 dwords]: W2^T of ONE net, split bf16
        // lane (q, i): hidden unit 16t + i; the lane's slot list holds the NF transformed features of lane group q
        const int j = idx - g.oA2TS;
        const int e = j & 3, lane = (j >> 2) & 63, rest = j >> 8;
        const int ni = rest % g.NI2, tile = rest / g.NI2;
        const int D = 4 * ni + e, v = D / 3, p = D % 3;
        if (v >= g.NF) return 0.f;
        const int q = lane >> 4, i = lane & 15, net = tile / g.HT;
        const int hid = 16 * (tile % g.HT) + i;
        return __uint_as_float(split::a_dword(RNVP_W2(net, feat_trans(g.NF, q, v, pc), hid), p));
    }
    if (idx >= g.oA1S) {                                   // A1S [tile][NI1][lane][4 dwords]: A1 split bf16 (same k order)
        const int j = idx - g.oA1S;
        const int e = j & 3, lane = (j >> 2) & 63, rest = j >> 8;
        const int ni = rest % g.NI1, tile = rest / g.NI1;
        const int D = 4 * ni + e, kk = D / 3, p = D % 3;
        const int q = lane >> 4, i = lane & 15;
        const int net = tile / g.HT, hid = 16 * (tile % g.HT) + i;
        int col;
        if (kk < g.NF) col = xcol(feat_cond(g.NF, q, kk, pc));
        else if (kk < g.KS1) col = ccol(q * g.CQ + (kk - g.NF));
        else return 0.f;
        return __uint_as_float(split::a_dword((k.act == RNVP_ACT_TANH ? kTanhScale : 1.0f) * RNVP_W1(net, hid, col), p));
    }
    if (idx < g.oA1T) {                                    // A2T [tile][otl][lane][4 rho]
        const int j = idx - g.oA2T;
        const int rho = j & 3, lane = (j >> 2) & 63, rest = j >> 8;
        const int otl = rest % g.OTL, tile = rest / g.OTL;
        const int q = lane >> 4, i = lane & 15, net = tile / g.HT;
        const int hid = 16 * (tile % g.HT) + i;
        int f, net_out;
        if (g.NF >= 4) { f = 4 * otl + rho; net_out = net; }
        else { f = rho & 1; net_out = rho >> 1; }
        if (net_out != net) return 0.f;
        return RNVP_W2(net, feat_trans(g.NF, q, f, pc), hid);
    }
    if (idx >= g.oA1X) {                                   // A1X [tile][og][lane][4 rho]  (d == 16)
        // 4x4x1 blocks: lane (q, r), i = r & 3 supplies W1[hid 16t+4q+rho][conditioning feature (og, i)],
        // feature owner q_f = 2*og + (i >> 1), slot f = i & 1
        const int j = idx - g.oA1X;
        const int rho = j & 3, lane = (j >> 2) & 63, rest = j >> 8;
        const int og = rest & 1, tile = rest >> 1;
        const int q = lane >> 4, i = lane & 3, net = tile / g.HT;
        const int hid = 16 * (tile % g.HT) + 4 * q + rho;
        return RNVP_W1(net, hid, xcol(feat_cond(g.NF, 2 * og + (i >> 1), i & 1, pc)));
    }
    if (idx >= g.oA2X) {                                   // A2X [tile][og][lane][4 rho]  (d == 16)
        const int j = idx - g.oA2X;
        const int rho = j & 3, lane = (j >> 2) & 63, rest = j >> 8;
        const int og = rest & 1, tile = rest >> 1;
        const int q = lane >> 4, i = lane & 3, net = tile / g.HT;
        const int hid = 16 * (tile % g.HT) + 4 * q + rho;
        return RNVP_W2(net, feat_trans(g.NF, 2 * og + (i >> 1), i & 1, pc), hid);
    }
    {                                                      // A1T [tile][mt][lane][4 rho]
        const int j = idx - g.oA1T;
        const int rho = j & 3, lane = (j >> 2) & 63, rest = j >> 8;
        const int mt = rest % g.MTI, tile = rest / g.MTI;
        const int q = lane >> 4, i = lane & 15, net = tile / g.HT;
        const int hid = 16 * (tile % g.HT) + 4 * q + rho;
        const int qi = i >> 2, ri = i & 3;
        int f;
        if (g.NF >= 4) f = 4 * mt + ri;
        else { if (ri >= 2) return 0.f; f = ri; }
        return RNVP_W1(net, hid, xcol(feat_cond(g.NF, qi, f, pc)));
    }
#undef RNVP_W1
#undef RNVP_W2
}


}  // namespace mfma
}  // namespace rnvp
