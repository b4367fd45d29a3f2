// rnvp_mfma_pack.h -- the packed (MFMA fragment order) copy of one layer's parameters.
// Shared by k_pack_weights (rnvp_mfma.hip: the whole block from the flat parameters in global memory) and by the training
// step's finish kernel (rnvp_mfma_train.hip: a workgroup re-packs the slots of the hidden tile whose parameters it has just
// updated, from its LDS copy).  Slot orders: rnvp_mfma.h.  Every packed array but the second-Linear bias is indexed
// [tile = net * HT + ht][...]: pack_slot takes (array, net, ht, index inside the tile's part), so a caller that knows its tile
// pays no integer division; pack_value decodes a flat index into that form.
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

enum PackArray { kPkA1 = 0, kPkB1, kPkA2, kPkB2, kPkA2T, kPkA1T, kPkA2X, kPkA1X, kPkA1S, kPkA2TS, kPkArrays };

// floats of ONE tile's part of a packed array (kPkB2: of the whole array) and the array's offset in the layer block
__host__ __device__ inline int pack_per_tile(const Geo &g, int arr) {
    switch (arr) {
        case kPkA1: return g.K4 * 256;
        case kPkB1: return 16;
        case kPkA2: return g.OTL * 256;
        case kPkB2: return g.NT2 * 16;
        case kPkA2T: return g.OTL * 256;
        case kPkA1T: return g.MTI * 256;
        case kPkA2X: return g.NF == 2 ? 512 : 0;
        case kPkA1X: return g.NF == 2 ? 512 : 0;
        case kPkA1S: return g.NI1 * 256;
        default: return g.NI2 * 256;
    }
}
__host__ __device__ inline int pack_offset(const Geo &g, int arr) {
    switch (arr) {
        case kPkA1: return g.oA1;
        case kPkB1: return g.oB1;
        case kPkA2: return g.oA2;
        case kPkB2: return g.oB2;
        case kPkA2T: return g.oA2T;
        case kPkA1T: return g.oA1T;
        case kPkA2X: return g.oA2X;
        case kPkA1X: return g.oA1X;
        case kPkA1S: return g.oA1S;
        default: return g.oA2TS;
    }
}

// value of slot j of tile (net, ht)'s part of packed array `arr` in a layer of parity pc = (l + alt) & 1
template <class Src>
__device__ __forceinline__ float pack_slot(const KShape &k, const Geo &g, int pc, int arr, int net, int ht, int j, const Src &src) {
    const int h = k.nout[0];                                  // REAL sizes: flat indexing; padded slots -> 0
#define RNVP_W1(hid, col) (((hid) < h && (col) >= 0) ? src.w1(net, hid, col) : 0.f)
#define RNVP_W2(feat, hid) (((hid) < h && (feat) < k.d) ? src.w2(net, feat, hid) : 0.f)
    // real input column of a padded feature / condition slot, or -1
    auto xcol = [&](int feat) { return feat < k.d ? feat : -1; };
    auto ccol = [&](int ci) { return ci < k.c ? k.d + ci : -1; };
    const float tscale = k.act == RNVP_ACT_TANH ? kTanhScale : 1.0f;      // tanh: W1, b1 pre-scaled, see tanh4
    const int e = j & 3, lane = (j >> 2) & 63, sub = j >> 8;            // [sub][lane][4] inside the tile's part
    const int q = lane >> 4, i = lane & 15;
    switch (arr) {
        case kPkA1: {                                          // A1 [tile][k4][lane][4]
            const int kk = 4 * sub + e, hid = 16 * ht + i;
            int col;
            if (kk < g.NF) col = xcol(feat_cond(g.NF, q, kk, pc));
            else if (kk < g.KS1) col = ccol(q * g.CQ + (kk - g.NF));
            else return 0.f;
            return tscale * RNVP_W1(hid, col);
        }
        case kPkB1: {                                          // bias1 [tile][q][4]
            const int hid = 16 * ht + 4 * ((j >> 2) & 3) + e;
            return hid < h ? tscale * src.b1(net, hid) : 0.f;
        }
        case kPkA2: {                                          // A2 [tile][otl][lane][4 rho]
            const int hid = 16 * ht + 4 * q + e;
            const int qo = i >> 2, ro = i & 3;
            int f, net_out;
            if (g.NF >= 4) { f = 4 * sub + ro; net_out = net; }
            else { f = ro & 1; net_out = ro >> 1; }
            if (net_out != net) return 0.f;
            return RNVP_W2(feat_trans(g.NF, qo, f, pc), hid);
        }
        case kPkA2T: {                                         // A2T [tile][otl][lane][4 rho]
            const int hid = 16 * ht + i;
            int f, net_out;
            if (g.NF >= 4) { f = 4 * sub + e; net_out = net; }
            else { f = e & 1; net_out = e >> 1; }
            if (net_out != net) return 0.f;
            return RNVP_W2(feat_trans(g.NF, q, f, pc), hid);
        }
        case kPkA1T: {                                         // A1T [tile][mt][lane][4 rho]
            const int hid = 16 * ht + 4 * q + e;
            const int qi = i >> 2, ri = i & 3;
            int f;
            if (g.NF >= 4) f = 4 * sub + ri;
            else { if (ri >= 2) return 0.f; f = ri; }
            return RNVP_W1(hid, xcol(feat_cond(g.NF, qi, f, pc)));
        }
        case kPkA2X: {                                         // A2X [tile][og][lane][4 rho]  (d == 16)
            const int hid = 16 * ht + 4 * q + e, i4 = lane & 3;
            return RNVP_W2(feat_trans(g.NF, 2 * (sub & 1) + (i4 >> 1), i4 & 1, pc), hid);
        }
        case kPkA1X: {                                         // A1X [tile][og][lane][4 rho]  (d == 16)
            // 4x4x1 blocks: lane (q, r), i = r & 3 supplies W1[hid 16t+4q+rho][conditioning feature (og, i)],
            // feature owner q_f = 2*og + (i >> 1), slot f = i & 1
            const int hid = 16 * ht + 4 * q + e, i4 = lane & 3;
            return RNVP_W1(hid, xcol(feat_cond(g.NF, 2 * (sub & 1) + (i4 >> 1), i4 & 1, pc)));
        }
        case kPkA1S: {                                         // A1S [tile][NI1][lane][4 dwords]: A1 split bf16 (same k order)
            const int D = 4 * sub + e, kk = D / 3, p = D % 3;
            const int hid = 16 * ht + i;
            int col;
            if (kk < g.NF) col = xcol(feat_cond(g.NF, q, kk, pc));
            else if (kk < g.KS1) col = ccol(q * g.CQ + (kk - g.NF));
            else return 0.f;
            return __uint_as_float(split::a_dword(tscale * RNVP_W1(hid, col), p));
        }
        case kPkA2TS: {                                        // A2TS [tile][NI2][lane][4 dwords]: W2^T of ONE net, split bf16
            // lane (q, i): hidden unit 16t + i; the lane's slot list holds the NF transformed features of lane group q
            const int D = 4 * sub + e, v = D / 3, p = D % 3;
            if (v >= g.NF) return 0.f;
            const int hid = 16 * ht + i;
            return __uint_as_float(split::a_dword(RNVP_W2(feat_trans(g.NF, q, v, pc), hid), p));
        }
        default: {                                             // kPkB2: bias2 [ot][q][4] (no tile index: net, ht unused)
            const int ro = j & 3, qo = (j >> 2) & 3, ot = j >> 4;
            int f, bnet;
            if (g.NF >= 4) { bnet = ot >= g.OTL; f = 4 * (ot - bnet * g.OTL) + ro; }
            else { bnet = ro >> 1; f = ro & 1; }
            const int feat = feat_trans(g.NF, qo, f, pc);
            return feat < k.d ? src.b2(bnet, feat) : 0.f;
        }
    }
#undef RNVP_W1
#undef RNVP_W2
}

// value of flat slot idx of layer l's packed block
template <class Src>
__device__ float pack_value(const KShape &k, const Geo &g, int l, int idx, const Src &src) {
    const int pc = (l + k.alt) & 1;
    // the arrays lie in this order in the block (make_geo): A1 B1 A2 B2 A2T A1T A2X A1X A1S A2TS
    int arr = kPkA2TS;
    if (idx < g.oB1) arr = kPkA1;
    else if (idx < g.oA2) arr = kPkB1;
    else if (idx < g.oB2) arr = kPkA2;
    else if (idx < g.oA2T) arr = kPkB2;
    else if (idx < g.oA1T) arr = kPkA2T;
    else if (idx < g.oA2X) arr = kPkA1T;
    else if (idx < g.oA1X) arr = kPkA2X;
    else if (idx < g.oA1S) arr = kPkA1X;
    else if (idx < g.oA2TS) arr = kPkA1S;
    const int j = idx - pack_offset(g, arr);
    if (arr == kPkB2) return pack_slot(k, g, pc, arr, 0, 0, j, src);
    const int per = pack_per_tile(g, arr);
    const int tile = j / per, net = tile / g.HT;
    return pack_slot(k, g, pc, arr, net, tile - net * g.HT, j - tile * per, src);
}

}  // namespace mfma
}  // namespace rnvp
