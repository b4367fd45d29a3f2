/*
 * rnvp_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, CPU restatement of the conditional RealNVP hot path of
 * hse-cs/probaforms @ 2024_10_08.  It exists only so that the hand-written
 * HIP kernels in probaforms_amd/csrc can be checked against something that is
 * (a) independent of them and (b) itself pinned against the real reference:
 * the .npz fixtures under tests/golden were produced by importing the reference in the build
 * container (tests/golden/make_golden.py) and tests/test_oracle_golden.py
 * checks every function below against them.  => parity is PINNED by those
 * fixtures; the reference's own test-suite pins nothing numerically
 * (/root/reference/tests/test_models.py:18,28 assert shapes only).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  Nothing under probaforms_amd/ imports it.
 *
 * The arithmetic the reference performs lives in PyTorch (third-party,
 * pyproject.toml:18 `torch = "^2.0.0"`, oracle container has 2.10.0): the
 * functions here restate the published semantics of nn.Linear (y = x W^T + b),
 * nn.Tanh / nn.ReLU, torch.exp, sum(dim=-1), MultivariateNormal(0, I).log_prob
 * and torch.optim.Adam (single-tensor, L2 weight decay), in the order the
 * reference calls them.
 *
 * Build: `make -C oracle` -> oracle/_build/librnvp_oracle32.so (REAL=float)
 *                            oracle/_build/librnvp_oracle64.so (REAL=double, referee)
 *
 * Parameter layout ("flat reference order" = order of nf.parameters(), i.e.
 * the state_dict order of /root/reference/probaforms/models/realnvp.py:69-70,
 * 196-204): for layer l = 0..L-1: net t, then net s; inside a net, for each
 * Linear k = 0..n_hidden: weight [out_k, in_k] row-major, then bias [out_k].
 * in_0 = d + c, out_k = hidden[k] (k < n_hidden), out_last = d.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef ORACLE_REAL
#define ORACLE_REAL float
#endif
typedef ORACLE_REAL real;

#define RNVP_MAX_HIDDEN 8

typedef struct {
    int32_t L;          /* number of coupling layers                       */
    int32_t d;          /* var_size                                        */
    int32_t c;          /* cond_size (0 when C is None)                    */
    int32_t n_hidden;   /* len(hidden)                                     */
    int32_t hidden[RNVP_MAX_HIDDEN];
    int32_t act;        /* 0 = tanh, 1 = relu (realnvp.py:32-37)           */
} rnvp_shape;

/* tanh in the working precision (float build must round like a float tanh) */
static real tanh_real(real v) {
#if defined(ORACLE_IS_DOUBLE)
    return tanh(v);
#else
    return tanhf(v);
#endif
}
static real exp_real(real v) {
#if defined(ORACLE_IS_DOUBLE)
    return exp(v);
#else
    return expf(v);
#endif
}

/* ---- shape helpers -------------------------------------------------------- */

/* parameters of ONE s- or t-net (gen_network, realnvp.py:19-43) */
static size_t net_param_count(const rnvp_shape *s) {
    size_t n = 0;
    int in = s->d + s->c;
    for (int k = 0; k < s->n_hidden; ++k) {
        n += (size_t)s->hidden[k] * in + s->hidden[k];
        in = s->hidden[k];
    }
    n += (size_t)s->d * in + s->d;
    return n;
}

size_t rnvp_oracle_param_count(const rnvp_shape *s) {
    return 2 * net_param_count(s) * (size_t)s->L;
}

static int max_width(const rnvp_shape *s) {
    int w = s->d + s->c;
    for (int k = 0; k < s->n_hidden; ++k)
        if (s->hidden[k] > w) w = s->hidden[k];
    if (s->d > w) w = s->d;
    return w;
}

/* ---- one s/t net forward on one row (realnvp.py:19-43) --------------------
 * acts: optional storage of every post-activation hidden vector, laid out
 *       consecutively (sum(hidden) reals); NULL to discard.
 * tmp0/tmp1: scratch of max_width reals each.                               */
static void net_forward_row(const rnvp_shape *s, const real *p, const real *in,
                            real *out, real *acts, real *tmp0, real *tmp1) {
    const real *cur = in;
    int nin = s->d + s->c;
    real *bufs[2] = {tmp0, tmp1};
    int which = 0;
    for (int k = 0; k < s->n_hidden; ++k) {
        int nout = s->hidden[k];
        const real *W = p, *b = p + (size_t)nout * nin;
        real *dst = acts ? acts : bufs[which];
        for (int o = 0; o < nout; ++o) {
            real acc = 0;
            for (int i = 0; i < nin; ++i) acc += cur[i] * W[(size_t)o * nin + i];
            acc += b[o];                              /* F.linear: x W^T + b  */
            dst[o] = (s->act == 0) ? tanh_real(acc) : (acc > 0 ? acc : (real)0);
        }
        p += (size_t)nout * nin + nout;
        cur = dst;
        if (acts) acts += nout; else which ^= 1;
        nin = nout;
    }
    {
        int nout = s->d;
        const real *W = p, *b = p + (size_t)nout * nin;
        for (int o = 0; o < nout; ++o) {
            real acc = 0;
            for (int i = 0; i < nin; ++i) acc += cur[i] * W[(size_t)o * nin + i];
            out[o] = acc + b[o];
        }
    }
}

/* ---- RealNVPLayer.f  (realnvp.py:73-101) ----------------------------------
 * p_layer points at this layer's parameters (net t then net s).
 * mask: d bytes in {0,1}.  c may be NULL iff s->c == 0.                      */
void rnvp_oracle_layer_f(const rnvp_shape *s, const real *p_layer, const uint8_t *mask,
                         const real *x, const real *c, int64_t B,
                         real *x_new, real *log_det) {
    const int d = s->d, nc = s->c, W = max_width(s);
    const size_t npn = net_param_count(s);
    real *xc = (real *)malloc(sizeof(real) * (size_t)(d + nc + 2 * d + 2 * W));
    real *T = xc + d + nc, *S = T + d, *t0 = S + d, *t1 = t0 + W;
    for (int64_t r = 0; r < B; ++r) {
        const real *xr = x + r * d;
        for (int j = 0; j < d; ++j) xc[j] = xr[j] * (real)mask[j];         /* :92 X*mask   */
        for (int j = 0; j < nc; ++j) xc[d + j] = c[r * nc + j];            /* :92 cat      */
        net_forward_row(s, p_layer, xc, T, NULL, t0, t1);                  /* :96 nn_t     */
        net_forward_row(s, p_layer + npn, xc, S, NULL, t0, t1);            /* :97 nn_s     */
        real ld = 0;
        for (int j = 0; j < d; ++j) {
            real m = (real)mask[j];
            /* :99  (X*exp(S)+T)*(1-mask) + X*mask */
            x_new[r * d + j] = (xr[j] * exp_real(S[j]) + T[j]) * ((real)1 - m) + xr[j] * m;
            ld += S[j] * ((real)1 - m);                                    /* :100         */
        }
        log_det[r] = ld;
    }
    free(xc);
}

/* ---- RealNVPLayer.g  (realnvp.py:104-129) --------------------------------- */
void rnvp_oracle_layer_g(const rnvp_shape *s, const real *p_layer, const uint8_t *mask,
                         const real *x, const real *c, int64_t B, real *x_new) {
    const int d = s->d, nc = s->c, W = max_width(s);
    const size_t npn = net_param_count(s);
    real *xc = (real *)malloc(sizeof(real) * (size_t)(d + nc + 2 * d + 2 * W));
    real *T = xc + d + nc, *S = T + d, *t0 = S + d, *t1 = t0 + W;
    for (int64_t r = 0; r < B; ++r) {
        const real *xr = x + r * d;
        for (int j = 0; j < d; ++j) xc[j] = xr[j] * (real)mask[j];         /* :121 */
        for (int j = 0; j < nc; ++j) xc[d + j] = c[r * nc + j];
        net_forward_row(s, p_layer, xc, T, NULL, t0, t1);                  /* :125 */
        net_forward_row(s, p_layer + npn, xc, S, NULL, t0, t1);            /* :126 */
        for (int j = 0; j < d; ++j) {
            real m = (real)mask[j];
            /* :128 ((X-T)*exp(-S))*(1-mask) + X*mask */
            x_new[r * d + j] = ((xr[j] - T[j]) * exp_real(-S[j])) * ((real)1 - m) + xr[j] * m;
        }
    }
    free(xc);
}

/* ---- NormalizingFlow.log_prob (nflow.py:90-117), per-sample and mean ------
 * z_out [B,d] (may be NULL), logp [B] (may be NULL), mean_out (may be NULL).
 * prior = MultivariateNormal(0, I): log_prob(z) = -0.5*(d*ln(2pi) + |z|^2)
 * (realnvp.py:189-191; closed form measured bit-equal).          */
void rnvp_oracle_log_prob(const rnvp_shape *s, const real *params, const uint8_t *masks,
                          const real *x, const real *c, int64_t B,
                          real *z_out, real *logp, real *mean_out) {
    const int d = s->d;
    const size_t npl = 2 * net_param_count(s);
    real *cur = (real *)malloc(sizeof(real) * (size_t)B * d * 2);
    real *nxt = cur + (size_t)B * d;
    real *ld = (real *)malloc(sizeof(real) * (size_t)B * 2);
    real *acc = ld + B;
    memcpy(cur, x, sizeof(real) * (size_t)B * d);
    for (int64_t r = 0; r < B; ++r) acc[r] = 0;
    for (int l = 0; l < s->L; ++l) {                                       /* nflow.py:109 */
        rnvp_oracle_layer_f(s, params + l * npl, masks + (size_t)l * d, cur, c, B, nxt, ld);
        for (int64_t r = 0; r < B; ++r) acc[r] = (l == 0) ? ld[r] : acc[r] + ld[r];   /* :111-114 */
        real *t = cur; cur = nxt; nxt = t;
    }
    const real half_log2pi_d = (real)(0.5 * (double)d * 1.8378770664093453); /* d/2 * ln(2pi) */
    double tot = 0;
    for (int64_t r = 0; r < B; ++r) {
        real ss = 0;
        for (int j = 0; j < d; ++j) ss += cur[r * d + j] * cur[r * d + j];
        real lp = acc[r] + (-(real)0.5 * ss - half_log2pi_d);              /* nflow.py:115 */
        if (logp) logp[r] = lp;
        tot += (double)lp;
    }
    if (mean_out) *mean_out = (real)(tot / (double)B);                     /* nflow.py:117 */
    if (z_out) memcpy(z_out, cur, sizeof(real) * (size_t)B * d);
    free(ld);
    free(cur < nxt ? cur : nxt);
}

/* ---- NormalizingFlow.sample with z given (nflow.py:141-145) --------------- */
void rnvp_oracle_sample(const rnvp_shape *s, const real *params, const uint8_t *masks,
                        const real *z, const real *c, int64_t n, real *x_out) {
    const int d = s->d;
    const size_t npl = 2 * net_param_count(s);
    real *tmp = (real *)malloc(sizeof(real) * (size_t)n * d);
    real *cur = x_out, *nxt = tmp;
    memcpy(cur, z, sizeof(real) * (size_t)n * d);
    for (int l = s->L - 1; l >= 0; --l) {                                  /* layers[::-1] */
        rnvp_oracle_layer_g(s, params + l * npl, masks + (size_t)l * d, cur, c, n, nxt);
        real *t = cur; cur = nxt; nxt = t;
    }
    if (cur != x_out) memcpy(x_out, cur, sizeof(real) * (size_t)n * d);
    free(tmp);
}

/* ---- backward of one net on one row ---------------------------------------
 * Given d(loss)/d(out) for the net output, accumulate parameter gradients into
 * g (same layout as p) and ADD d(loss)/d(in) into gin[d+c].                  */
static void net_backward_row(const rnvp_shape *s, const real *p, real *g, const real *in,
                             const real *acts, const real *gout, real *gin,
                             real *tmp0, real *tmp1) {
    /* offsets of every Linear */
    size_t off[RNVP_MAX_HIDDEN + 1];
    int nins[RNVP_MAX_HIDDEN + 1], nouts[RNVP_MAX_HIDDEN + 1];
    size_t aoff[RNVP_MAX_HIDDEN + 1];
    int nin = s->d + s->c;
    size_t o = 0, ao = 0;
    for (int k = 0; k <= s->n_hidden; ++k) {
        int nout = (k < s->n_hidden) ? s->hidden[k] : s->d;
        off[k] = o; nins[k] = nin; nouts[k] = nout; aoff[k] = ao;
        o += (size_t)nout * nin + nout;
        if (k < s->n_hidden) ao += nout;
        nin = nout;
    }
    real *gcur = tmp0, *gprev = tmp1;
    for (int j = 0; j < s->d; ++j) gcur[j] = gout[j];
    for (int k = s->n_hidden; k >= 0; --k) {
        const int ni = nins[k], no = nouts[k];
        const real *W = p + off[k];
        real *gW = g + off[k], *gb = gW + (size_t)no * ni;
        const real *inp = (k == 0) ? in : acts + aoff[k - 1];
        /* gcur = d/d(pre-activation of Linear k output) */
        if (k < s->n_hidden) {
            const real *a = acts + aoff[k];
            for (int q = 0; q < no; ++q) {
                if (s->act == 0) gcur[q] *= ((real)1 - a[q] * a[q]);        /* tanh'         */
                else gcur[q] = (a[q] > 0) ? gcur[q] : (real)0;             /* relu'         */
            }
        }
        for (int i = 0; i < ni; ++i) gprev[i] = 0;
        for (int q = 0; q < no; ++q) {
            const real gq = gcur[q];
            gb[q] += gq;
            for (int i = 0; i < ni; ++i) {
                gW[(size_t)q * ni + i] += gq * inp[i];
                gprev[i] += gq * W[(size_t)q * ni + i];
            }
        }
        real *t = gcur; gcur = gprev; gprev = t;
    }
    for (int i = 0; i < s->d + s->c; ++i) gin[i] += gcur[i];
}

/* ---- loss = -log_prob(X, C); gradient of loss wrt every parameter ---------
 * Replaces `loss.backward()` (realnvp.py:246-250; autograd in the reference).
 * inv_B scales each row's contribution (1/B_global for data-parallel shards).
 * grad [P] is OVERWRITTEN.  loss_out = -(sum_r logp_r) * inv_B.
 * Derivation: gld = -inv_B for every layer;  gz = z * inv_B;
 *   gs = (1-m)*(gy*x*exp(s) + gld), gt = (1-m)*gy,
 *   gx = gy*(m + (1-m)*exp(s)) + m*g_in[:d].                                */
void rnvp_oracle_loss_grad(const rnvp_shape *s, const real *params, const uint8_t *masks,
                           const real *x, const real *c, int64_t B, double inv_B,
                           real *grad, real *loss_out) {
    const int d = s->d, nc = s->c, W = max_width(s), L = s->L;
    const size_t npn = net_param_count(s), npl = 2 * npn, P = npl * L;
    int hs = 0;
    for (int k = 0; k < s->n_hidden; ++k) hs += s->hidden[k];
    for (size_t i = 0; i < P; ++i) grad[i] = 0;
    /* per-row saved state: x at the input of every layer */
    real *xs = (real *)malloc(sizeof(real) * (size_t)(L + 1) * d);
    real *xc = (real *)malloc(sizeof(real) * (size_t)(d + nc) * 2);
    real *gin = xc + d + nc;
    real *T = (real *)malloc(sizeof(real) * (size_t)(6 * d + 2 * W + 2 * hs + 2));
    real *S = T + d, *gy = S + d, *gT = gy + d, *gS = gT + d, *gx = gS + d;
    real *t0 = gx + d, *t1 = t0 + W, *actT = t1 + W, *actS = actT + hs;
    const real half_log2pi_d = (real)(0.5 * (double)d * 1.8378770664093453);
    const real gld = (real)(-inv_B);
    double loss = 0;
    for (int64_t r = 0; r < B; ++r) {
        /* forward, keeping layer inputs */
        for (int j = 0; j < d; ++j) xs[j] = x[r * d + j];
        real ldsum = 0;
        for (int l = 0; l < L; ++l) {
            const uint8_t *m = masks + (size_t)l * d;
            const real *xl = xs + (size_t)l * d;
            real *xn = xs + (size_t)(l + 1) * d;
            for (int j = 0; j < d; ++j) xc[j] = xl[j] * (real)m[j];
            for (int j = 0; j < nc; ++j) xc[d + j] = c[r * nc + j];
            net_forward_row(s, params + l * npl, xc, T, NULL, t0, t1);
            net_forward_row(s, params + l * npl + npn, xc, S, NULL, t0, t1);
            for (int j = 0; j < d; ++j) {
                real mj = (real)m[j];
                xn[j] = (xl[j] * exp_real(S[j]) + T[j]) * ((real)1 - mj) + xl[j] * mj;
                ldsum += S[j] * ((real)1 - mj);
            }
        }
        const real *z = xs + (size_t)L * d;
        real ss = 0;
        for (int j = 0; j < d; ++j) ss += z[j] * z[j];
        loss -= (double)(ldsum + (-(real)0.5 * ss - half_log2pi_d));
        /* backward */
        for (int j = 0; j < d; ++j) gy[j] = z[j] * (real)inv_B;             /* d(-prior)/dz  */
        for (int l = L - 1; l >= 0; --l) {
            const uint8_t *m = masks + (size_t)l * d;
            const real *xl = xs + (size_t)l * d;
            for (int j = 0; j < d; ++j) xc[j] = xl[j] * (real)m[j];
            for (int j = 0; j < nc; ++j) xc[d + j] = c[r * nc + j];
            net_forward_row(s, params + l * npl, xc, T, actT, t0, t1);
            net_forward_row(s, params + l * npl + npn, xc, S, actS, t0, t1);
            for (int j = 0; j < d; ++j) {
                real mj = (real)m[j], es = exp_real(S[j]);
                gT[j] = ((real)1 - mj) * gy[j];
                gS[j] = ((real)1 - mj) * (gy[j] * xl[j] * es + gld);
                gx[j] = gy[j] * (mj + ((real)1 - mj) * es);
            }
            for (int i = 0; i < d + nc; ++i) gin[i] = 0;
            net_backward_row(s, params + l * npl, grad + l * npl, xc, actT, gT, gin, t0, t1);
            net_backward_row(s, params + l * npl + npn, grad + l * npl + npn, xc, actS, gS, gin, t0, t1);
            for (int j = 0; j < d; ++j) gy[j] = gx[j] + (real)m[j] * gin[j];
        }
    }
    if (loss_out) *loss_out = (real)(loss * inv_B);
    free(T); free(xc); free(xs);
}

/* ---- torch.optim.Adam, one step over a flat buffer ------------------------
 * realnvp.py:205-207 (defaults betas (0.9,0.999), eps 1e-8, amsgrad False, L2
 * weight decay folded into the gradient).  `step` is the 1-based step count.
 * Scalar bookkeeping in double, tensor math in `real`, as torch does.        */
void rnvp_oracle_adam(real *p, const real *g_in, real *m, real *v, int64_t P,
                      double lr, double beta1, double beta2, double eps,
                      double weight_decay, int64_t step) {
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    const real step_size = (real)(lr / bc1);
    const real bc2_sqrt = (real)sqrt(bc2);
    const real w1 = (real)(1.0 - beta1), b2 = (real)beta2, w2 = (real)(1.0 - beta2);
    const real wd = (real)weight_decay, e = (real)eps;
    for (int64_t i = 0; i < P; ++i) {
        real g = g_in[i];
        if (weight_decay != 0) g = g + wd * p[i];                 /* grad.add(param, alpha=wd)  */
        m[i] = m[i] + w1 * (g - m[i]);                            /* exp_avg.lerp_(grad, 1-b1)  */
        v[i] = v[i] * b2 + w2 * (g * g);                          /* mul_(b2).addcmul_(g,g,1-b2)*/
#if defined(ORACLE_IS_DOUBLE)
        real denom = sqrt(v[i]) / bc2_sqrt + e;
#else
        real denom = sqrtf(v[i]) / bc2_sqrt + e;
#endif
        p[i] = p[i] - step_size * (m[i] / denom);                 /* addcdiv_(m, denom, -ss)    */
    }
}

/* ---- alternating masks (realnvp.py:199): mask[l][j] = (j + l) % 2 --------- */
void rnvp_oracle_default_masks(int L, int d, uint8_t *masks) {
    for (int l = 0; l < L; ++l)
        for (int j = 0; j < d; ++j) masks[(size_t)l * d + j] = (uint8_t)((j + l) % 2);
}

int rnvp_oracle_real_bytes(void) { return (int)sizeof(real); }

/* ---- the build's counter-based 'device' prior (no counterpart in the reference, whose prior draw is
 * torch's CPU generator: nflow.py:141) --------------------------------------------------------------------
 * Restates include/rnvp_hip.h `rnvp_prior_normal`: Philox4x32-10 (Salmon et al., "Parallel random numbers:
 * as easy as 1, 2, 3", SC'11; constants of the Random123 library) keyed by the seed, counter
 * (row_lo, row_hi, j / 4, 0); words (0,1) and (2,3) of the output give two Box-Muller pairs = the normals of
 * features 4*(j/4) .. +3.  Pinned by the published known-answer vectors (tests/test_oracle_golden.py).
 * Box-Muller is evaluated in double and rounded, so the HIP kernels (float logf / sincospif) agree to ~1e-7. */
void rnvp_oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static void box_muller_ref(uint32_t a, uint32_t b, real *z0, real *z1) {
    const float u1 = (float)(a >> 9) * 1.1920928955078125e-07f + 5.9604644775390625e-08f;
    const float f = (float)(b >> 8) * 5.9604644775390625e-08f;
    const double rad = sqrt(-2.0 * log((double)u1)), ang = 6.283185307179586476925286766559 * (double)f;
    *z0 = (real)(rad * cos(ang));
    *z1 = (real)(rad * sin(ang));
}

void rnvp_oracle_prior_normal(uint64_t seed, int64_t row0, int64_t n, int32_t d, real *z) {
    const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    for (int64_t r = 0; r < n; ++r) {
        const uint64_t row = (uint64_t)(row0 + r);
        for (int blk = 0; 4 * blk < d; ++blk) {
            const uint32_t ctr[4] = {(uint32_t)row, (uint32_t)(row >> 32), (uint32_t)blk, 0u};
            uint32_t w[4];
            real v[4];
            rnvp_oracle_philox4x32_10(ctr, key, w);
            box_muller_ref(w[0], w[1], &v[0], &v[1]);
            box_muller_ref(w[2], w[3], &v[2], &v[3]);
            for (int e = 0; e < 4 && 4 * blk + e < d; ++e) z[r * d + 4 * blk + e] = v[e];
        }
    }
}
