/*
 * cvae_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the conditional VAE of hse-cs/probaforms @ 2024_10_08
 * (/root/reference/probaforms/models/cvae.py), the "next" row of SURVEY.md 8(f) rank 1.
 * Pinned against fixtures produced by the reference itself (tests/golden/make_golden_cvae.py,
 * tests/test_cvae_oracle_golden.py).  Same access rules as rnvp_oracle.c.
 *
 * Parameter layout (flat, "oracle order"; the Python side maps the reference's state_dict onto it):
 *   encoder trunk: for each hidden Linear k: weight [h_k, in_k], bias [h_k]      (cvae.py:18-33)
 *   encoder heads: W_mu [lat, h], W_ls [lat, h], b_mu [lat], b_ls [lat]          (cvae.py:35-36)
 *   decoder      : for each Linear: weight, bias; last one maps to d outputs     (cvae.py:72-89)
 * i.e. the two heads are stored as ONE Linear with 2*lat outputs.
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

#define MAXH 8
typedef struct {
    int32_t d, c, lat, n_hidden;
    int32_t hidden[MAXH];
    int32_t act;              /* 0 tanh, 1 relu (cvae.py:26-32) */
} cvae_shape;

static real act_f(real v, int act) {
    if (act == 0) {
#ifdef ORACLE_IS_DOUBLE
        return tanh(v);
#else
        return tanhf(v);
#endif
    }
    return v > 0 ? v : (real)0;
}
static real exp_r(real v) {
#ifdef ORACLE_IS_DOUBLE
    return exp(v);
#else
    return expf(v);
#endif
}

/* an MLP: n_hidden activated Linears followed by one plain Linear of `nout` outputs */
static size_t mlp_params(const cvae_shape *s, int nin, int nout) {
    size_t n = 0;
    for (int k = 0; k < s->n_hidden; ++k) { n += (size_t)s->hidden[k] * nin + s->hidden[k]; nin = s->hidden[k]; }
    return n + (size_t)nout * nin + nout;
}
size_t cvae_oracle_param_count(const cvae_shape *s) {
    return mlp_params(s, s->d + s->c, 2 * s->lat) + mlp_params(s, s->lat + s->c, s->d);
}
static int hsum(const cvae_shape *s) { int h = 0; for (int k = 0; k < s->n_hidden; ++k) h += s->hidden[k]; return h; }
static int wmax(const cvae_shape *s) {
    int w = s->d + s->c; if (2 * s->lat > w) w = 2 * s->lat; if (s->lat + s->c > w) w = s->lat + s->c; if (s->d > w) w = s->d;
    for (int k = 0; k < s->n_hidden; ++k) if (s->hidden[k] > w) w = s->hidden[k];
    return w;
}

/* forward of one MLP on one row; acts (sum(hidden) reals) keeps every hidden activation */
static void mlp_fwd(const cvae_shape *s, const real *p, int nin, int nout, const real *in, real *acts, real *out) {
    const real *cur = in;
    for (int k = 0; k <= s->n_hidden; ++k) {
        const int no = (k < s->n_hidden) ? s->hidden[k] : nout;
        const real *W = p, *b = p + (size_t)no * nin;
        real *dst = (k < s->n_hidden) ? acts : out;
        for (int o = 0; o < no; ++o) {
            real a = 0;
            for (int i = 0; i < nin; ++i) a += cur[i] * W[(size_t)o * nin + i];
            a += b[o];
            dst[o] = (k < s->n_hidden) ? act_f(a, s->act) : a;
        }
        p += (size_t)no * nin + no;
        cur = dst;
        if (k < s->n_hidden) acts += no;
        nin = no;
    }
}

/* backward of one MLP on one row: accumulates parameter gradients into g, writes d/d(in) to gin */
static void mlp_bwd(const cvae_shape *s, const real *p, real *g, int nin0, int nout, const real *in, const real *acts,
                    const real *gout, real *gin, real *t0, real *t1) {
    size_t off[MAXH + 1]; int nins[MAXH + 1], nouts[MAXH + 1]; size_t aoff[MAXH + 1];
    size_t o = 0, ao = 0; int nin = nin0;
    for (int k = 0; k <= s->n_hidden; ++k) {
        int no = (k < s->n_hidden) ? s->hidden[k] : nout;
        off[k] = o; nins[k] = nin; nouts[k] = no; aoff[k] = ao;
        o += (size_t)no * nin + no; if (k < s->n_hidden) ao += no; nin = no;
    }
    real *gc = t0, *gp = t1;
    for (int j = 0; j < nout; ++j) gc[j] = gout[j];
    for (int k = s->n_hidden; k >= 0; --k) {
        const int ni = nins[k], no = nouts[k];
        const real *W = p + off[k]; real *gW = g + off[k], *gb = gW + (size_t)no * ni;
        const real *inp = (k == 0) ? in : acts + aoff[k - 1];
        if (k < s->n_hidden) {
            const real *a = acts + aoff[k];
            for (int q = 0; q < no; ++q) gc[q] = (s->act == 0) ? gc[q] * ((real)1 - a[q] * a[q]) : (a[q] > 0 ? gc[q] : (real)0);
        }
        for (int i = 0; i < ni; ++i) gp[i] = 0;
        for (int q = 0; q < no; ++q) {
            gb[q] += gc[q];
            for (int i = 0; i < ni; ++i) { gW[(size_t)q * ni + i] += gc[q] * inp[i]; gp[i] += gc[q] * W[(size_t)q * ni + i]; }
        }
        real *t = gc; gc = gp; gp = t;
    }
    for (int i = 0; i < nin0; ++i) gin[i] = gc[i];
}

/* CVAE.compute_loss (cvae.py:186-203) with eps given, and its gradient (autograd in the reference,
 * cvae.py:243-246).  loss = KL_weight * mean_b(-0.5 sum_j(1 + ls - mu^2 - e^ls)) + mean_{b,j}((x - x_rec)^2).
 * inv_B = 1/B_global.  grad may be NULL (loss only, cvae.py:254-259).                                 */
void cvae_oracle_loss_grad(const cvae_shape *s, const real *params, const real *x, const real *c, const real *eps,
                           int64_t B, double inv_B, double kl_weight, real *grad, real *loss_out) {
    const int d = s->d, nc = s->c, lat = s->lat, hs = hsum(s), W = wmax(s);
    const size_t pe = mlp_params(s, d + nc, 2 * lat), P = cvae_oracle_param_count(s);
    if (grad) for (size_t i = 0; i < P; ++i) grad[i] = 0;
    real *buf = (real *)malloc(sizeof(real) * (size_t)(3 * (d + nc) + 2 * lat + 2 * hs + 4 * lat + 3 * d + 3 * W + 8));
    real *ein = buf, *ea = ein + d + nc, *eo = ea + hs, *din = eo + 2 * lat, *da = din + lat + nc, *xr = da + hs;
    real *gx = xr + d, *gz = gx + d, *ge = gz + lat + nc, *t0 = ge + 2 * lat, *t1 = t0 + W, *gdump = t1 + W;
    double kl = 0, se = 0;
    for (int64_t r = 0; r < B; ++r) {
        for (int j = 0; j < d; ++j) ein[j] = x[r * d + j];                                   /* cvae.py:58 cat(X, C) */
        for (int j = 0; j < nc; ++j) ein[d + j] = c[r * nc + j];
        mlp_fwd(s, params, d + nc, 2 * lat, ein, ea, eo);                                     /* mu | log_sigma       */
        for (int j = 0; j < lat; ++j) din[j] = eo[j] + exp_r(eo[lat + j] / 2) * eps[r * lat + j];   /* cvae.py:188     */
        for (int j = 0; j < nc; ++j) din[lat + j] = c[r * nc + j];
        mlp_fwd(s, params + pe, lat + nc, d, din, da, xr);                                    /* decoder              */
        real k1 = 0, s1 = 0;
        for (int j = 0; j < lat; ++j) k1 += (real)1 + eo[lat + j] - eo[j] * eo[j] - exp_r(eo[lat + j]);
        for (int j = 0; j < d; ++j) { real df = x[r * d + j] - xr[j]; s1 += df * df; }
        kl += (double)((real)-0.5 * k1); se += (double)s1;
        if (!grad) continue;
        for (int j = 0; j < d; ++j) gx[j] = (real)(2.0 * inv_B / d) * (xr[j] - x[r * d + j]);
        mlp_bwd(s, params + pe, grad + pe, lat + nc, d, din, da, gx, gz, t0, t1);
        for (int j = 0; j < lat; ++j) {
            const real mu = eo[j], ls = eo[lat + j], e2 = exp_r(ls / 2);
            ge[j] = gz[j] + (real)(kl_weight * inv_B) * mu;
            ge[lat + j] = gz[j] * eps[r * lat + j] * (real)0.5 * e2 + (real)(kl_weight * inv_B) * (real)-0.5 * ((real)1 - exp_r(ls));
        }
        mlp_bwd(s, params, grad, d + nc, 2 * lat, ein, ea, ge, gdump, t0, t1);      /* d/d(x,c) is not needed */
    }
    if (loss_out) *loss_out = (real)(kl_weight * kl * inv_B + se * inv_B / d);
    free(buf);
}

/* Decoder.forward (cvae.py:92-113): x = decoder([z || c]) */
void cvae_oracle_decode(const cvae_shape *s, const real *params, const real *z, const real *c, int64_t n, real *x_out) {
    const int d = s->d, nc = s->c, lat = s->lat, hs = hsum(s);
    const size_t pe = mlp_params(s, d + nc, 2 * lat);
    real *din = (real *)malloc(sizeof(real) * (size_t)(lat + nc + hs));
    for (int64_t r = 0; r < n; ++r) {
        for (int j = 0; j < lat; ++j) din[j] = z[r * lat + j];
        for (int j = 0; j < nc; ++j) din[lat + j] = c[r * nc + j];
        mlp_fwd(s, params + pe, lat + nc, d, din, din + lat + nc, x_out + r * d);
    }
    free(din);
}

/* Encoder.forward (cvae.py:39-64): mu, log_sigma */
void cvae_oracle_encode(const cvae_shape *s, const real *params, const real *x, const real *c, int64_t n, real *mu, real *ls) {
    const int d = s->d, nc = s->c, lat = s->lat, hs = hsum(s);
    real *ein = (real *)malloc(sizeof(real) * (size_t)(d + nc + hs + 2 * lat));
    for (int64_t r = 0; r < n; ++r) {
        for (int j = 0; j < d; ++j) ein[j] = x[r * d + j];
        for (int j = 0; j < nc; ++j) ein[d + j] = c[r * nc + j];
        real *eo = ein + d + nc + hs;
        mlp_fwd(s, params, d + nc, 2 * lat, ein, ein + d + nc, eo);
        for (int j = 0; j < lat; ++j) { mu[r * lat + j] = eo[j]; ls[r * lat + j] = eo[lat + j]; }
    }
    free(ein);
}
