/*
 * oracle/nfa_oracle.c -- TEST INFRASTRUCTURE ONLY.  NOT PRODUCT CODE.
 *
 * Plain-C, scalar, single-threaded restatement of the arithmetic on the nflows coupling-layer
 * hot path.  It exists so that the HIP kernels in nflows_amd/csrc can be checked on a box where
 * /root/reference is absent.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library; nflows_amd itself never does.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function here against
 * vectors produced by the real reference (imported from /root/reference in the build container
 * by tests/golden/make_golden.py, outputs committed under tests/golden/).
 *
 * Each function cites the reference lines it restates (paths relative to /root/reference).
 * The file is compiled twice (REAL=float / REAL=double); the float build rounds after every
 * arithmetic step exactly where aten's fp32 CPU kernels round (build with -ffp-contract=off),
 * transcendental functions are evaluated in double and rounded once ("correctly rounded"),
 * which is within 1 ulp of aten's SLEEF results.  The double build is the fp64 ground truth.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

#ifndef REAL
#define REAL float
#endif
#ifndef SUF
#define SUF _f32
#endif
#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUF)

/* status bits, identical to include/nflows_amd.h */
#define ORACLE_STATUS_OUTSIDE_DOMAIN 1
#define ORACLE_STATUS_NEG_DISCRIMINANT 2
#define ORACLE_STATUS_BAD_INDEX 4

typedef struct {
    int32_t num_bins;
    int32_t tails; /* 0: constrained spline (K+1 derivative logits); 1: linear tails (K-1) */
    double left, right, bottom, top;
    double min_bin_width, min_bin_height, min_derivative;
    double softplus_beta; /* 1, or ln2/(1-min_derivative) when enable_identity_init */
    double tail_logit;    /* log(exp(1-min_derivative)-1), rational_quadratic.py:34 */
    double wh_divisor;    /* sqrt(hidden_features) (coupling.py:554-556) or 0 = no scaling */
} oracle_rqs_spec;

static REAL r_exp(REAL x) { return (REAL)exp((double)x); }
static REAL r_log(REAL x) { return (REAL)log((double)x); }
static REAL r_log1p(REAL x) { return (REAL)log1p((double)x); }
static REAL r_sqrt(REAL x) { return (REAL)sqrt((double)x); }

/* torch.nn.functional.softplus(x, beta, threshold=20) as used at rational_quadratic.py:104,
 * coupling.py:225, autoregressive.py:101:  x if x*beta > 20 else log1p(exp(x*beta))/beta */
static REAL softplus(REAL x, REAL beta) {
    REAL xb = x * beta;
    if (xb > (REAL)20) return x;
    return r_log1p(r_exp(xb)) / beta;
}

/* F.softmax(u, -1) -> min + (1-min*K)*p -> cumsum -> pad 0 -> affine -> forced end knots.
 * rational_quadratic.py:91-97 (widths) and :106-112 (heights).  knots has K+1 entries.
 * aten's CPU cumsum of fp32 accumulates in double and rounds every prefix. */
static void knots_from_logits(const REAL *u, int K, REAL divisor, REAL lo, REAL hi, REAL minbin,
                              REAL *knots, REAL *scratch) {
    REAL m = -INFINITY;
    for (int i = 0; i < K; ++i) {
        scratch[i] = (divisor != (REAL)0) ? u[i] / divisor : u[i];
        if (scratch[i] > m) m = scratch[i];
    }
    double s = 0.0;
    for (int i = 0; i < K; ++i) {
        scratch[i] = r_exp(scratch[i] - m);
        s += (double)scratch[i];
    }
    REAL sum = (REAL)s;
    REAL one_minus = (REAL)(1.0 - (double)minbin * K); /* python-float arithmetic, then cast */
    double acc = 0.0;
    REAL span = (REAL)((double)hi - (double)lo);
    knots[0] = lo;
    for (int i = 0; i < K; ++i) {
        REAL p = scratch[i] / sum;
        REAL w = minbin + one_minus * p;
        acc += (double)w;
        REAL c = (REAL)acc;
        knots[i + 1] = span * c + lo;
    }
    knots[K] = hi;
}

/* rational_quadratic_spline for ONE element, rational_quadratic.py:66-181.
 * ud_full points at K+1 derivative logits (already padded for linear tails).
 * Returns status bits. */
static int rqs_one(REAL x, const REAL *uw, const REAL *uh, const REAL *ud_full,
                   const oracle_rqs_spec *sp, int inverse, REAL *y, REAL *lad, int32_t *bin_out) {
    enum { KMAX = 256 };
    int K = sp->num_bins;
    REAL cw[KMAX + 1], ch[KMAX + 1], tmp[KMAX];
    REAL left = (REAL)sp->left, right = (REAL)sp->right;
    REAL bottom = (REAL)sp->bottom, top = (REAL)sp->top;
    REAL div = (REAL)sp->wh_divisor;
    int status = 0;

    knots_from_logits(uw, K, div, left, right, (REAL)sp->min_bin_width, cw, tmp);
    knots_from_logits(uh, K, div, bottom, top, (REAL)sp->min_bin_height, ch, tmp);

    /* torchutils.searchsorted, torchutils.py:134-136: last knot += 1e-6, count of >=, minus 1 */
    const REAL *loc = inverse ? ch : cw;
    int cnt = 0;
    for (int j = 0; j <= K; ++j) {
        REAL kn = loc[j];
        if (j == K) kn = kn + (REAL)1e-6;
        if (x >= kn) ++cnt;
    }
    int k = cnt - 1;
    if (bin_out) *bin_out = k;
    if (k < 0 || k >= K) { /* reference would raise InputOutsideDomain / index error */
        *y = x;
        *lad = 0;
        return ORACLE_STATUS_OUTSIDE_DOMAIN;
    }

    REAL in_cw = cw[k], in_w = cw[k + 1] - cw[k];       /* :98, :120-121 */
    REAL in_ch = ch[k], in_h = ch[k + 1] - ch[k];       /* :113, :123, :130 */
    REAL delta = in_h / in_w;                           /* :124 */
    REAL beta = (REAL)sp->softplus_beta;
    REAL mind = (REAL)sp->min_derivative;
    REAL d0 = mind + softplus(ud_full[k], beta);        /* :104, :127 */
    REAL d1 = mind + softplus(ud_full[k + 1], beta);    /* :128 */
    REAL s = (d0 + d1) - (REAL)2 * delta;

    if (inverse) { /* :132-160 */
        REAL yc = x - in_ch;
        REAL a = yc * s + in_h * (delta - d0);
        REAL b = in_h * d0 - yc * s;
        REAL c = (-delta) * yc;
        REAL disc = b * b - ((REAL)4 * a) * c;
        if (!(disc >= (REAL)0)) status |= ORACLE_STATUS_NEG_DISCRIMINANT;
        REAL root = ((REAL)2 * c) / ((-b) - r_sqrt(disc));
        *y = root * in_w + in_cw;
        REAL t1mt = root * ((REAL)1 - root);
        REAL den = delta + s * t1mt;
        REAL omr = (REAL)1 - root;
        REAL dnum = (delta * delta) *
                    ((d1 * (root * root) + ((REAL)2 * delta) * t1mt) + d0 * (omr * omr));
        *lad = -(r_log(dnum) - (REAL)2 * r_log(den));
    } else { /* :162-181 */
        REAL theta = (x - in_cw) / in_w;
        REAL t1mt = theta * ((REAL)1 - theta);
        REAL num = in_h * (delta * (theta * theta) + d0 * t1mt);
        REAL den = delta + s * t1mt;
        *y = in_ch + num / den;
        REAL omt = (REAL)1 - theta;
        REAL dnum = (delta * delta) *
                    ((d1 * (theta * theta) + ((REAL)2 * delta) * t1mt) + d0 * (omt * omt));
        *lad = r_log(dnum) - (REAL)2 * r_log(den);
    }
    return status;
}

/* One element of either functional:
 *   tails==1: unconstrained_rational_quadratic_spline, rational_quadratic.py:13-63
 *             (inclusive interval test :26, identity + zero logabsdet outside :38-39,
 *              derivative logits padded with tail_logit :33-36)
 *   tails==0: rational_quadratic_spline with a domain check (:81-82). */
static int rqs_elem(REAL x, const REAL *uw, const REAL *uh, const REAL *ud, int nd,
                    const oracle_rqs_spec *sp, int inverse, REAL *y, REAL *lad, int32_t *bin_out) {
    enum { KMAX = 256 };
    int K = sp->num_bins;
    if (sp->tails == 1) {
        REAL tb_lo = (REAL)sp->left, tb_hi = (REAL)sp->right;
        if (!(x >= tb_lo && x <= tb_hi)) {
            *y = x;
            *lad = 0;
            if (bin_out) *bin_out = -1;
            return 0;
        }
        /* F.pad(ud, (1, 1)) then [...,0] = [...,-1] = constant (:33-36); only padded entries
         * 0..K are ever gathered, so with nd > K-1 logits the right constant is never reached */
        REAL full[KMAX + 1];
        full[0] = (REAL)sp->tail_logit;
        for (int i = 1; i <= K; ++i) full[i] = (i - 1 < nd) ? ud[i - 1] : (REAL)sp->tail_logit;
        return rqs_one(x, uw, uh, full, sp, inverse, y, lad, bin_out);
    }
    REAL lo = inverse ? (REAL)sp->bottom : (REAL)sp->left;
    REAL hi = inverse ? (REAL)sp->top : (REAL)sp->right;
    (void)lo; (void)hi;
    /* the reference checks inputs against [left, right] in BOTH directions (:81) */
    if (x < (REAL)sp->left || x > (REAL)sp->right) {
        *y = x;
        *lad = 0;
        if (bin_out) *bin_out = -1;
        return ORACLE_STATUS_OUTSIDE_DOMAIN;
    }
    return rqs_one(x, uw, uh, ud, sp, inverse, y, lad, bin_out);
}

/* Elementwise functional over N elements; logits given as three row-strided arrays
 * (strides in elements).  bins may be NULL. */
int FN(oracle_rqs_elementwise)(const REAL *x, const REAL *uw, int64_t sw, const REAL *uh, int64_t sh,
                               const REAL *ud, int64_t sd, int nd, int64_t n,
                               const oracle_rqs_spec *sp, int inverse, REAL *y, REAL *lad,
                               int32_t *bins) {
    int status = 0;
    if (sp->num_bins < 1 || sp->num_bins > 256) return -1;
    for (int64_t i = 0; i < n; ++i)
        status |= rqs_elem(x[i], uw + i * sw, uh + i * sh, ud + i * sd, nd, sp, inverse, y + i,
                           lad + i, bins ? bins + i : NULL);
    return status;
}

/* The knots of one axis as rational_quadratic.py:91-98 (widths: axis 0) / :106-113 (heights: axis 1) builds them --
 * `cumwidths` / `cumheights`, K + 1 per element, before searchsorted nudges the last one.  For the tests that ask
 * where an input sits relative to the knot between two bins (tests/test_gpu_bin_index.py). */
void FN(oracle_rqs_knots)(const REAL *u, int64_t stride, int64_t n, const oracle_rqs_spec *sp, int axis,
                          REAL *knots) {
    enum { KMAX = 256 };
    int K = sp->num_bins;
    REAL tmp[KMAX];
    if (K < 1 || K > KMAX) return;
    for (int64_t i = 0; i < n; ++i)
        knots_from_logits(u + i * stride, K, (REAL)sp->wh_divisor, axis ? (REAL)sp->bottom : (REAL)sp->left,
                          axis ? (REAL)sp->top : (REAL)sp->right,
                          axis ? (REAL)sp->min_bin_height : (REAL)sp->min_bin_width, knots + i * (K + 1), tmp);
}

/* torch.sum(x, dim=1) (torchutils.sum_except_batch, torchutils.py:19-24); the oracle
 * accumulates in double and rounds once, so it is the best fp32 answer, not aten's order. */
void FN(oracle_rowsum)(const REAL *x, int64_t rows, int64_t cols, REAL *out) {
    for (int64_t r = 0; r < rows; ++r) {
        double s = 0.0;
        for (int64_t c = 0; c < cols; ++c) s += (double)x[r * cols + c];
        out[r] = (REAL)s;
    }
}

/* Permutation._permute on dim 1: out[:, c] = in[:, perm[c]]  (permutations.py:27-39). */
int FN(oracle_permute_cols)(const REAL *x, const int64_t *perm, int64_t rows, int64_t cols,
                            REAL *out) {
    for (int64_t c = 0; c < cols; ++c)
        if (perm[c] < 0 || perm[c] >= cols) return ORACLE_STATUS_BAD_INDEX;
    for (int64_t r = 0; r < rows; ++r)
        for (int64_t c = 0; c < cols; ++c) out[r * cols + c] = x[r * cols + perm[c]];
    return 0;
}

/* PiecewiseRationalQuadraticCouplingTransform forward/inverse given the conditioner output:
 * CouplingTransform.forward/inverse split+scatter (coupling.py:82-83, :96-98, :111-112,
 * :126-128), params reshape [B, d_t*P] -> [B, d_t, P] (:289), per-feature slices (:550-552),
 * /sqrt(hidden) on width/height logits (:554-556), functional, row-sum (:293).
 * in_perm (may be NULL) applies a preceding Permutation: x'[:, c] = x[:, in_perm[c]]
 * (permutations.py:37); out_scatter (may be NULL) a following Permutation.inverse:
 * out[:, out_scatter[c]] = y[:, c]  (permutations.py:22-24, :44-45). */
int FN(oracle_rqs_coupling)(const REAL *x, const REAL *params, const int64_t *transform_idx,
                            const int64_t *in_perm, const int64_t *out_scatter, int64_t batch, int64_t features,
                            int64_t num_transform, const oracle_rqs_spec *sp, int inverse,
                            REAL *out, REAL *logabsdet) {
    int K = sp->num_bins;
    if (K < 1 || K > 256) return -1;
    int64_t P = sp->tails == 1 ? 3 * K - 1 : 3 * K + 1;
    int status = 0;
    for (int64_t b = 0; b < batch; ++b) {
        const REAL *xr = x + b * features;
        REAL *orow = out + b * features;
        for (int64_t c = 0; c < features; ++c)
            orow[out_scatter ? out_scatter[c] : c] = in_perm ? xr[in_perm[c]] : xr[c];
        double acc = 0.0;
        for (int64_t j = 0; j < num_transform; ++j) {
            int64_t col = transform_idx[j];
            if (col < 0 || col >= features) return ORACLE_STATUS_BAD_INDEX;
            const REAL *p = params + (b * num_transform + j) * P;
            REAL xin = in_perm ? xr[in_perm[col]] : xr[col];
            REAL y, lad;
            status |= rqs_elem(xin, p, p + K, p + 2 * K, (int)(P - 2 * K), sp, inverse, &y, &lad, NULL);
            orow[out_scatter ? out_scatter[col] : col] = y;
            acc += (double)lad;
        }
        logabsdet[b] = (REAL)acc;
    }
    return status;
}

/* AffineCouplingTransform / AdditiveCouplingTransform given the conditioner output.
 * coupling.py:234-252 (shift = first d_t columns, scale logits = last d_t), activations
 * :224-225, additive :263-269.
 *   activation 0: sigmoid(u+2)+1e-3   1: clamp(softplus(u)+1e-3, 0, 3)
 *   activation 2: additive (params has d_t columns; scale == 1, logabsdet == 0 exactly)
 *   activation 3: `scale` holds a ready-made [B, d_t] scale tensor, params' first d_t cols shift */
int FN(oracle_affine_coupling)(const REAL *x, const REAL *params, const REAL *scale_in,
                               const int64_t *transform_idx, const int64_t *in_perm,
                               const int64_t *out_scatter, int64_t batch, int64_t features,
                               int64_t num_transform, int activation, int inverse,
                               REAL *out, REAL *logabsdet) {
    int64_t pcols = (activation == 2) ? num_transform : 2 * num_transform;
    for (int64_t b = 0; b < batch; ++b) {
        const REAL *xr = x + b * features;
        REAL *orow = out + b * features;
        for (int64_t c = 0; c < features; ++c)
            orow[out_scatter ? out_scatter[c] : c] = in_perm ? xr[in_perm[c]] : xr[c];
        double acc = 0.0;
        for (int64_t j = 0; j < num_transform; ++j) {
            int64_t col = transform_idx[j];
            if (col < 0 || col >= features) return ORACLE_STATUS_BAD_INDEX;
            REAL shift = params[b * pcols + j];
            REAL scale;
            if (activation == 0) {
                REAL u = params[b * pcols + num_transform + j] + (REAL)2;
                scale = (REAL)1 / ((REAL)1 + r_exp(-u)) + (REAL)1e-3;
            } else if (activation == 1) {
                scale = softplus(params[b * pcols + num_transform + j], (REAL)1) + (REAL)1e-3;
                if (scale < (REAL)0) scale = 0;
                if (scale > (REAL)3) scale = 3;
            } else if (activation == 2) {
                scale = 1;
            } else if (activation == 4) { /* autoregressive.py:101 */
                scale = softplus(params[b * pcols + num_transform + j], (REAL)1) + (REAL)1e-3;
            } else {
                scale = scale_in[b * num_transform + j];
            }
            REAL xin = in_perm ? xr[in_perm[col]] : xr[col];
            REAL ls = r_log(scale);
            if (inverse) {
                orow[out_scatter ? out_scatter[col] : col] = (xin - shift) / scale;
                acc -= (double)ls;
            } else {
                orow[out_scatter ? out_scatter[col] : col] = xin * scale + shift;
                acc += (double)ls;
            }
        }
        logabsdet[b] = (REAL)acc;
    }
    return 0;
}

/* StandardNormal._log_prob: -0.5*sum(x^2) - log_z   (distributions/normal.py:23-33). */
void FN(oracle_standard_normal_log_prob)(const REAL *x, int64_t rows, int64_t cols, REAL *out) {
    double log_z = 0.5 * (double)cols * log(2.0 * 3.14159265358979323846);
    for (int64_t r = 0; r < rows; ++r) {
        double s = 0.0;
        for (int64_t c = 0; c < cols; ++c) {
            REAL v = x[r * cols + c];
            s += (double)(REAL)(v * v);
        }
        REAL neg_energy = (REAL)-0.5 * (REAL)s;
        out[r] = neg_energy - (REAL)log_z; /* 0-dim fp64 buffer is cast to the tensor dtype */
    }
}

/* torchutils.searchsorted (utils/torchutils.py:134-136) on explicit knots: the last knot is
 * nudged by eps, index = (#knots <= x) - 1.  Pins the reference's known-answer test
 * tests/utils/torchutils_test.py:80-90. */
void FN(oracle_searchsorted)(const REAL *knots, int64_t num_knots, const REAL *x, int64_t n,
                             int64_t *idx) {
    for (int64_t i = 0; i < n; ++i) {
        int64_t cnt = 0;
        for (int64_t j = 0; j < num_knots; ++j) {
            REAL kn = knots[j];
            if (j == num_knots - 1) kn = kn + (REAL)1e-6;
            if (x[i] >= kn) ++cnt;
        }
        idx[i] = cnt - 1;
    }
}

/* MaskedAffineAutoregressiveTransform._elementwise_forward/_inverse (autoregressive.py:96-128):
 * params [B, D, 2] interleaved, [...,0] = scale logit, [...,1] = shift. */
void FN(oracle_affine_autoregressive)(const REAL *x, const REAL *params, int64_t batch, int64_t features,
                                      int inverse, REAL *out, REAL *logabsdet) {
    for (int64_t b = 0; b < batch; ++b) {
        double acc = 0.0;
        for (int64_t c = 0; c < features; ++c) {
            REAL scale = softplus(params[(b * features + c) * 2], (REAL)1) + (REAL)1e-3;
            REAL shift = params[(b * features + c) * 2 + 1];
            REAL ls = r_log(scale);
            REAL xv = x[b * features + c];
            if (inverse) {
                out[b * features + c] = (xv - shift) / scale;
                acc -= (double)ls;
            } else {
                out[b * features + c] = scale * xv + shift;
                acc += (double)ls;
            }
        }
        logabsdet[b] = (REAL)acc;
    }
}

/* ------------------------------------------------------------------------------------------
 * Sibling piecewise-polynomial splines.
 * The spec's box, tails and minimum sizes are read; derivative fields are unused. */

/* F.softmax over K logits (optionally divided by `divisor` first), fp32 sum accumulated in double
 * like knots_from_logits above */
static void softmax_k(const REAL *u, int K, REAL divisor, REAL *p) {
    REAL m = -INFINITY;
    for (int i = 0; i < K; ++i) {
        p[i] = (divisor != (REAL)0) ? u[i] / divisor : u[i];
        if (p[i] > m) m = p[i];
    }
    double s = 0.0;
    for (int i = 0; i < K; ++i) {
        p[i] = r_exp(p[i] - m);
        s += (double)p[i];
    }
    REAL sum = (REAL)s;
    for (int i = 0; i < K; ++i) p[i] = p[i] / sum;
}

/* torchutils.searchsorted (torchutils.py:134-136) over knots[0..K], last knot + 1e-6 */
static int search_knots(const REAL *knots, int K, REAL x) {
    int cnt = 0;
    for (int j = 0; j <= K; ++j) {
        REAL kn = knots[j];
        if (j == K) kn = kn + (REAL)1e-6;
        if (x >= kn) ++cnt;
    }
    return cnt - 1;
}

static REAL clamp01(REAL v) { /* torch.clamp(v, 0, 1): NaN stays NaN */
    if (v < (REAL)0) return (REAL)0;
    if (v > (REAL)1) return (REAL)1;
    return v;
}

/* linear_spline for ONE element: splines/linear.py:40-105 */
static int linear_one(REAL x, const REAL *updf, const oracle_rqs_spec *sp, int inverse, REAL *y, REAL *lad) {
    enum { KMAX = 256 };
    int K = sp->num_bins;
    REAL pdf[KMAX], cdf[KMAX + 1];
    REAL left = (REAL)sp->left, right = (REAL)sp->right, bottom = (REAL)sp->bottom;
    if (x < left || x > right) { /* :47-48 (compares against left/right in both directions) */
        *y = x;
        *lad = 0;
        return ORACLE_STATUS_OUTSIDE_DOMAIN;
    }
    REAL u = inverse ? (x - bottom) / (REAL)(sp->top - sp->bottom) : (x - left) / (REAL)(sp->right - sp->left);
    softmax_k(updf, K, (REAL)0, pdf);
    double acc = 0.0;
    cdf[0] = 0;
    for (int i = 0; i < K; ++i) {
        acc += (double)pdf[i];
        cdf[i + 1] = (REAL)acc;
    }
    cdf[K] = (REAL)1;
    REAL out;
    if (inverse) {
        int k = search_knots(cdf, K, u);
        if (k < 0 || k >= K) {
            *y = x;
            *lad = 0;
            return ORACLE_STATUS_OUTSIDE_DOMAIN;
        }
        /* torch.linspace(0, 1, K+1) (linear.py:67-71) is a float32 tensor whatever the input
         * dtype: start + i*step for the first half, end - (K-i)*step for the second, in float */
        float step = 1.0f / (float)K;
        float fb0 = (k < (K + 1) / 2) ? (float)k * step : 1.0f - (float)(K - k) * step;
        float fb1 = (k + 1 < (K + 1) / 2) ? (float)(k + 1) * step : 1.0f - (float)(K - k - 1) * step;
        REAL b0 = (REAL)fb0, b1 = (REAL)fb1;
        /* torchutils.searchsorted adds its eps to the last knot IN PLACE (torchutils.py:135), so the
         * slope / offset of the last bin (linear.py:73-76) see cdf[K] = 1 + 1e-6 */
        cdf[K] = cdf[K] + (REAL)1e-6;
        REAL slope = (cdf[k + 1] - cdf[k]) / (b1 - b0);
        REAL offset = cdf[k + 1] - slope * b1;
        out = clamp01((u - offset) / slope);
        *lad = -r_log(slope);
    } else {
        REAL pos = u * (REAL)K;
        int k = (int)floor((double)pos);
        if (k >= K) k = K - 1;
        REAL alpha = pos - (REAL)k;
        out = cdf[k];
        out = out + alpha * pdf[k];
        out = clamp01(out);
        *lad = r_log(pdf[k]) - (REAL)log(1.0 / K);
    }
    if (inverse) *y = out * (REAL)(sp->right - sp->left) + left;
    else *y = out * (REAL)(sp->top - sp->bottom) + bottom;
    return 0;
}

/* quadratic_spline for ONE element: splines/quadratic.py:55-159.  nh = K-1 (boundary heights
 * derived, :93-107) or K+1 */
static int quadratic_one(REAL x, const REAL *uw, const REAL *uh, int nh, const oracle_rqs_spec *sp,
                         int inverse, REAL *y, REAL *lad) {
    enum { KMAX = 256 };
    int K = sp->num_bins;
    REAL w[KMAX], he[KMAX + 1], h[KMAX + 1], lcdf[KMAX + 1], loc[KMAX + 1];
    REAL left = (REAL)sp->left, right = (REAL)sp->right, bottom = (REAL)sp->bottom;
    REAL minw = (REAL)sp->min_bin_width, minh = (REAL)sp->min_bin_height;
    if (x < left || x > right) {
        *y = x;
        *lad = 0;
        return ORACLE_STATUS_OUTSIDE_DOMAIN;
    }
    REAL u = inverse ? (x - bottom) / (REAL)(sp->top - sp->bottom) : (x - left) / (REAL)(sp->right - sp->left);
    softmax_k(uw, K, (REAL)sp->wh_divisor, w);
    REAL one_minus = (REAL)(1.0 - sp->min_bin_width * K);
    for (int i = 0; i < K; ++i) w[i] = minw + one_minus * w[i];
    REAL div = (REAL)sp->wh_divisor;
    if (nh == K - 1) {
        for (int i = 0; i < nh; ++i) {
            REAL v = (div != (REAL)0) ? uh[i] / div : uh[i];
            he[i + 1] = softplus(v, (REAL)1) + (REAL)1e-3;
        }
        REAL fw = (REAL)0.5 * w[0], lw = (REAL)0.5 * w[K - 1];
        REAL s = 0; /* torch.sum over K-2 terms, fp32 */
        for (int i = 0; i + 1 < nh; ++i) s += ((he[i + 1] + he[i + 2]) / (REAL)2) * w[i + 1];
        REAL num = (REAL)0.5 * fw * he[1] + (REAL)0.5 * lw * he[nh] + s;
        REAL c = num / ((REAL)1 - (REAL)0.5 * fw - (REAL)0.5 * lw);
        he[0] = c;
        he[K] = c;
    } else {
        for (int i = 0; i <= K; ++i) {
            REAL v = (div != (REAL)0) ? uh[i] / div : uh[i];
            he[i] = softplus(v, (REAL)1) + (REAL)1e-3;
        }
    }
    REAL area = 0;
    for (int i = 0; i < K; ++i) area += ((he[i] + he[i + 1]) / (REAL)2) * w[i];
    REAL om_h = (REAL)(1.0 - sp->min_bin_height);
    for (int i = 0; i <= K; ++i) h[i] = minh + om_h * (he[i] / area);
    double a1 = 0.0, a2 = 0.0;
    lcdf[0] = 0;
    loc[0] = 0;
    for (int i = 0; i < K; ++i) {
        a1 += (double)(((h[i] + h[i + 1]) / (REAL)2) * w[i]);
        lcdf[i + 1] = (REAL)a1;
        a2 += (double)w[i];
        loc[i + 1] = (REAL)a2;
    }
    lcdf[K] = (REAL)1;
    loc[K] = (REAL)1;
    int k = search_knots(inverse ? lcdf : loc, K, u);
    if (k < 0 || k >= K) {
        *y = x;
        *lad = 0;
        return ORACLE_STATUS_OUTSIDE_DOMAIN;
    }
    REAL bw = w[k], hl = h[k], hr = h[k + 1];
    REAL a = (REAL)0.5 * (hr - hl) * bw, b = hl * bw, c = lcdf[k];
    REAL out;
    if (inverse) {
        REAL c_ = c - u;
        REAL alpha = (-b + r_sqrt(b * b - (REAL)4 * a * c_)) / ((REAL)2 * a);
        out = clamp01(alpha * bw + loc[k]);
        *lad = -r_log(alpha * (hr - hl) + hl);
    } else {
        REAL alpha = (u - loc[k]) / bw;
        out = clamp01(a * (alpha * alpha) + b * alpha + c);
        *lad = r_log(alpha * (hr - hl) + hl);
    }
    if (inverse) *y = out * (REAL)(sp->right - sp->left) + left;
    else *y = out * (REAL)(sp->top - sp->bottom) + bottom;
    return 0;
}

/* linear_spline / unconstrained_linear_spline (tails = 1: box [-B, B]^2 with B = sp->right,
 * outside elements pass through, splines/linear.py:9-37) over n elements */
int FN(oracle_linear_spline)(const REAL *x, const REAL *updf, int64_t stride, int64_t n,
                             const oracle_rqs_spec *sp, int inverse, REAL *y, REAL *lad) {
    int status = 0;
    if (sp->num_bins < 1 || sp->num_bins > 256) return -1;
    for (int64_t i = 0; i < n; ++i) {
        REAL B = (REAL)sp->right;
        if (sp->tails == 1 && !(x[i] >= -B && x[i] <= B)) {
            y[i] = x[i];
            lad[i] = 0;
            continue;
        }
        status |= linear_one(x[i], updf + i * stride, sp, inverse, y + i, lad + i);
    }
    return status;
}

int FN(oracle_quadratic_spline)(const REAL *x, const REAL *uw, int64_t sw, const REAL *uh, int64_t sh,
                                int nh, int64_t n, const oracle_rqs_spec *sp, int inverse, REAL *y,
                                REAL *lad) {
    int status = 0;
    if (sp->num_bins < 1 || sp->num_bins > 256) return -1;
    for (int64_t i = 0; i < n; ++i) {
        REAL B = (REAL)sp->right;
        if (sp->tails == 1 && !(x[i] >= -B && x[i] <= B)) {
            y[i] = x[i];
            lad[i] = 0;
            continue;
        }
        status |= quadratic_one(x[i], uw + i * sw, uh + i * sh, nh, sp, inverse, y + i, lad + i);
    }
    return status;
}

/* cubic_spline for ONE element: splines/cubic.py:63-267 (Steffen-style monotone cubic, inverse by
 * Blinn's cubic solver with a quadratic fall-back for |a| < quadratic_threshold).  udl / udr: the
 * two boundary-derivative logits. */
static REAL r_cbrt(REAL x) { /* torchutils.cbrt: sign(x) * exp(log|x| / 3) (torchutils.py:139-141) */
    REAL sg = (x > 0) ? (REAL)1 : ((x < 0) ? (REAL)-1 : (REAL)0);
    REAL ax = x < 0 ? -x : x;
    return sg * r_exp(r_log(ax) / (REAL)3);
}

static int cubic_one(REAL x, const REAL *uw, const REAL *uh, REAL udl, REAL udr, const oracle_rqs_spec *sp,
                     int inverse, REAL *y, REAL *lad) {
    enum { KMAX = 256 };
    int K = sp->num_bins;
    REAL w[KMAX], h[KMAX], cw[KMAX + 1], ch[KMAX + 1], sl[KMAX], dv[KMAX + 1];
    REAL left = (REAL)sp->left, right = (REAL)sp->right, bottom = (REAL)sp->bottom;
    REAL eps = (REAL)1e-5, qthr = (REAL)1e-3; /* DEFAULT_EPS, DEFAULT_QUADRATIC_THRESHOLD */
    if (K < 1) return -1;
    sl[0] = 0;
    if (x < left || x > right) {
        *y = x;
        *lad = 0;
        return ORACLE_STATUS_OUTSIDE_DOMAIN;
    }
    REAL u = inverse ? (x - bottom) / (REAL)(sp->top - sp->bottom) : (x - left) / (REAL)(sp->right - sp->left);
    REAL div = (REAL)sp->wh_divisor;
    softmax_k(uw, K, div, w);
    softmax_k(uh, K, div, h);
    REAL omw = (REAL)(1.0 - sp->min_bin_width * K), omh = (REAL)(1.0 - sp->min_bin_height * K);
    double aw = 0.0, ah = 0.0;
    cw[0] = 0;
    ch[0] = 0;
    for (int i = 0; i < K; ++i) {
        w[i] = (REAL)sp->min_bin_width + omw * w[i];
        h[i] = (REAL)sp->min_bin_height + omh * h[i];
        aw += (double)w[i];
        ah += (double)h[i];
        cw[i + 1] = (REAL)aw;
        ch[i + 1] = (REAL)ah;
    }
    cw[K] = 1;
    ch[K] = 1;
    for (int i = 0; i < K; ++i) sl[i] = h[i] / w[i];
    for (int i = 0; i + 1 < K; ++i) {
        REAL a0 = sl[i] < 0 ? -sl[i] : sl[i], a1 = sl[i + 1] < 0 ? -sl[i + 1] : sl[i + 1];
        REAL m1 = a0 < a1 ? a0 : a1;
        REAL m2 = (REAL)0.5 * (w[i + 1] * sl[i] + w[i] * sl[i + 1]) / (w[i] + w[i + 1]);
        REAL m = m1 < m2 ? m1 : m2;
        REAL s0 = sl[i] > 0 ? (REAL)1 : (sl[i] < 0 ? (REAL)-1 : (REAL)0);
        REAL s1 = sl[i + 1] > 0 ? (REAL)1 : (sl[i + 1] < 0 ? (REAL)-1 : (REAL)0);
        dv[i + 1] = m * (s0 + s1);
    }
    dv[0] = ((REAL)1 / ((REAL)1 + r_exp(-udl))) * (REAL)3 * sl[0];
    dv[K] = ((REAL)1 / ((REAL)1 + r_exp(-udr))) * (REAL)3 * sl[K - 1];
    int k = search_knots(inverse ? ch : cw, K, u);
    if (k < 0 || k >= K) {
        *y = x;
        *lad = 0;
        return ORACLE_STATUS_OUTSIDE_DOMAIN;
    }
    REAL a = (dv[k] + dv[k + 1] - (REAL)2 * sl[k]) / (w[k] * w[k]);
    REAL b = ((REAL)3 * sl[k] - (REAL)2 * dv[k] - dv[k + 1]) / w[k];
    REAL c = dv[k];
    REAL d = ch[k];
    REAL lcw = cw[k], rcw = cw[k + 1];
    REAL out;
    if (inverse) {
        REAL b_ = (b / a) / (REAL)3, c_ = (c / a) / (REAL)3, d_ = (d - u) / a;
        REAL d1 = -(b_ * b_) + c_;
        REAL d2 = -c_ * b_ + d_;
        REAL d3 = b_ * d_ - c_ * c_;
        REAL disc = (REAL)4 * d1 * d3 - d2 * d2;
        REAL dep1 = (REAL)-2 * b_ * d1 + d2;
        REAL dep2 = d1;
        out = 0;
        if (disc < 0) {
            REAL sq = r_sqrt(-disc);
            REAL p = r_cbrt((-dep1 + sq) / (REAL)2);
            REAL q = r_cbrt((-dep1 - sq) / (REAL)2);
            out = (p + q) - b_ + lcw;
        } else if (disc >= 0) {
            REAL theta = (REAL)atan2((double)r_sqrt(disc), (double)(-dep1));
            theta = theta / (REAL)3;
            REAL c1 = (REAL)cos((double)theta), s1 = (REAL)sin((double)theta);
            REAL half_sqrt3 = (REAL)(0.5 * sqrt(3.0));
            REAL r1 = c1;
            REAL r2 = (REAL)-0.5 * c1 - half_sqrt3 * s1;
            REAL r3 = (REAL)-0.5 * c1 + half_sqrt3 * s1;
            REAL scale = (REAL)2 * r_sqrt(-dep2);
            REAL shift = -b_ + lcw;
            r1 = r1 * scale + shift;
            r2 = r2 * scale + shift;
            r3 = r3 * scale + shift;
            int m1 = ((lcw - eps) < r1) && (r1 < (rcw + eps));
            int m2 = ((lcw - eps) < r2) && (r2 < (rcw + eps));
            int m3 = ((lcw - eps) < r3) && (r3 < (rcw + eps));
            out = m1 ? r1 : (m2 ? r2 : (m3 ? r3 : r1)); /* argsort(masks, descending)[0] */
        }
        REAL aa = a < 0 ? -a : a;
        if (aa < qthr) { /* almost quadratic (:219-226) */
            REAL qa = b, qb = c, qc = d - u;
            REAL alpha = (-qb + r_sqrt(qb * qb - (REAL)4 * qa * qc)) / ((REAL)2 * qa);
            out = alpha + lcw;
        }
        REAL so = out - lcw;
        *lad = -r_log((REAL)3 * a * (so * so) + (REAL)2 * b * so + c);
        *y = out * (REAL)(sp->right - sp->left) + left;
    } else {
        REAL si = u - lcw;
        out = a * (si * si * si) + b * (si * si) + c * si + d;
        *lad = r_log((REAL)3 * a * (si * si) + (REAL)2 * b * si + c);
        *y = out * (REAL)(sp->top - sp->bottom) + bottom;
    }
    return 0;
}

int FN(oracle_cubic_spline)(const REAL *x, const REAL *uw, int64_t sw, const REAL *uh, int64_t sh,
                            const REAL *udl, int64_t sdl, const REAL *udr, int64_t sdr, int64_t n,
                            const oracle_rqs_spec *sp, int inverse, REAL *y, REAL *lad) {
    int status = 0;
    if (sp->num_bins < 1 || sp->num_bins > 256) return -1;
    for (int64_t i = 0; i < n; ++i) {
        REAL B = (REAL)sp->right;
        if (sp->tails == 1 && !(x[i] >= -B && x[i] <= B)) {
            y[i] = x[i];
            lad[i] = 0;
            continue;
        }
        status |= cubic_one(x[i], uw + i * sw, uh + i * sh, udl[i * sdl], udr[i * sdr], sp, inverse, y + i, lad + i);
    }
    return status;
}
