/*  ff_oracle.c -- CPU oracle (TEST INFRASTRUCTURE, see ff_oracle.h for the pinning status).
 *
 *  A plain-C, scalar restatement of the flip-flop basecalling hot path of
 *  nanoporetech/flappie v2.1.3.  Written from the behaviour of the reference, not from
 *  its text: no SSE, no BLAS, index-space loops.  Build with -ffp-contract=off so that
 *  the element-wise functions reproduce the reference's non-FMA SSE arithmetic lane for
 *  lane (the reference is built -march=ivybridge, CMakeLists.txt:115).
 *
 *  Matrix products are accumulated left to right in fp32; the reference delegates them
 *  to an unpinned system OpenBLAS whose summation order is implementation defined.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "ff_oracle.h"

/* ------------------------------------------------------------------ matrices */

/* flappie_matrix.c:20-51  zero-filled, rows padded to x4 */
fo_mat *fo_make_mat(size_t nr, size_t nc) {
    if (nr == 0 || nc == 0) return NULL;
    fo_mat *m = malloc(sizeof(*m));
    if (!m) return NULL;
    m->nr = nr;
    m->nrq = (nr + 3) / 4;
    m->nc = nc;
    m->stride = m->nrq * 4;
    m->f = calloc(m->stride * nc, sizeof(float));
    if (!m->f) { free(m); return NULL; }
    return m;
}

/* flappie_matrix.c:142-148  returns NULL for the x = free(x) idiom */
fo_mat *fo_free_mat(fo_mat *m) {
    if (m) { free(m->f); free(m); }
    return NULL;
}

fo_imat *fo_make_imat(size_t nr, size_t nc) {
    if (nr == 0 || nc == 0) return NULL;
    fo_imat *m = malloc(sizeof(*m));
    if (!m) return NULL;
    m->nr = nr;
    m->nrq = (nr + 3) / 4;
    m->nc = nc;
    m->stride = m->nrq * 4;
    m->f = calloc(m->stride * nc, sizeof(int32_t));
    if (!m->f) { free(m); return NULL; }
    return m;
}

fo_imat *fo_free_imat(fo_imat *m) {
    if (m) { free(m->f); free(m); }
    return NULL;
}

/* flappie_matrix.c mat_from_array: dense [nr x nc] column-major -> padded */
fo_mat *fo_mat_from_array(const float *x, size_t nr, size_t nc) {
    fo_mat *m = fo_make_mat(nr, nc);
    if (!m) return NULL;
    for (size_t c = 0; c < nc; c++)
        memcpy(m->f + c * m->stride, x + c * nr, nr * sizeof(float));
    return m;
}

/* ------------------------------------------------------------------ vector math */

static inline float as_float(int32_t i) { float f; memcpy(&f, &i, 4); return f; }

/* sse_mathfun.h:225-301 exp_ps, one lane.  Clamp to +-88.376, Cody-Waite reduction with
 * floor-by-truncate-and-fix, degree-5 polynomial, scale by 2^n built in the exponent. */
float fo_expf_cephes(float x) {
    x = (x < 88.3762626647949f) ? x : 88.3762626647949f;     /* _mm_min_ps */
    x = (x > -88.3762626647949f) ? x : -88.3762626647949f;   /* _mm_max_ps */
    float fx = x * 1.44269504088896341f;
    fx = fx + 0.5f;
    int32_t n = (int32_t)fx;                 /* cvttps: truncation toward zero */
    float tmp = (float)n;
    float one_if_gt = (tmp > fx) ? 1.0f : 0.0f;
    fx = tmp - one_if_gt;                    /* floor */
    tmp = fx * 0.693359375f;
    float z = fx * -2.12194440e-4f;
    x = x - tmp;
    x = x - z;
    z = x * x;
    float y = 1.9875691500E-4f;
    y = y * x; y = y + 1.3981999507E-3f;
    y = y * x; y = y + 8.3334519073E-3f;
    y = y * x; y = y + 4.1665795894E-2f;
    y = y * x; y = y + 1.6666665459E-1f;
    y = y * x; y = y + 5.0000001201E-1f;
    y = y * z;
    y = y + x;
    y = y + 1.0f;
    n = (int32_t)fx;
    n = (n + 0x7f) << 23;
    return y * as_float(n);
}

/* util.h:329-332 */
float fo_logisticf(float x) { return 1.0f / (1.0f + fo_expf_cephes(-x)); }

/* util.h:334-337   tanh(x) = 2 logistic(2x) - 1 */
float fo_tanhf(float x) {
    const float y = fo_logisticf(x + x);
    return (y + y) - 1.0f;
}

/* util.h:339-347   x >= 0 ? x : exp(x) - 1  (the and/andnot mask keeps -0.0 as -0.0 >= 0) */
float fo_eluf(float x) { return (x >= 0.0f) ? x : (fo_expf_cephes(x) - 1.0f); }

/* util.h:276-282 */
float fo_logsumexpf(float x, float y) { return fmaxf(x, y) + log1pf(expf(-fabsf(x - y))); }
double fo_logsumexp(double x, double y) { return fmax(x, y) + log1p(exp(-fabs(x - y))); }

/* util.h:284-305  qscoref + phredf; char arithmetic as in the reference */
char fo_phredf(float p) {
    const float p_clip = (p < 0.99999) ? p : 0.99999;       /* MAX_POST_PROB is a double literal */
    const float q = -(10.0f * 0.43429448190325182765) * log1pf(-p_clip);
    char ph = roundf(33.0f + q);
    return (ph < 126) ? ph : 126;
}

/* Array forms of the scalar functions above, for the bit-for-bit sweeps against the reference's header-inline code
 * (oracle/ref_inline.c, tests/test_ref_pins.py).  kind: 0 exp, 1 log, 2 logistic, 3 tanh, 4 elu. */
int fo_map_array(int kind, const float *in, float *out, size_t n) {
    float (*fn)(float) = NULL;
    switch (kind) {
    case 0: fn = fo_expf_cephes; break;
    case 1: fn = fo_logf_cephes; break;
    case 2: fn = fo_logisticf; break;
    case 3: fn = fo_tanhf; break;
    case 4: fn = fo_eluf; break;
    default: return -1;
    }
    for (size_t i = 0; i < n; i++) out[i] = fn(in[i]);
    return 0;
}
void fo_logsumexpf_array(const float *x, const float *y, float *out, size_t n) {
    for (size_t i = 0; i < n; i++) out[i] = fo_logsumexpf(x[i], y[i]);
}
void fo_logsumexp_array(const double *x, const double *y, double *out, size_t n) {
    for (size_t i = 0; i < n; i++) out[i] = fo_logsumexp(x[i], y[i]);
}
void fo_phredf_array(const float *p, char *out, size_t n) {
    for (size_t i = 0; i < n; i++) out[i] = fo_phredf(p[i]);
}

/* layers.c:24-33  applied to every stored element, pad lanes included */
void fo_swish_inplace(fo_mat *C) {
    if (!C) return;
    const size_t n = C->stride * C->nc;
    for (size_t i = 0; i < n; i++) C->f[i] = C->f[i] * fo_logisticf(C->f[i]);
}

/* layers.c:40-49 */
void fo_tanh_inplace(fo_mat *C) {
    if (!C) return;
    const size_t n = C->stride * C->nc;
    for (size_t i = 0; i < n; i++) C->f[i] = fo_tanhf(C->f[i]);
}

/* layers.c:56-66 */
void fo_exp_inplace(fo_mat *C) {
    if (!C) return;
    const size_t n = C->stride * C->nc;
    for (size_t i = 0; i < n; i++) C->f[i] = fo_expf_cephes(C->f[i]);
}

/* sse_mathfun.h:123-208 log_ps, one lane: x <= 0 -> NaN (all-ones), denormals clamped to FLT_MIN */
float fo_logf_cephes(float x) {
    const int invalid = (x <= 0.0f);
    x = (x > 1.17549435e-38f) ? x : 1.17549435e-38f;
    union { float f; int i; unsigned u; } b = { .f = x };
    float e = (float)((int)(b.u >> 23) - 0x7f);
    b.i = (b.i & ~0x7f800000) | 0x3f000000;
    x = b.f;
    e = e + 1.0f;
    const int small = (x < 0.707106781186547524f);
    const float tmp0 = small ? x : 0.0f;
    x = x - 1.0f;
    e = e - (small ? 1.0f : 0.0f);
    x = x + tmp0;
    const float z = x * x;
    float y = 7.0376836292E-2f;
    y = y * x; y = y + -1.1514610310E-1f;
    y = y * x; y = y + 1.1676998740E-1f;
    y = y * x; y = y + -1.2420140846E-1f;
    y = y * x; y = y + 1.4249322787E-1f;
    y = y * x; y = y + -1.6668057665E-1f;
    y = y * x; y = y + 2.0000714765E-1f;
    y = y * x; y = y + -2.4999993993E-1f;
    y = y * x; y = y + 3.3333331174E-1f;
    y = y * x;
    y = y * z;
    float tmp = e * -2.12194440e-4f;
    y = y + tmp;
    tmp = z * 0.5f;
    y = y - tmp;
    tmp = e * 0.693359375f;
    x = x + y;
    x = x + tmp;
    if (invalid) { b.i = -1; return b.f; }
    return x;
}

/* layers.c:73-81 */
void fo_log_inplace(fo_mat *C) {
    if (!C) return;
    const size_t n = C->stride * C->nc;
    for (size_t i = 0; i < n; i++) C->f[i] = fo_logf_cephes(C->f[i]);
}

/* layers.c:88-96 */
void fo_elu_inplace(fo_mat *C) {
    if (!C) return;
    const size_t n = C->stride * C->nc;
    for (size_t i = 0; i < n; i++) C->f[i] = fo_eluf(C->f[i]);
}

/* layers.c:109-124  log(min_prob + (1 - min_prob) * x) */
void fo_robustlog_inplace(fo_mat *C, float min_prob) {
    if (!C) return;
    const size_t n = C->stride * C->nc;
    const float mpm1 = 1.0f - min_prob;
    for (size_t i = 0; i < n; i++) C->f[i] = fo_logf_cephes(min_prob + mpm1 * C->f[i]);
}

/* flappie_matrix.c:392-419  C = Wf^T Xf + Wb^T Xb + b, the two products accumulated one after the other */
fo_mat *fo_affine_map2(const fo_mat *Xf, const fo_mat *Xb, const fo_mat *Wf, const fo_mat *Wb, const fo_mat *b) {
    if (!Xf || !Xb || !Wf || !Wb || !b) return NULL;
    fo_mat *C = fo_affine_map(Xf, Wf, b);
    if (!C) return NULL;
    for (size_t c = 0; c < C->nc; c++)
        for (size_t r = 0; r < C->nr; r++) {
            float acc = 0.0f;
            for (size_t k = 0; k < Wb->nr; k++) acc += Wb->f[r * Wb->stride + k] * Xb->f[c * Xb->stride + k];
            C->f[c * C->stride + r] += acc;
        }
    return C;
}

/* flappie_matrix.c:425-447   each column divided by the sum of its nr real rows.
 * The reference adds 4-lane partial sums, removes the pad lanes of the last quad, then hadd. */
void fo_row_normalise_inplace(fo_mat *C) {
    if (!C) return;
    const size_t npad = C->stride - C->nr;
    for (size_t col = 0; col < C->nc; col++) {
        float *x = C->f + col * C->stride;
        float lane[4] = { x[0], x[1], x[2], x[3] };
        for (size_t q = 1; q < C->nrq; q++)
            for (int l = 0; l < 4; l++) lane[l] += x[4 * q + l];
        /* mask = lanes (3: npad>=1, 2: npad>=2, 1: npad>=3) */
        const float *last = x + 4 * (C->nrq - 1);
        if (npad >= 1) lane[3] -= last[3];
        if (npad >= 2) lane[2] -= last[2];
        if (npad >= 3) lane[1] -= last[1];
        const float p0 = lane[0] + lane[1], p1 = lane[2] + lane[3];   /* hadd twice */
        const float tsum = p0 + p1;
        const float recip = 1.0f / tsum;
        for (size_t r = 0; r < C->stride; r++) x[r] *= recip;
    }
}

/* flappie_matrix.c:450-467  sequential logsumexpf chain over rows 0..nr-1 */
void fo_log_row_normalise_inplace(fo_mat *C) {
    if (!C) return;
    for (size_t col = 0; col < C->nc; col++) {
        float *x = C->f + col * C->stride;
        float row_logsum = x[0];
        for (size_t r = 1; r < C->nr; r++) row_logsum = fo_logsumexpf(row_logsum, x[r]);
        for (size_t r = 0; r < C->nr; r++) x[r] -= row_logsum;
    }
}

/* ------------------------------------------------------------------ network layers */

/* nnfeatures.c:15-28  [1 x nsample], one float every 4 */
fo_mat *fo_features_from_raw(const float *raw, size_t start, size_t end) {
    if (!raw || end <= start) return NULL;
    fo_mat *m = fo_make_mat(1, end - start);
    if (!m) return NULL;
    for (size_t i = 0; i < end - start; i++) m->f[4 * i] = raw[start + i];
    return m;
}

/* How dot products are summed.  0 (default, THE oracle): float, term by term in index order.  1: the same terms in a double
 * accumulator, rounded to float once per dot product -- "what the float32 network would give without summation error", the
 * yardstick tests use to tell the GPU's rounding from the oracle's own (tests/test_fuzz_tail_gpu.py).  2: vectorised kernels
 * (cpu_ref.c; 16 interleaved partial sums), the shapes a BLAS would run -- only for bench.py's cpu_baseline leg. */
static int g_dot_mode = 0;
void fo_set_dot_mode(int mode) { g_dot_mode = mode; }
int fo_get_dot_mode(void) { return g_dot_mode; }
void fo_fast_gemv_t(float *y, const float *W, size_t ldW, size_t nout, size_t len, const float *x);
void fo_fast_gemm_tn(float *Y, size_t ldY, const float *W, size_t ldW, size_t nout, size_t len,
                     const float *X, size_t ldX, size_t ncol);

/* Dot mode 3: the same two shapes through a real OpenBLAS, where the host has one (bench.py's cpu_baseline: "Flappie's own OpenBLAS
 * CPU path" -- the reference calls cblas_sgemv(ColMajor, Trans, ...) at layers.c:1009 / :697 / :224 / :268 and cblas_sgemm(ColMajor,
 * Trans, NoTrans, ...) at flappie_matrix.c:384 / layers.c:250 with exactly these arguments).  The image has no <cblas.h>; the library
 * is dlopen()ed and called through the prototypes below (the CBLAS ABI, LP64 build: scipy's `scipy_cblas_*`, or plain `cblas_*`). */
#include <dlfcn.h>
#include <stdio.h>
typedef void (*sgemv_fn)(int order, int trans, int m, int n, float alpha, const float *a, int lda, const float *x, int incx, float beta, float *y, int incy);
typedef void (*sgemm_fn)(int order, int ta, int tb, int m, int n, int k, float alpha, const float *a, int lda, const float *b, int ldb, float beta, float *c, int ldc);
static sgemv_fn g_sgemv = NULL;
static sgemm_fn g_sgemm = NULL;
static char g_blas_config[256];
/* 0 = ready (mode 3 usable); the library's threads are set to one, as the reference's README asks (OPENBLAS_NUM_THREADS=1) */
int fo_blas_open(const char *path) {
    void *h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!h) return -1;
    static const char *pre[2] = { "scipy_", "" };
    for (int k = 0; k < 2; k++) {
        char name[64];
        snprintf(name, sizeof name, "%scblas_sgemv", pre[k]);
        sgemv_fn gv = (sgemv_fn)dlsym(h, name);
        snprintf(name, sizeof name, "%scblas_sgemm", pre[k]);
        sgemm_fn gm = (sgemm_fn)dlsym(h, name);
        if (!gv || !gm) continue;
        snprintf(name, sizeof name, "%sopenblas_set_num_threads", pre[k]);
        void (*setn)(int) = (void (*)(int))dlsym(h, name);
        if (setn) setn(1);
        snprintf(name, sizeof name, "%sopenblas_get_config", pre[k]);
        char *(*cfg)(void) = (char *(*)(void))dlsym(h, name);
        snprintf(g_blas_config, sizeof g_blas_config, "%s", cfg ? cfg() : "unknown BLAS");
        g_sgemv = gv; g_sgemm = gm;
        return 0;
    }
    return -2;
}
const char *fo_blas_config(void) { return g_sgemv ? g_blas_config : ""; }
static void dot_gemv_t(float *y, const float *W, size_t ldW, size_t nout, size_t len, const float *x) {
    if (g_dot_mode == 3 && g_sgemv) g_sgemv(102 /*ColMajor*/, 112 /*Trans*/, (int)len, (int)nout, 1.0f, W, (int)ldW, x, 1, 1.0f, y, 1);
    else fo_fast_gemv_t(y, W, ldW, nout, len, x);
}
static void dot_gemm_tn(float *Y, size_t ldY, const float *W, size_t ldW, size_t nout, size_t len, const float *X, size_t ldX, size_t ncol) {
    if (g_dot_mode == 3 && g_sgemm) g_sgemm(102, 112, 111 /*NoTrans*/, (int)nout, (int)ncol, (int)len, 1.0f, W, (int)ldW, X, (int)ldX, 1.0f, Y, (int)ldY);
    else fo_fast_gemm_tn(Y, ldY, W, ldW, nout, len, X, ldX, ncol);
}
void fo_fast_lstm_gates(const float *xF, float *state, float *hout, size_t size);
void fo_fast_grumod_gates(float *xF, const float *x, const float *hprev, float *hout, size_t size);

/* y[f] += sum_{i<len} W[woff + i + f*ldW] * x[i]     (the reference's sgemv(T) shape) */
static void window_accumulate(float *y, const fo_mat *W, size_t woff, size_t len, const float *x) {
    if (g_dot_mode >= 2) {
        dot_gemv_t(y, W->f + woff, W->stride, W->nc, len, x);
        return;
    }
    if (g_dot_mode == 1) {
        for (size_t f = 0; f < W->nc; f++) {
            const float *w = W->f + f * W->stride + woff;
            double acc = 0.0;
            for (size_t i = 0; i < len; i++) acc += (double)w[i] * (double)x[i];
            y[f] = (float)((double)y[f] + acc);
        }
        return;
    }
    for (size_t f = 0; f < W->nc; f++) {
        const float *w = W->f + f * W->stride + woff;
        float acc = y[f];
        for (size_t i = 0; i < len; i++) acc += w[i] * x[i];
        y[f] = acc;
    }
}

/* layers.c:189-276  strided convolution INCLUDING the reference's right-edge behaviour.
 * X columns are contiguous (stride ldX = 4*ceil(features/4)) so a window of `winlen` columns is
 * one contiguous vector of W->nr floats.  Three regions, in the reference's order:
 *   left edge  (:220-226)  partial windows starting before X
 *   body       (:239-254)  full windows, in nstepC interleaved column families
 *   right edge (:257-271)  partial windows, accumulated at the column index the reference
 *                          computes -- which for stride>1 is not always the naive one. */
fo_mat *fo_convolution(const fo_mat *X, const fo_mat *W, const fo_mat *b, size_t stride) {
    if (!X || !W || !b || stride == 0) return NULL;
    if (W->nrq % X->nrq != 0) return NULL;
    const long winlen = (long)(W->nrq / X->nrq);
    const long s = (long)stride;
    const long T = (long)X->nc;
    const long padL = (winlen - 1) / 2;
    const long padR = winlen / 2;
    const long ncolC = (T + s - 1) / s;
    const long ldX = (long)X->stride;
    const long ncolsL = (padL + s - 1) / s;
    const long shiftX = ncolsL * s - padL;
    const long nstepC = (winlen + s - 1) / s;
    const long nstepX = s * nstepC;
    /* domain of the reference: every body family must have a non-negative column count */
    if (T - shiftX - (winlen - 1) < 0 || T < winlen) return NULL;

    fo_mat *C = fo_make_mat(W->nc, (size_t)ncolC);
    if (!C) return NULL;
    const long ldC = (long)C->stride;

    for (long c = 0; c < ncolC; c++)                          /* bias fill :215-217 */
        memcpy(C->f + c * ldC, b->f, C->stride * sizeof(float));

    for (long w = 0; w < padL; w += s) {                      /* left edge */
        const long woff = ldX * (padL - w);
        window_accumulate(C->f + ldC * (w / s), W, (size_t)woff, W->nr - (size_t)woff, X->f);
    }

    for (long w = 0; w < winlen; w += s) {                    /* body */
        const long ncol = (T - shiftX - w) / nstepX;          /* ifloor, :248 */
        const long col0 = ncolsL + w / s;
        if (g_dot_mode >= 2 && ncol > 0) {                    /* the reference's one sgemm per family, :250 */
            dot_gemm_tn(C->f + ldC * col0, (size_t)(ldC * nstepC), W->f, W->stride, W->nc, W->nr,
                            X->f + ldX * (shiftX + w), (size_t)(ldX * nstepX), (size_t)ncol);
            continue;
        }
        for (long k = 0; k < ncol; k++) {
            const long xstart = shiftX + w + nstepX * k;
            window_accumulate(C->f + ldC * (col0 + nstepC * k), W, 0, W->nr, X->f + ldX * xstart);
        }
    }

    {                                                         /* right edge */
        const long maxCol = (T - shiftX) / nstepX;
        const long rem = (T - shiftX) % nstepX;
        const long colR = ncolsL + nstepC * (maxCol - 1) + rem / s + 1;      /* offsetC_R / ldC */
        const long xR = T - winlen + 1;
        const long startR = s - (padL + T - winlen) % s - 1;
        for (long w = startR; w < padR; w += s) {
            const long woff = ldX * (w + 1);
            const long col = colR + w / s;
            if (col < 0 || col >= ncolC) continue;            /* reference would write out of bounds */
            window_accumulate(C->f + ldC * col, W, 0, W->nr - (size_t)woff, X->f + ldX * (xR + w));
        }
    }
    return C;
}

/* flappie_matrix.c:361-389   C = W^T X + b ;  X [nr x nc], W [nr x nk], b [nk] */
fo_mat *fo_affine_map(const fo_mat *X, const fo_mat *W, const fo_mat *b) {
    if (!X || !W || !b || W->nr != X->nr) return NULL;
    fo_mat *C = fo_make_mat(W->nc, X->nc);
    if (!C) return NULL;
    for (size_t c = 0; c < X->nc; c++) memcpy(C->f + c * C->stride, b->f, C->stride * sizeof(float));
    if (g_dot_mode >= 2) {
        dot_gemm_tn(C->f, C->stride, W->f, W->stride, W->nc, W->nr, X->f, X->stride, X->nc);
        return C;
    }
    for (size_t c = 0; c < X->nc; c++) window_accumulate(C->f + c * C->stride, W, 0, W->nr, X->f + c * X->stride);
    return C;
}

/* layers.c:979-1026 lstm_step.  gate order i,f,g,o in blocks of `size` rows:
 *   c = sigma(f) * c + sigma(i) * tanh(g) ;  h = sigma(o) * tanh(c) */
static void lstm_step(const float *xaff, const float *hprev, const fo_mat *sW,
                      float *xF, float *state, float *hout) {
    const size_t size = sW->nr;
    memcpy(xF, xaff, 4 * size * sizeof(float));
    window_accumulate(xF, sW, 0, size, hprev);
    if (g_dot_mode >= 2) { fo_fast_lstm_gates(xF, state, hout, size); return; }
    for (size_t i = 0; i < size; i++) {
        const float forget = fo_logisticf(xF[size + i]) * state[i];
        const float update = fo_logisticf(xF[i]) * fo_tanhf(xF[2 * size + i]);
        state[i] = forget + update;
        hout[i] = fo_logisticf(xF[3 * size + i]) * fo_tanhf(state[i]);
    }
}

/* layers.c:877-976 lstm_forward / lstm_backward.  h and c start at zero; forward walks
 * t = 0..n-1 reading out[t-1], backward walks t = n-1..0 reading out[t+1]. */
fo_mat *fo_lstm(const fo_mat *Xaffine, const fo_mat *sW, int backward) {
    if (!Xaffine || !sW) return NULL;
    const size_t size = sW->nr, n = Xaffine->nc;
    if (Xaffine->nr != 4 * size || sW->nc != 4 * size || size % 4 != 0) return NULL;
    fo_mat *out = fo_make_mat(size, n);
    float *xF = calloc(4 * size, sizeof(float));
    float *state = calloc(size, sizeof(float));
    float *zero = calloc(size, sizeof(float));
    if (!out || !xF || !state || !zero) { free(xF); free(state); free(zero); return fo_free_mat(out); }
    for (size_t i = 0; i < n; i++) {
        const size_t t = backward ? n - 1 - i : i;
        const float *hprev = (i == 0) ? zero
                           : out->f + (backward ? t + 1 : t - 1) * out->stride;
        lstm_step(Xaffine->f + t * Xaffine->stride, hprev, sW, xF, state, out->f + t * out->stride);
    }
    free(xF); free(state); free(zero);
    return out;
}

/* layers.c:664-715 grumod_step.  gate order z,r,candidate:
 *   xF[0:2H] = x[0:2H] + sW[:,0:2H]^T h ; u = sW[:,2H:3H]^T h  (third chunk zeroed first, :691)
 *   z = sigma, r = sigma ; hbar = tanh(r*u + x[2H:3H]) ; h' = z*h + (1-z)*hbar */
static void grumod_step(const float *x, const float *hprev, const fo_mat *sW, float *xF, float *hout) {
    const size_t size = sW->nr;
    memcpy(xF, x, 3 * size * sizeof(float));
    memset(xF + 2 * size, 0, size * sizeof(float));
    window_accumulate(xF, sW, 0, size, hprev);
    if (g_dot_mode >= 2) { fo_fast_grumod_gates(xF, x, hprev, hout, size); return; }
    for (size_t i = 0; i < 2 * size; i++) xF[i] = fo_logisticf(xF[i]);
    const float *z = xF, *r = xF + size;
    float *hbar = xF + 2 * size;
    for (size_t i = 0; i < size; i++) hbar[i] = r[i] * hbar[i] + x[2 * size + i];
    for (size_t i = 0; i < size; i++) hbar[i] = fo_tanhf(hbar[i]);
    for (size_t i = 0; i < size; i++) hout[i] = z[i] * hprev[i] + (1.0f - z[i]) * hbar[i];
}

/* layers.c:571-661 grumod_forward / grumod_backward */
fo_mat *fo_grumod(const fo_mat *X, const fo_mat *sW, int backward) {
    if (!X || !sW) return NULL;
    const size_t size = sW->nr, n = X->nc;
    if (X->nr != 3 * size || sW->nc != 3 * size || size % 4 != 0) return NULL;
    fo_mat *out = fo_make_mat(size, n);
    float *xF = calloc(3 * size, sizeof(float));
    float *zero = calloc(size, sizeof(float));
    if (!out || !xF || !zero) { free(xF); free(zero); return fo_free_mat(out); }
    for (size_t i = 0; i < n; i++) {
        const size_t t = backward ? n - 1 - i : i;
        const float *hprev = (i == 0) ? zero
                           : out->f + (backward ? t + 1 : t - 1) * out->stride;
        grumod_step(X->f + t * X->stride, hprev, sW, xF, out->f + t * out->stride);
    }
    free(xF); free(zero);
    return out;
}

/* layers.c:513-568 gru_step and layers.c:819-874 gru_relu_step -- the sloika GRU (gate order z, r, candidate; used only by
 * networks.c:403/:492, which no registered model reaches):
 *   xF = x ; xF[0:2H] += sW^T h ; z, r = sigma ; hbar = act(x[2H:3H] + sW2^T (r*h)) ; h' = z*h + (1-z)*hbar */
static void gru_step(const float *x, const float *hprev, const fo_mat *sW, const fo_mat *sW2, int relu, float *xF, float *hout) {
    const size_t size = sW2->nc;
    memcpy(xF, x, 3 * size * sizeof(float));
    window_accumulate(xF, sW, 0, size, hprev);
    for (size_t i = 0; i < 2 * size; i++) xF[i] = fo_logisticf(xF[i]);
    const float *z = xF;
    float *r = xF + size, *hbar = xF + 2 * size;
    for (size_t i = 0; i < size; i++) r[i] *= hprev[i];
    window_accumulate(hbar, sW2, 0, size, r);
    for (size_t i = 0; i < size; i++) hbar[i] = relu ? fmaxf(hbar[i], 0.0f) : fo_tanhf(hbar[i]);
    for (size_t i = 0; i < size; i++) hout[i] = z[i] * hprev[i] + (1.0f - z[i]) * hbar[i];
}

/* layers.c:412-510 gru_forward / gru_backward, layers.c:718-816 gru_relu_forward / gru_relu_backward.
 * h0 (nullable) replaces the zero start state: with n = 1 this is gru_step / gru_relu_step itself. */
fo_mat *fo_gru(const fo_mat *X, const fo_mat *sW, const fo_mat *sW2, int backward, int relu, const float *h0) {
    if (!X || !sW || !sW2) return NULL;
    const size_t size = sW2->nc, n = X->nc;
    if (X->nr != 3 * size || sW->nr != size || sW2->nr != size || sW->nc != 2 * size || size % 4 != 0) return NULL;
    fo_mat *out = fo_make_mat(size, n);
    float *xF = calloc(3 * size, sizeof(float));
    float *zero = calloc(size, sizeof(float));
    if (!out || !xF || !zero) { free(xF); free(zero); return fo_free_mat(out); }
    for (size_t i = 0; i < n; i++) {
        const size_t t = backward ? n - 1 - i : i;
        const float *hprev = (i == 0) ? (h0 ? h0 : zero) : out->f + (backward ? t + 1 : t - 1) * out->stride;
        gru_step(X->f + t * X->stride, hprev, sW, sW2, relu, xF, out->f + t * out->stride);
    }
    free(xF); free(zero);
    return out;
}

/* layers.c:1029-1032 */
size_t fo_nbase_from_nparam(size_t nparam) {
    return (size_t)roundf((-1.0f + sqrtf(1 + 2 * nparam)) / 2.0f);
}

/* layers.c:1035-1079  fp64 forward recursion from zeros */
double fo_partition_function(const fo_mat *C) {
    if (!C) return NAN;
    const size_t nbase = fo_nbase_from_nparam(C->nr);
    const size_t nstate = 2 * nbase;
    if (nstate * (nbase + 1) != C->nr) return NAN;
    double mem[2 * 64] = { 0 };
    if (nstate > 64) return NAN;
    double *curr = mem, *prev = mem + nstate;
    for (size_t c = 0; c < C->nc; c++) {
        const float *col = C->f + c * C->stride;
        const float *stay = col + nstate * nbase;
        { double *t = curr; curr = prev; prev = t; }
        for (size_t st = nbase; st < nstate; st++) {
            const size_t from = st - nbase;
            curr[st] = fo_logsumexp(prev[st] + stay[st], prev[from] + stay[from]);
        }
        for (size_t to = 0; to < nbase; to++) {
            const float *row = col + to * nstate;
            curr[to] = row[0] + prev[0];
            for (size_t from = 1; from < nstate; from++)
                curr[to] = fo_logsumexp(curr[to], row[from] + prev[from]);
        }
    }
    double logZ = curr[0];
    for (size_t st = 1; st < nstate; st++) logZ = fo_logsumexp(logZ, curr[st]);
    return logZ;
}

/* layers.c:1082-1106 globalnorm_manystay == globalnorm_flipflop.
 * tanh on everything, (x - 0)/(temperature/5) on the nr real rows (flappie_matrix.c:625-633),
 * logZ/nblock rounded to float, subtracted from the nr real rows. */
fo_mat *fo_globalnorm_flipflop(const fo_mat *X, const fo_mat *W, const fo_mat *b, float temperature) {
    fo_mat *C = fo_affine_map(X, W, b);
    if (!C) return NULL;
    fo_tanh_inplace(C);
    const float scale = temperature / 5.0f;
    for (size_t c = 0; c < C->nc; c++)
        for (size_t r = 0; r < C->nr; r++)
            C->f[c * C->stride + r] = (C->f[c * C->stride + r] - 0.0f) / scale;
    const float logZ = fo_partition_function(C) / (double)C->nc;
    for (size_t c = 0; c < C->nc; c++)
        for (size_t r = 0; r < C->nr; r++)
            C->f[c * C->stride + r] -= logZ;
    return C;
}

/* networks.c:539-586 (LSTM5) and :450-489 (GRUMOD5) */
fo_mat *fo_transitions(const float *raw, size_t start, size_t end, float temperature,
                       const fo_model *net) {
    if (!raw || !net || end <= start) return NULL;
    fo_mat *x = fo_features_from_raw(raw, start, end);
    for (int l = 0; x && l < net->nconv; l++) {
        fo_mat *y = fo_convolution(x, net->conv_W[l], net->conv_b[l], (size_t)net->conv_stride[l]);
        if (net->kind != FO_NET_GRUMOD5) fo_swish_inplace(y); else fo_tanh_inplace(y);
        fo_free_mat(x);
        x = y;
    }
    for (int l = 0; x && l < 5; l++) {
        const int backward = (l % 2 == 0);                   /* B,F,B,F,B */
        fo_mat *in = fo_affine_map(x, net->rnn_iW[l], net->rnn_b[l]);
        fo_free_mat(x);
        x = (net->kind != FO_NET_GRUMOD5) ? fo_lstm(in, net->rnn_sW[l], backward)
                                        : fo_grumod(in, net->rnn_sW[l], backward);
        fo_free_mat(in);
    }
    if (!x) return NULL;
    fo_mat *trans = (net->kind == FO_NET_LSTM5_RLE) ? fo_globalnorm_runlengthV2(x, net->FF_W, net->FF_b, temperature)    /* networks.c:716-722 */
                                                    : fo_globalnorm_flipflop(x, net->FF_W, net->FF_b, temperature);
    fo_free_mat(x);
    return trans;
}

size_t fo_nblock_for(const fo_model *net, size_t nsample) {
    size_t n = nsample;
    for (int l = 0; l < net->nconv; l++) n = (n + net->conv_stride[l] - 1) / net->conv_stride[l];
    return n;
}

/* ------------------------------------------------------------------ decode */

/* decode.c:104-114 */
static size_t trans_lookup(size_t from, size_t to, size_t nbase) {
    const size_t nstate = 2 * nbase;
    return (to < nbase) ? (to * nstate + from) : (nbase * nstate + from);
}

/* decode.c:377-497 transpost_crf_flipflop */
fo_mat *fo_transpost(const fo_mat *trans, int return_log) {
    if (!trans) return NULL;
    const size_t nblk = trans->nc;
    const size_t nbase = fo_nbase_from_nparam(trans->nr);
    const size_t nstate = 2 * nbase;
    if (nstate * (nbase + 1) != trans->nr || nstate > 64) return NULL;
    fo_mat *fwd = fo_make_mat(nstate, nblk + 1);
    fo_mat *tpost = fo_make_mat(trans->nr, nblk);
    if (!fwd || !tpost) { fo_free_mat(fwd); return fo_free_mat(tpost); }

    for (size_t blk = 0; blk < nblk; blk++) {                 /* forwards :396-423 */
        const float *T = trans->f + blk * trans->stride;
        const float *Tflop = T + nstate * nbase;
        const float *prev = fwd->f + blk * fwd->stride;
        float *curr = fwd->f + (blk + 1) * fwd->stride;
        for (size_t b2 = nbase; b2 < nstate; b2++) {
            const size_t fb = b2 - nbase;
            const float stay = prev[b2] + Tflop[b2];
            const float move = prev[fb] + Tflop[fb];
            curr[b2] = fo_logsumexpf(stay, move);
        }
        for (size_t b1 = 0; b1 < nbase; b1++) {
            const float *row = T + b1 * nstate;
            curr[b1] = row[0] + prev[0];
            for (size_t from = 1; from < nstate; from++)
                curr[b1] = fo_logsumexpf(curr[b1], row[from] + prev[from]);
        }
    }

    float mem[2 * 64] = { 0 };
    float *prev = mem, *curr = mem + nstate;
    for (size_t blk = nblk; blk > 0; blk--) {                 /* backwards :434-484 */
        const float *F = fwd->f + (blk - 1) * fwd->stride;
        const float *T = trans->f + (blk - 1) * trans->stride;
        const float *Tflop = T + nstate * nbase;
        float *P = tpost->f + (blk - 1) * tpost->stride;
        float *Pflop = P + nstate * nbase;
        { float *t = prev; prev = curr; curr = t; }
        for (size_t b1 = 0; b1 < nbase; b1++)
            for (size_t st = 0; st < nstate; st++)
                P[b1 * nstate + st] = F[st] + prev[b1] + T[b1 * nstate + st];
        for (size_t b = nbase; b < nstate; b++) {
            const size_t fb = b - nbase;
            Pflop[b] = F[b] + prev[b] + Tflop[b];
            Pflop[fb] = F[fb] + prev[b] + Tflop[fb];
        }
        for (size_t b2 = nbase; b2 < nstate; b2++) {
            const size_t fb = b2 - nbase;
            curr[b2] = prev[b2] + Tflop[b2];
            curr[fb] = prev[b2] + Tflop[fb];
        }
        for (size_t b1 = 0; b1 < nbase; b1++)
            for (size_t from = 0; from < nstate; from++)
                curr[from] = fo_logsumexpf(curr[from], T[b1 * nstate + from] + prev[b1]);
    }
    fo_free_mat(fwd);
    fo_log_row_normalise_inplace(tpost);
    if (!return_log) fo_exp_inplace(tpost);
    return tpost;
}

/* decode.c:119-204  Viterbi with the reference's tie rules: stay beats move unless move is
 * strictly greater; flip states scan from=0.. and keep the lowest index on ties; final argmax is
 * the first maximum (util.c:17-31). */
float fo_decode_viterbi(const fo_mat *trans, int combine_stays, int *path, float *qpath) {
    if (!trans || !path || !qpath) return NAN;
    const size_t nblk = trans->nc;
    const size_t nbase = fo_nbase_from_nparam(trans->nr);
    const size_t nstate = 2 * nbase;
    if (nstate * (nbase + 1) != trans->nr || nstate > 64) return NAN;
    fo_imat *tb = fo_make_imat(nstate, nblk);
    if (!tb) return NAN;
    float mem[2 * 64] = { 0 };
    float *curr = mem, *prev = mem + nstate;

    for (size_t blk = 0; blk < nblk; blk++) {
        const float *T = trans->f + blk * trans->stride;
        const float *Tflop = T + nstate * nbase;
        int32_t *tbc = tb->f + blk * tb->stride;
        { float *t = curr; curr = prev; prev = t; }
        for (size_t b2 = nbase; b2 < nstate; b2++) {
            const size_t fb = b2 - nbase;
            curr[b2] = prev[b2] + Tflop[b2];
            tbc[b2] = (int32_t)b2;
            const float score = prev[fb] + Tflop[fb];
            if (score > curr[b2]) { curr[b2] = score; tbc[b2] = (int32_t)fb; }
        }
        for (size_t b1 = 0; b1 < nbase; b1++) {
            const float *row = T + b1 * nstate;
            curr[b1] = row[0] + prev[0];
            tbc[b1] = 0;
            for (size_t from = 1; from < nstate; from++) {
                const float score = row[from] + prev[from];
                if (score > curr[b1]) { curr[b1] = score; tbc[b1] = (int32_t)from; }
            }
        }
    }

    float score = curr[0];
    int imax = 0;
    for (size_t i = 1; i < nstate; i++) if (curr[i] > score) { score = curr[i]; imax = (int)i; }
    path[nblk] = imax;
    for (size_t blk = nblk; blk > 0; blk--) {
        path[blk - 1] = tb->f[(blk - 1) * tb->stride + path[blk]];
        qpath[blk] = trans->f[(blk - 1) * trans->stride
                              + trans_lookup((size_t)path[blk - 1], (size_t)path[blk], nbase)];
    }
    qpath[0] = NAN;
    if (combine_stays)
        for (size_t blk = 0; blk <= nblk; blk++)
            path[blk] = (path[blk] < (int)nbase) ? path[blk] : -1;
    fo_free_imat(tb);
    return score;
}

/* decode.c:66-79 */
size_t fo_change_positions(const int *path, size_t npos, int *chpos) {
    if (!path || !chpos) return 0;
    size_t nch = 0;
    for (size_t pos = 1; pos < npos; pos++) {
        if (path[pos] == path[pos - 1]) continue;
        chpos[nch++] = (int)pos;
    }
    return nch;
}

/* decode.c:499-543  tpost holds probabilities (after exp_activation_inplace) */
fo_imat *fo_trace_from_posterior(const fo_mat *tpost) {
    if (!tpost) return NULL;
    const size_t nbase = fo_nbase_from_nparam(tpost->nr);
    const size_t nstate = 2 * nbase;
    if ((nbase + 1) * nstate != tpost->nr) return NULL;
    fo_imat *trace = fo_make_imat(nstate, tpost->nc + 1);
    if (!trace) return NULL;
    for (size_t from = 0; from < nstate; from++) {            /* first position, by from-state */
        float sum = 0.0f;
        for (size_t to = 0; to < nbase; to++) sum += tpost->f[to * nstate + from];
        sum += tpost->f[nbase * nstate + from];
        trace->f[from] = (int32_t)roundf(255.0f * sum);
    }
    for (size_t blk = 0; blk < tpost->nc; blk++) {
        int32_t *tr = trace->f + (blk + 1) * trace->stride;
        const float *P = tpost->f + blk * tpost->stride;
        for (size_t to = 0; to < nbase; to++) {
            float sum = P[to * nstate];
            for (size_t from = 1; from < nstate; from++) sum += P[to * nstate + from];
            tr[to] = (int32_t)roundf(255.0f * sum);
        }
        const float *Pflop = P + nbase * nstate;
        for (size_t to = nbase; to < nstate; to++) {
            const float sum = Pflop[to - nbase] + Pflop[to];
            tr[to] = (int32_t)roundf(255.0f * sum);
        }
    }
    return trace;
}

/* flappie.c:245-316 calculate_post, from the normalised signal onwards */
int fo_basecall_read(const float *raw, size_t start, size_t end, float temperature,
                     const fo_model *net, int viterbi_only,
                     fo_read_result *res, int *path, float *qpath,
                     char *basecall, char *quality, int32_t *trace,
                     float *trans_out, float *post_out) {
    static const char base_lookup[5] = { 'A', 'C', 'G', 'T', 'Z' };    /* decode.h:16 */
    fo_mat *trans = fo_transitions(raw, start, end, temperature, net);
    if (!trans) return -1;
    const size_t nbase = fo_nbase_from_nparam(trans->nr);
    const size_t nstate = 2 * nbase;
    const size_t nblock = trans->nc;
    if (trans_out)
        for (size_t c = 0; c < nblock; c++)
            memcpy(trans_out + c * trans->nr, trans->f + c * trans->stride, trans->nr * sizeof(float));
    fo_mat *post = trans;
    if (!viterbi_only) {
        post = fo_transpost(trans, 1);
        fo_free_mat(trans);
        if (!post) return -1;
    }
    if (post_out)
        for (size_t c = 0; c < nblock; c++)
            memcpy(post_out + c * post->nr, post->f + c * post->stride, post->nr * sizeof(float));
    const float score = fo_decode_viterbi(post, 0, path, qpath);
    int *idx = calloc(nblock + 2, sizeof(int));
    const size_t nidx = fo_change_positions(path, nblock, idx);
    for (size_t i = 0; i < nidx; i++) {
        basecall[i] = base_lookup[path[idx[i]] % (int)nbase];
        quality[i] = fo_phredf(expf(qpath[idx[i]]));
    }
    basecall[nidx] = 0;
    quality[nidx] = 0;
    free(idx);
    if (trace) {
        fo_exp_inplace(post);
        fo_imat *tr = fo_trace_from_posterior(post);
        for (size_t c = 0; c <= nblock; c++)
            memcpy(trace + c * nstate, tr->f + c * tr->stride, nstate * sizeof(int32_t));
        fo_free_imat(tr);
    }
    fo_free_mat(post);
    if (res) {
        res->nblock = nblock; res->nbase = nbase; res->nstate = nstate;
        res->nparam = nstate * (nbase + 1);
        res->score = score; res->basecall_length = nidx;
    }
    return 0;
}

/* ------------------------------------------------------------------ signal preparation */

/* util.c:74-80 floatcmp: returns -1 for equal elements, as the reference does */
static int floatcmp(const void *x, const void *y) {
    const float d = *(const float *)x - *(const float *)y;
    return (d > 0) ? 1 : -1;
}

/* util.c:100-138 */
void fo_quantilef(const float *x, size_t nx, float *p, size_t np) {
    if (!p) return;
    float *space = x ? malloc(nx * sizeof(float)) : NULL;
    if (!space) { for (size_t i = 0; i < np; i++) p[i] = NAN; return; }
    memcpy(space, x, nx * sizeof(float));
    qsort(space, nx, sizeof(float), floatcmp);
    for (size_t i = 0; i < np; i++) {
        const size_t idx = p[i] * (nx - 1);
        const float remf = p[i] * (nx - 1) - idx;
        if (idx < nx - 1) p[i] = (1.0 - remf) * space[idx] + remf * space[idx + 1];
        else p[i] = space[idx];
    }
    free(space);
}

/* util.c:150-154 */
float fo_medianf(const float *x, size_t n) { float p = 0.5; fo_quantilef(x, n, &p, 1); return p; }

/* util.c:164-187 */
float fo_madf(const float *x, size_t n, const float *med) {
    const float mad_scaling_factor = 1.4826;
    if (!x) return NAN;
    if (n == 1) return 0.0f;
    float *absdiff = malloc(n * sizeof(float));
    if (!absdiff) return NAN;
    const float m = med ? *med : fo_medianf(x, n);
    for (size_t i = 0; i < n; i++) absdiff[i] = fabsf(x[i] - m);
    const float mad = fo_medianf(absdiff, n);
    free(absdiff);
    return mad * mad_scaling_factor;
}

/* util.c:198-212 */
void fo_medmad_normalise_array(float *x, size_t n) {
    if (!x) return;
    if (n == 1) { x[0] = 0.0; return; }
    const float xmed = fo_medianf(x, n);
    const float xmad = fo_madf(x, n, &xmed);
    for (size_t i = 0; i < n; i++) x[i] = (x[i] - xmed) / xmad;
}

/* flappie_common.c:47-81 */
int fo_trim_raw_by_mad(const float *raw, size_t *start, size_t *end, size_t chunk_size, float perc) {
    if (!raw || chunk_size < 2) return -1;
    const size_t nsample = *end - *start;
    const size_t nchunk = nsample / chunk_size;
    *end = nchunk * chunk_size;
    float *madarr = malloc((nchunk ? nchunk : 1) * sizeof(float));
    if (!madarr) return -1;
    for (size_t i = 0; i < nchunk; i++)
        madarr[i] = fo_madf(raw + *start + i * chunk_size, chunk_size, NULL);
    fo_quantilef(madarr, nchunk, &perc, 1);
    const float thresh = perc;
    for (size_t i = 0; i < nchunk; i++) { if (madarr[i] > thresh) break; *start += chunk_size; }
    for (size_t i = nchunk; i > 0; i--) { if (madarr[i - 1] > thresh) break; *end -= chunk_size; }
    free(madarr);
    return 0;
}

/* flappie_common.c:13-28 */
int fo_trim_and_segment_raw(const float *raw, size_t n, size_t *start, size_t *end,
                            size_t trim_start, size_t trim_end, size_t varseg_chunk, float varseg_thresh) {
    if (fo_trim_raw_by_mad(raw, start, end, varseg_chunk, varseg_thresh)) return -1;
    *start = (n - *start) > trim_start ? *start + trim_start : n;
    *end = (*end > trim_end) ? *end - trim_end : 0;
    return (*start >= *end) ? -1 : 0;
}

/* ------------------------------------------------------------------ run-length (runnie) head and decoders */

/* util.h:83-85 */
float fo_softplusf(float x) { return log1pf(expf(-fabsf(x))) + ((x >= 0.0f) ? x : 0.f); }

/* layers.c:1127-1174  first-generation run-length model: rows shape, scale, move, stay (nbase each); fp64 recursion over
 * the nbase states: a move into b1 from every other base, or a stay */
double fo_runlength_partition_function(const fo_mat *C) {
    if (!C || C->nr % 4 != 0 || C->nr / 4 > 64) return NAN;
    const size_t nbase = C->nr / 4;
    double mem[2 * 64] = { 0 };
    double *curr = mem, *prev = mem + nbase;
    for (size_t c = 0; c < C->nc; c++) {
        const float *move = C->f + c * C->stride + 2 * nbase, *stay = move + nbase;
        { double *tmp = curr; curr = prev; prev = tmp; }
        for (size_t b1 = 0; b1 < nbase; b1++) {
            curr[b1] = -HUGE_VAL;
            for (size_t b2 = 0; b2 < nbase; b2++)
                if (b1 != b2) curr[b1] = fo_logsumexp(curr[b1], prev[b2]);
            curr[b1] += move[b1];
        }
        for (size_t b = 0; b < nbase; b++) curr[b] = fo_logsumexp(curr[b], prev[b] + stay[b]);
    }
    double logZ = curr[0];
    for (size_t st = 1; st < nbase; st++) logZ = fo_logsumexp(logZ, curr[st]);
    return logZ;
}

/* layers.c:1197-1228 */
fo_mat *fo_globalnorm_runlength(const fo_mat *X, const fo_mat *W, const fo_mat *b, float temperature) {
    fo_mat *C = fo_affine_map(X, W, b);
    if (!C) return NULL;
    if (C->nr % 4 != 0) return fo_free_mat(C);
    const size_t nbase = C->nr / 4;
    for (size_t c = 0; c < C->nc; c++) {
        float *x = C->f + c * C->stride;
        for (size_t k = 0; k < nbase; k++) {
            x[k] = 1.0f + fo_softplusf(x[k]);
            x[nbase + k] = 1e-1f + fo_softplusf(x[nbase + k]);
            x[2 * nbase + k] = 5.0f * tanhf(x[2 * nbase + k]) / temperature;
            x[3 * nbase + k] = 5.0f * tanhf(x[3 * nbase + k]) / temperature;
        }
    }
    const float logZ = fo_runlength_partition_function(C) / (float)C->nc;
    for (size_t c = 0; c < C->nc; c++)
        for (size_t r = 2 * nbase; r < 4 * nbase; r++) C->f[c * C->stride + r] -= logZ;
    return C;
}

/* layers.c:1241-1246 */
static size_t rle_trans_lookup(size_t base_from, int stay_from, size_t base_to, int stay_to, size_t nbase) {
    (void)stay_to;
    return base_to * 2 * nbase + base_from + (stay_from ? nbase : 0);
}

/* layers.c:1255-1302  fp64 forward recursion; the stay states are combined with the FLOAT logsumexpf, as the reference does */
double fo_runlengthV2_partition_function(const fo_mat *C) {
    if (!C) return NAN;
    const size_t nbase = fo_nbase_from_nparam(C->nr), nstate = 2 * nbase;
    double mem[2 * 32] = { 0 };
    if (nstate > 32) return NAN;
    double *curr = mem, *prev = mem + nstate;
    for (size_t c = 0; c < C->nc; c++) {
        const float *S = C->f + c * C->stride + nstate;
        { double *tmp = curr; curr = prev; prev = tmp; }
        for (size_t b1 = 0; b1 < nbase; b1++) {
            curr[b1] = -HUGE_VAL;
            for (size_t b2 = 0; b2 < nbase; b2++) {
                if (b1 == b2) continue;
                curr[b1] = fo_logsumexp(curr[b1], prev[b2] + S[rle_trans_lookup(b2, 0, b1, 0, nbase)]);
                curr[b1] = fo_logsumexp(curr[b1], prev[b2 + nbase] + S[rle_trans_lookup(b2, 1, b1, 0, nbase)]);
            }
        }
        for (size_t b = 0; b < nbase; b++)
            curr[b + nbase] = fo_logsumexpf(prev[b] + S[rle_trans_lookup(b, 0, b, 1, nbase)],
                                            prev[b + nbase] + S[rle_trans_lookup(b, 1, b, 1, nbase)]);
    }
    double logZ = curr[0];
    for (size_t st = 1; st < nstate; st++) logZ = fo_logsumexp(logZ, curr[st]);
    return logZ;
}

/* layers.c:1325-1358 */
fo_mat *fo_globalnorm_runlengthV2(const fo_mat *X, const fo_mat *W, const fo_mat *b, float temperature) {
    fo_mat *C = fo_affine_map(X, W, b);
    if (!C) return NULL;
    const size_t nbase = fo_nbase_from_nparam(C->nr), nrunparam = 2 * nbase;
    for (size_t c = 0; c < C->nc; c++) {
        float *x = C->f + c * C->stride;
        for (size_t k = 0; k < nbase; k++) {
            x[k] = 1.0f + fo_softplusf(x[k]);
            x[nbase + k] = 1e-8f + fo_softplusf(x[nbase + k]);
        }
        for (size_t p = nrunparam; p < C->nr; p++) x[p] = 5.0f * tanhf(x[p]) / temperature;
    }
    const float logZ = fo_runlengthV2_partition_function(C) / (float)C->nc;
    for (size_t c = 0; c < C->nc; c++)
        for (size_t r = nrunparam; r < C->nr; r++) C->f[c * C->stride + r] -= logZ;
    return C;
}

/* decode.c:927-1013  Viterbi over move/stay states; path[blk] in [0, 2*nbase): < nbase = a new base */
float fo_decode_crf_runlength(const fo_mat *param, int *path) {
    if (!param || !path) return NAN;
    const size_t nblk = param->nc, nbase = fo_nbase_from_nparam(param->nr), nstate = 2 * nbase;
    float *mem = calloc(2 * nstate, sizeof(float));
    char *traceback = calloc(nstate * nblk, sizeof(char));
    if (!mem || !traceback) { free(mem); free(traceback); return NAN; }
    float *prev = mem, *curr = mem + nstate;
    for (size_t blk = 0; blk < nblk; blk++) {
        const float *S = param->f + blk * param->stride + nstate;
        char *tb = traceback + blk * nstate;
        { float *tmp = prev; prev = curr; curr = tmp; }
        for (size_t st = 0; st < nstate; st++) curr[st] = -HUGE_VAL;
        for (size_t b1 = 0; b1 < nbase; b1++)
            for (size_t b2 = 0; b2 < nbase; b2++) {
                if (b1 == b2) continue;
                const float move_score = prev[b2] + S[rle_trans_lookup(b2, 0, b1, 0, nbase)];
                if (move_score > curr[b1]) { curr[b1] = move_score; tb[b1] = (char)b2; }
                const float stay_score = prev[b2 + nbase] + S[rle_trans_lookup(b2, 1, b1, 0, nbase)];
                if (stay_score > curr[b1]) { curr[b1] = stay_score; tb[b1] = (char)(b2 + nbase); }
            }
        for (size_t b = 0; b < nbase; b++) {
            const float stay_score = prev[b + nbase] + S[rle_trans_lookup(b, 1, b, 1, nbase)];
            const float move_score = prev[b] + S[rle_trans_lookup(b, 0, b, 1, nbase)];
            if (stay_score > move_score) { curr[b + nbase] = stay_score; tb[b + nbase] = (char)(b + nbase); }
            else { curr[b + nbase] = move_score; tb[b + nbase] = (char)b; }
        }
    }
    size_t last_state = 0;                                   /* argmaxf: first maximum (util.c:17-31) */
    for (size_t st = 1; st < nstate; st++) if (curr[st] > curr[last_state]) last_state = st;
    const float logscore = curr[last_state];
    for (size_t blk = nblk; blk > 0; blk--) {
        const char state = traceback[(blk - 1) * nstate + last_state];
        path[blk - 1] = (int)last_state;
        last_state = (size_t)state;
    }
    free(traceback);
    free(mem);
    return logscore;
}

/* decode.c:1037-1159  transition posteriors (NOT normalised per block, unlike the flip-flop version); the shape
 * and scale rows are copied through */
fo_mat *fo_transpost_crf_runlength(const fo_mat *param) {
    if (!param) return NULL;
    const size_t nblk = param->nc, nparam = param->nr, nbase = fo_nbase_from_nparam(nparam), nstate = 2 * nbase;
    fo_mat *fwd = fo_make_mat(nstate, nblk + 1), *post = fo_make_mat(nparam, nblk);
    float *mem = calloc(2 * nstate, sizeof(float));
    if (!fwd || !post || !mem) { fo_free_mat(fwd); fo_free_mat(post); free(mem); return NULL; }
    for (size_t blk = 0; blk < nblk; blk++) {
        const float *S = param->f + blk * param->stride + nstate;
        const float *prev = fwd->f + blk * fwd->stride;
        float *curr = fwd->f + (blk + 1) * fwd->stride;
        for (size_t b1 = 0; b1 < nbase; b1++) {
            curr[b1] = -HUGE_VAL;
            for (size_t b2 = 0; b2 < nbase; b2++) {
                if (b1 == b2) continue;
                const float stay_score = prev[b2 + nbase] + S[rle_trans_lookup(b2, 1, b1, 0, nbase)];
                const float move_score = prev[b2] + S[rle_trans_lookup(b2, 0, b1, 0, nbase)];
                curr[b1] = fo_logsumexpf(curr[b1], fo_logsumexpf(stay_score, move_score));
            }
        }
        for (size_t b = 0; b < nbase; b++) {
            const float stay_score = prev[b + nbase] + S[rle_trans_lookup(b, 1, b, 1, nbase)];
            const float move_score = prev[b] + S[rle_trans_lookup(b, 0, b, 1, nbase)];
            curr[b + nbase] = fo_logsumexpf(stay_score, move_score);
        }
    }
    float *prev = mem, *curr = mem + nstate;
    for (size_t blk = nblk; blk > 0; blk--) {
        const float *F = fwd->f + (blk - 1) * fwd->stride;
        const float *S = param->f + (blk - 1) * param->stride + nstate;
        float *P = post->f + (blk - 1) * post->stride + nstate;
        { float *tmp = curr; curr = prev; prev = tmp; }
        for (size_t b1 = 0; b1 < nbase; b1++) {
            curr[b1] = -HUGE_VAL;
            curr[b1 + nbase] = -HUGE_VAL;
            for (size_t b2 = 0; b2 < nbase; b2++) {
                if (b1 == b2) continue;
                const size_t move_idx = rle_trans_lookup(b1, 0, b2, 0, nbase);
                curr[b1] = fo_logsumexpf(curr[b1], prev[b2] + S[move_idx]);
                P[move_idx] = F[b1] + prev[b2] + S[move_idx];
                const size_t stay_idx = rle_trans_lookup(b1, 1, b2, 0, nbase);
                curr[b1 + nbase] = fo_logsumexpf(curr[b1 + nbase], prev[b2] + S[stay_idx]);
                P[stay_idx] = F[b1 + nbase] + prev[b2] + S[stay_idx];
            }
        }
        for (size_t b = 0; b < nbase; b++) {
            const size_t idx = rle_trans_lookup(b, 0, b, 1, nbase);
            curr[b] = fo_logsumexpf(curr[b], prev[b + nbase] + S[idx]);
            P[idx] = F[b] + S[idx] + prev[b + nbase];
        }
        for (size_t b = 0; b < nbase; b++) {
            const size_t idx = rle_trans_lookup(b, 1, b, 1, nbase);
            curr[b + nbase] = fo_logsumexpf(curr[b + nbase], prev[b + nbase] + S[idx]);
            P[idx] = F[b + nbase] + S[idx] + prev[b + nbase];
        }
        for (size_t p = 0; p < nstate; p++) post->f[(blk - 1) * post->stride + p] = param->f[(blk - 1) * param->stride + p];
    }
    free(mem);
    fo_free_mat(fwd);
    return post;
}


/* ---- decoders of the FIRST run-length head (globalnorm_runlength's [4 nbase x nblk] output: two run parameters, a move weight and a stay
 * weight per base).  No registry entry reaches them; restated for the drop-in boundary's last five prototypes (decode.h:26-36). ---- */

/* decode.c:552-562  approximate mean of a discrete Weibull: sum_{i=1..maxval} exp(-(i / scale)^shape), float arithmetic, libm powf / expf */
float fo_dwmean(float shape, float scale, int maxval) {
    float m = 0.0f;
    for (int i = 1; i <= maxval; i++) m += expf(-powf((float)i / scale, shape));
    return m;
}

/* decode.c:576-603  runlength[blk] = 1 + round(dwmean of the entered base's shape / scale rows), 0 where path[blk] < 0; returns the sum */
size_t fo_runlengths_mean(const fo_mat *param, const int *path, int *runlength) {
    if (!param || !path || !runlength) return 0;
    const size_t nblk = param->nc, nbase = param->nr / 4;           /* nbase_from_runlength_nparam (layers.c:1115-1119) */
    size_t seqlen = 0;
    for (size_t blk = 0; blk < nblk; blk++) {
        runlength[blk] = 0;
        if (path[blk] < 0) continue;
        const float *col = param->f + blk * param->stride + path[blk];
        runlength[blk] = 1 + roundf(fo_dwmean(col[0], col[nbase], 100));      /* (float sum, converted on assignment, as written there) */
        seqlen += runlength[blk];
    }
    return seqlen;
}

/* decode.c:616-635 */
size_t fo_runlengths_unit(const fo_mat *param, const int *path, int *runlength) {
    if (!param || !path || !runlength) return 0;
    size_t seqlen = 0;
    for (size_t blk = 0; blk < param->nc; blk++) {
        runlength[blk] = (path[blk] < 0) ? 0 : 1;
        seqlen += runlength[blk];
    }
    return seqlen;
}

/* decode.c:646-672  caller frees */
char *fo_runlength_to_basecall(const int *path, const int *runlength, size_t nblk) {
    if (!path || !runlength) return NULL;
    static const char lookup[5] = { 'A', 'C', 'G', 'T', 'Z' };      /* decode.h:16 */
    int seqlen = 0;
    for (size_t blk = 0; blk < nblk; blk++) seqlen += runlength[blk];
    char *seq = calloc(seqlen + 1, sizeof(char));
    if (!seq) return NULL;
    size_t i = 0;
    for (size_t blk = 0; blk < nblk; blk++) {
        if (path[blk] < 0) continue;
        for (int rl = 0; rl < runlength[blk]; rl++) seq[i++] = lookup[path[blk]];
    }
    return seq;
}

/* decode.c:694-767  Viterbi.  A base is entered from the best OTHER base (the block's first maximum; for that base itself the first maximum
 * of the rest), or kept through its stay weight when that is strictly better.  path[blk] = entered base, -1 while staying. */
float fo_decode_runlength(const fo_mat *param, int *path) {
    if (!param || !path) return NAN;
    const size_t nblk = param->nc, nbase = param->nr / 4;
    float *mem = calloc(2 * nbase, sizeof(float));
    char *traceback = calloc(nbase * nblk, sizeof(char));
    if (!mem || !traceback) { free(mem); free(traceback); return NAN; }
    float *prev = mem, *curr = mem + nbase;
    for (size_t blk = 0; blk < nblk; blk++) {
        const float *move = param->f + blk * param->stride + 2 * nbase, *stay = move + nbase;
        char *tb = traceback + blk * nbase;
        { float *tmp = prev; prev = curr; curr = tmp; }
        size_t idx = 0;                                          /* argmaxf (util.c:17-31): first maximum */
        for (size_t i = 1; i < nbase; i++) if (prev[i] > prev[idx]) idx = i;
        const float max_score = prev[idx];
        prev[idx] = -HUGE_VAL;
        size_t idx2 = 0;
        for (size_t i = 1; i < nbase; i++) if (prev[i] > prev[idx2]) idx2 = i;
        prev[idx] = max_score;
        for (size_t b = 0; b < nbase; b++) { curr[b] = max_score; tb[b] = (char)idx; }
        curr[idx] = prev[idx2];
        tb[idx] = (char)idx2;
        for (size_t b = 0; b < nbase; b++) curr[b] += move[b];
        for (size_t b = 0; b < nbase; b++) {
            const float stay_score = prev[b] + stay[b];
            if (stay_score > curr[b]) { curr[b] = stay_score; tb[b] = (char)(b + nbase); }
        }
    }
    for (size_t blk = 0; blk < nblk; blk++) path[blk] = -1;
    size_t last_state = 0;
    for (size_t st = 1; st < nbase; st++) if (curr[st] > curr[last_state]) last_state = st;
    const float logscore = curr[last_state];
    for (size_t blk = nblk; blk > 0; blk--) {
        const char state = traceback[(blk - 1) * nbase + last_state];
        if ((size_t)state < nbase) { path[blk - 1] = (int)last_state; last_state = (size_t)state; }
    }
    free(traceback);
    free(mem);
    return logscore;
}

/* decode.c:793-892  log posteriors of the move and stay weights; [nparam x nblk + 1], everything else zero */
fo_mat *fo_posterior_runlength(const fo_mat *param) {
    if (!param) return NULL;
    const size_t nblk = param->nc, nparam = param->nr, nbase = nparam / 4;
    fo_mat *fwd = fo_make_mat(nbase, nblk + 1), *post = fo_make_mat(nparam, nblk + 1);
    float *mem = calloc(2 * nbase, sizeof(float));
    if (!fwd || !post || !mem) { fo_free_mat(fwd); fo_free_mat(post); free(mem); return NULL; }
    for (size_t blk = 0; blk < nblk; blk++) {                    /* forward */
        const float *move = param->f + blk * param->stride + 2 * nbase, *stay = move + nbase;
        const float *prev = fwd->f + blk * fwd->stride;
        float *curr = fwd->f + (blk + 1) * fwd->stride;
        for (size_t b1 = 0; b1 < nbase; b1++) {
            curr[b1] = -HUGE_VAL;
            for (size_t b2 = 0; b2 < nbase; b2++) if (b1 != b2) curr[b1] = fo_logsumexpf(curr[b1], prev[b2]);
            curr[b1] += move[b1];
        }
        for (size_t b = 0; b < nbase; b++) curr[b] = fo_logsumexpf(curr[b], prev[b] + stay[b]);
    }
    float *prev = mem, *curr = mem + nbase;
    for (size_t blk = nblk; blk > 0; blk--) {                    /* backward, with the posterior of block blk - 1 on the way */
        const float *move = param->f + (blk - 1) * param->stride + 2 * nbase, *stay = move + nbase;
        const float *f = fwd->f + (blk - 1) * fwd->stride;
        float *pm = post->f + (blk - 1) * post->stride + 2 * nbase, *ps = pm + nbase;
        { float *tmp = curr; curr = prev; prev = tmp; }
        for (size_t b1 = 0; b1 < nbase; b1++) {
            curr[b1] = -HUGE_VAL;
            pm[b1] = -HUGE_VAL;
            for (size_t b2 = 0; b2 < nbase; b2++) {
                if (b1 == b2) continue;
                curr[b1] = fo_logsumexpf(curr[b1], prev[b2] + move[b2]);
                pm[b1] = fo_logsumexpf(pm[b1], f[b2]);
            }
            pm[b1] += prev[b1] + move[b1];
        }
        for (size_t b = 0; b < nbase; b++) {
            curr[b] = fo_logsumexpf(curr[b], prev[b] + stay[b]);
            ps[b] = f[b] + stay[b] + prev[b];
        }
    }
    free(mem);
    fo_free_mat(fwd);
    return post;
}

/* runnie.c:282-313  one record per called base: (base index, block of the call, dwell); returns the count */
size_t fo_runlength_records(const int *path, size_t nblock, size_t nbase, int *base, int *block, int *dwell) {
    size_t n = 0;
    int d = 1, last_blk = -1;
    for (size_t blk = 0; blk < nblock; blk++) {
        if ((size_t)path[blk] >= nbase) { d += 1; continue; }
        if (last_blk >= 0) { base[n] = path[last_blk]; block[n] = last_blk; dwell[n] = d; n++; }
        last_blk = (int)blk;
        d = 1;
    }
    if (last_blk >= 0) { base[n] = path[last_blk]; block[n] = last_blk; dwell[n] = d; n++; }
    return n;
}
