/*  cpu_ref.c -- vectorised dot-product kernels for the oracle's "fast" mode.  TEST / BASELINE INFRASTRUCTURE.
 *
 *  The oracle proper (ff_oracle.c, dot mode 0) adds every dot product term by term in index order -- a definite order to
 *  hold the GPU against, and ~19x slower than the BLAS the reference links (~10 k samples/s/core at
 *  H = 384 under OpenBLAS).  bench.py's cpu_baseline leg needs a CPU number that stands for the reference's speed, not for
 *  a scalar loop, so dot mode 2 routes the same algorithm through the kernels below -- the shapes OpenBLAS would run:
 *
 *    fo_fast_gemv_t    y[f] += sum_i W[i + f*ldW] x[i]          (cblas_sgemv Trans: lstm_step layers.c:1003, grumod_step :697,
 *                                                                convolution edge windows :224,:268)
 *    fo_fast_gemm_tn   Y[:, c] += W^T X[:, c], 4 x 6 register    (cblas_sgemm: affine_map flappie_matrix.c:380, convolution body
 *                      tiles                                     layers.c:250)
 *
 *  Summation order (documented because it differs from mode 0, hence results differ in the last bits; tests/test_oracle_cpu.py
 *  bounds the difference): 16 interleaved partial sums over i (lane l takes i = l mod 16), reduced pairwise
 *  ((0+8)+(4+12))... as written in hsum16, the tail (len mod 16) added last in index order.  No FMA contraction
 *  (-ffp-contract=off in oracle/Makefile), as the reference build (-march=ivybridge has no FMA).
 *
 *  target_clones: the .so is built in one container and run on another host; the dynamic loader picks the widest of
 *  AVX-512 / AVX2 / baseline the CPU has.
 */
#include <stddef.h>
#include <string.h>

typedef float v16f __attribute__((vector_size(64), aligned(4)));

#define CLONES __attribute__((target_clones("avx512f", "avx2", "default")))

static inline v16f ld16(const float *p) {
    v16f v;
    memcpy(&v, p, sizeof v);
    return v;
}

static inline float hsum16(v16f a) {
    float t[16];
    memcpy(t, &a, sizeof t);
    const float s0 = (t[0] + t[8]) + (t[4] + t[12]);
    const float s1 = (t[1] + t[9]) + (t[5] + t[13]);
    const float s2 = (t[2] + t[10]) + (t[6] + t[14]);
    const float s3 = (t[3] + t[11]) + (t[7] + t[15]);
    return (s0 + s2) + (s1 + s3);
}

/* y[f] += W_f . x for f < nout; W_f = W + f*ldW, all of length len */
CLONES void fo_fast_gemv_t(float *y, const float *W, size_t ldW, size_t nout, size_t len, const float *x) {
    const size_t len16 = len & ~(size_t)15;
    size_t f = 0;
    for (; f + 4 <= nout; f += 4) {
        const float *w0 = W + f * ldW, *w1 = w0 + ldW, *w2 = w1 + ldW, *w3 = w2 + ldW;
        v16f a0 = { 0 }, a1 = { 0 }, a2 = { 0 }, a3 = { 0 };
        for (size_t i = 0; i < len16; i += 16) {
            const v16f xv = ld16(x + i);
            a0 += ld16(w0 + i) * xv;
            a1 += ld16(w1 + i) * xv;
            a2 += ld16(w2 + i) * xv;
            a3 += ld16(w3 + i) * xv;
        }
        float s0 = hsum16(a0), s1 = hsum16(a1), s2 = hsum16(a2), s3 = hsum16(a3);
        for (size_t i = len16; i < len; i++) {
            s0 += w0[i] * x[i];
            s1 += w1[i] * x[i];
            s2 += w2[i] * x[i];
            s3 += w3[i] * x[i];
        }
        y[f] += s0;
        y[f + 1] += s1;
        y[f + 2] += s2;
        y[f + 3] += s3;
    }
    for (; f < nout; f++) {
        const float *w = W + f * ldW;
        v16f a = { 0 };
        for (size_t i = 0; i < len16; i += 16) a += ld16(w + i) * ld16(x + i);
        float s = hsum16(a);
        for (size_t i = len16; i < len; i++) s += w[i] * x[i];
        y[f] += s;
    }
}

/* Y[f + c*ldY] += W_f . X_c for a 4-output x 6-column register tile: 24 accumulators, 4 + 6 loads per 24 multiply-adds */
CLONES static void gemm_tile_4x6(float *Y, size_t ldY, const float *W, size_t ldW, size_t len, const float *X, size_t ldX) {
    const size_t len16 = len & ~(size_t)15;
    v16f acc[4][6];
    for (int f = 0; f < 4; f++)
        for (int c = 0; c < 6; c++) acc[f][c] = (v16f){ 0 };
    for (size_t i = 0; i < len16; i += 16) {
        const v16f w0 = ld16(W + i), w1 = ld16(W + ldW + i), w2 = ld16(W + 2 * ldW + i), w3 = ld16(W + 3 * ldW + i);
#pragma GCC unroll 6
        for (int c = 0; c < 6; c++) {
            const v16f xv = ld16(X + c * ldX + i);
            acc[0][c] += w0 * xv;
            acc[1][c] += w1 * xv;
            acc[2][c] += w2 * xv;
            acc[3][c] += w3 * xv;
        }
    }
    for (int f = 0; f < 4; f++) {
        const float *w = W + f * ldW;
        for (int c = 0; c < 6; c++) {
            float s = hsum16(acc[f][c]);
            const float *xc = X + c * ldX;
            for (size_t i = len16; i < len; i++) s += w[i] * xc[i];
            Y[f + c * ldY] += s;
        }
    }
}

/* Y[:, c] += W^T X[:, c] for c < ncol: 4 x 6 tiles, the fringes through the gemv kernel (same summation order per element) */
void fo_fast_gemm_tn(float *Y, size_t ldY, const float *W, size_t ldW, size_t nout, size_t len,
                     const float *X, size_t ldX, size_t ncol) {
    const size_t nout4 = nout & ~(size_t)3;
    size_t c0 = 0;
    for (; c0 + 6 <= ncol; c0 += 6) {
        for (size_t f = 0; f < nout4; f += 4) gemm_tile_4x6(Y + f + c0 * ldY, ldY, W + f * ldW, ldW, len, X + c0 * ldX, ldX);
        if (nout4 < nout)
            for (size_t c = c0; c < c0 + 6; c++) fo_fast_gemv_t(Y + nout4 + c * ldY, W + nout4 * ldW, ldW, nout - nout4, len, X + c * ldX);
    }
    for (; c0 < ncol; c0++) fo_fast_gemv_t(Y + c0 * ldY, W, ldW, nout, len, X + c0 * ldX);
}

/* ---- gate arithmetic, vectorisable: the oracle's scalar functions (ff_oracle.c, restating sse_mathfun.h:225-301 and
 * util.h:329-337) written as straight-line code over arrays.  Same operations in the same order per element, so the
 * results are bit-identical to the scalar forms (no FMA contraction, no reassociation). */
static inline float exp_cephes(float x) {
    x = (x < 88.3762626647949f) ? x : 88.3762626647949f;
    x = (x > -88.3762626647949f) ? x : -88.3762626647949f;
    float fx = x * 1.44269504088896341f;
    fx = fx + 0.5f;
    float tmp = (float)(int)fx;
    fx = tmp - ((tmp > fx) ? 1.0f : 0.0f);
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
    union { int i; float f; } p;
    p.i = ((int)fx + 0x7f) << 23;
    return y * p.f;
}
static inline float logistic_(float x) { return 1.0f / (1.0f + exp_cephes(-x)); }
static inline float tanh_(float x) { const float y = logistic_(x + x); return (y + y) - 1.0f; }

/* lstm_step's element-wise half (layers.c:1008-1023): xF holds the four gate pre-activations i,f,g,o in blocks of `size` */
CLONES void fo_fast_lstm_gates(const float *xF, float *state, float *hout, size_t size) {
    for (size_t i = 0; i < size; i++) {
        const float forget = logistic_(xF[size + i]) * state[i];
        const float update = logistic_(xF[i]) * tanh_(xF[2 * size + i]);
        const float c = forget + update;
        state[i] = c;
        hout[i] = logistic_(xF[3 * size + i]) * tanh_(c);
    }
}

/* grumod_step's element-wise half (layers.c:699-712): xF = {z, r pre-activations, u = sW_c^T h}; x the projected input */
CLONES void fo_fast_grumod_gates(float *xF, const float *x, const float *hprev, float *hout, size_t size) {
    for (size_t i = 0; i < size; i++) {
        const float z = logistic_(xF[i]), r = logistic_(xF[size + i]);
        const float hbar = tanh_(r * xF[2 * size + i] + x[2 * size + i]);
        hout[i] = z * hprev[i] + (1.0f - z) * hbar;
    }
}
