/*  ref_inline.c -- array entry points onto the REFERENCE's header-only arithmetic.  TEST INFRASTRUCTURE.
 *
 *  Nothing of the reference is copied: this file #include's /root/reference/src/util.h (which pulls in
 *  sse_mathfun.h) where it lies (-I$(REF)/src in oracle/Makefile) and exports thin loops over the inline
 *  functions those headers define, so that tests can hold the oracle's scalar restatements (fo_expf_cephes,
 *  fo_logf_cephes, fo_logisticf, fo_tanhf, fo_eluf, fo_logsumexpf, fo_logsumexp, fo_phredf) -- and through them the
 *  GPU's device math -- to the reference's own object code bit for bit:
 *
 *      exp_ps / log_ps                       sse_mathfun.h:123-301
 *      expfv logfv logisticfv tanhfv elufv   util.h:319-346
 *      logsumexpf / logsumexp                util.h:276-282
 *      qscoref / phredf                      util.h:285-305
 *
 *  Built with the reference's own flags (CMakeLists.txt:115: -O3 -march=ivybridge, no FMA contraction possible on
 *  that target) into oracle/_ref/libflappie_inlref.so.  Only gcc and libm are needed.
 */
#include <stddef.h>
#include "util.h"

#define REF_V4_LOOP(name, fn)                                                    \
    void name(const float *in, float *out, size_t n4) {                          \
        for (size_t i = 0; i < n4; i++) {                                        \
            _mm_storeu_ps(out + 4 * i, fn(_mm_loadu_ps(in + 4 * i)));            \
        }                                                                        \
    }

/* n4 = number of 4-lane vectors; in/out hold 4*n4 floats */
REF_V4_LOOP(ref_expfv, expfv)
REF_V4_LOOP(ref_logfv, logfv)
REF_V4_LOOP(ref_logisticfv, logisticfv)
REF_V4_LOOP(ref_tanhfv, tanhfv)
REF_V4_LOOP(ref_elufv, elufv)

void ref_logsumexpf(const float *x, const float *y, float *out, size_t n) {
    for (size_t i = 0; i < n; i++) {
        out[i] = logsumexpf(x[i], y[i]);
    }
}

void ref_logsumexp(const double *x, const double *y, double *out, size_t n) {
    for (size_t i = 0; i < n; i++) {
        out[i] = logsumexp(x[i], y[i]);
    }
}

void ref_qscoref(const float *p, float *out, size_t n) {
    for (size_t i = 0; i < n; i++) {
        out[i] = qscoref(p[i]);
    }
}

void ref_phredf(const float *p, char *out, size_t n) {
    for (size_t i = 0; i < n; i++) {
        out[i] = phredf(p[i]);
    }
}
