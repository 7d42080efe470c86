/*  ref_decode_glue.c -- the nine symbols the REFERENCE's decode.c needs from translation units that cannot be
 *  compiled in this image.  TEST INFRASTRUCTURE.
 *
 *  /root/reference/src/decode.c (Viterbi, forward/backward posterior, trace, change positions, the run-length
 *  decoders) and util.c (argmaxf / valmaxf) compile from their own sources with gcc + libm alone, and
 *  oracle/Makefile builds them where they lie into oracle/_ref/libflappie_decref.so.  decode.c however calls
 *
 *      make_flappie_matrix make_flappie_imatrix free_flappie_matrix free_flappie_imatrix
 *      row_normalise_inplace log_row_normalise_inplace                      (flappie_matrix.c)
 *      exp_activation_inplace nbase_from_flipflop_nparam
 *      nbase_from_runlength_nparam nbase_from_crf_runlength_nparam          (layers.c)
 *
 *  and flappie_matrix.c / layers.c #include <cblas.h>, which this image does not have -- those two files are
 *  unbuildable here and no stand-in header is used.  This file supplies the ten functions instead, in own words,
 *  against the reference's own type declarations (flappie_matrix.h is #include'd from $(REF)/src).  They are
 *  allocation, a normalisation loop and a size formula; what arithmetic they contain goes through the reference's
 *  header-inline logsumexpf / expfv (util.h), i.e. is the reference's code again:
 *
 *      make/free              zero-filled, 16-byte aligned, rows padded to 4     flappie_matrix.c:20-61,94-140,142-155
 *      row_normalise_inplace      column sum over rows < nr, multiply by 1/sum    flappie_matrix.c:425-447 (known answers
 *                                 of test_flappie_matrix.c:32-46 are asserted for it in tests/test_ref_pins.py)
 *      log_row_normalise_inplace  sequential logsumexpf chain, subtract           flappie_matrix.c:450-467
 *      exp_activation_inplace     expfv over every 4-lane group, pads included    layers.c:56-66
 *      nbase_from_*               (sqrt(1+2n)-1)/2 rounded; n/4                   layers.c:1029-1032,1115-1119,1235-1239
 *
 *  STATUS this gives the decode rows: the restatement in ff_oracle.c is compared bit for bit with the reference's
 *  compiled decode.c (tests/test_ref_pins.py); the glue above is the only part that is not reference object code.
 *  DESIGN.md section 2 states this as "pinned through a partial reference build with declared glue".
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "flappie_matrix.h"
#include "util.h"

static void *zeroed_quads(size_t nquad) {
    void *p = NULL;
    if (nquad == 0 || nquad > ((size_t)-1) / 16 || 0 != posix_memalign(&p, 16, nquad * 16)) {
        return NULL;
    }
    memset(p, 0, nquad * 16);
    return p;
}

flappie_matrix make_flappie_matrix(size_t nr, size_t nc) {
    flappie_matrix m = malloc(sizeof(*m));
    if (NULL == m) {
        return NULL;
    }
    m->nr = nr;
    m->nrq = (nr + 3) / 4;
    m->nc = nc;
    m->stride = 4 * m->nrq;
    m->data.v = zeroed_quads(m->nrq * nc);
    if (NULL == m->data.v) {
        free(m);
        return NULL;
    }
    return m;
}

flappie_imatrix make_flappie_imatrix(size_t nr, size_t nc) {
    flappie_imatrix m = malloc(sizeof(*m));
    if (NULL == m) {
        return NULL;
    }
    m->nr = nr;
    m->nrq = (nr + 3) / 4;
    m->nc = nc;
    m->stride = 4 * m->nrq;
    m->data.v = zeroed_quads(m->nrq * nc);
    if (NULL == m->data.v) {
        free(m);
        return NULL;
    }
    return m;
}

flappie_matrix free_flappie_matrix(flappie_matrix m) {
    if (m) {
        free(m->data.v);
        free(m);
    }
    return NULL;
}

flappie_imatrix free_flappie_imatrix(flappie_imatrix m) {
    if (m) {
        free(m->data.v);
        free(m);
    }
    return NULL;
}

/* The reference sums the 4-lane groups lane-wise, removes the pad lanes of the last group and adds the four lane
 * sums horizontally as (l0 + l1) + (l2 + l3) -- two hadd_ps (flappie_matrix.c:431-441); then multiplies by the
 * reciprocal. */
void row_normalise_inplace(flappie_matrix C) {
    if (NULL == C) {
        return;
    }
    for (size_t col = 0; col < C->nc; col++) {
        float *x = C->data.f + col * C->stride;
        float lane[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
        for (size_t q = 0; q < C->nrq; q++) {
            for (int l = 0; l < 4; l++) {
                const float v = x[4 * q + l];
                lane[l] = (q == 0) ? v : lane[l] + v;
            }
        }
        for (size_t r = C->nr; r < C->stride; r++) {
            lane[r % 4] -= x[r];
        }
        const float total = (lane[0] + lane[1]) + (lane[2] + lane[3]);
        const float recip = 1.0f / total;
        for (size_t r = 0; r < C->stride; r++) {
            x[r] *= recip;
        }
    }
}

void log_row_normalise_inplace(flappie_matrix C) {
    if (NULL == C) {
        return;
    }
    for (size_t col = 0; col < C->nc; col++) {
        float *x = C->data.f + col * C->stride;
        float acc = x[0];
        for (size_t r = 1; r < C->nr; r++) {
            acc = logsumexpf(acc, x[r]);
        }
        for (size_t r = 0; r < C->nr; r++) {
            x[r] -= acc;
        }
    }
}

void exp_activation_inplace(flappie_matrix C) {
    if (NULL == C) {
        return;
    }
    for (size_t i = 0; i < C->nrq * C->nc; i++) {
        C->data.v[i] = expfv(C->data.v[i]);
    }
}

size_t nbase_from_flipflop_nparam(size_t nparam) {
    return (size_t)roundf((sqrtf((float)(1 + 2 * nparam)) - 1.0f) / 2.0f);
}

size_t nbase_from_runlength_nparam(size_t nparam) {
    return nparam / 4;
}

size_t nbase_from_crf_runlength_nparam(size_t nparam) {
    return nbase_from_flipflop_nparam(nparam);
}
