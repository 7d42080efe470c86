/*  signal_prep.c -- signal preparation (include/flappie_common.h) over the HIP engine.
 *  Names, arguments and failure behaviour of /root/reference/src/util.c:100-223,416-438 and
 *  flappie_common.c:13-81; the order statistics, trimming decision and normalisation run on the GPU
 *  (ffhip_prep.hip: exact selection instead of qsort).  Without a usable device the functions warn and
 *  return NAN / a zeroed table.
 */
#include <err.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/ffhip.h"
#include "../../include/flappie_common.h"
#include "../../include/networks.h"

void quantilef(const float *x, size_t nx, float *p, size_t np) {
    if (NULL == p) return;
    struct ffhip_engine *eng = (NULL != x) ? flappie_hip_engine() : NULL;
    if (NULL == eng || 0 != ffhip_quantiles(eng, x, nx, p, np)) {
        if (NULL != eng) warnx("quantilef: %s", ffhip_last_error());
        for (size_t i = 0; i < np; i++) p[i] = NAN;
    }
}

float medianf(const float *x, size_t n) {
    float p = 0.5;
    quantilef(x, n, &p, 1);
    return p;
}

float madf(const float *x, size_t n, const float *med) {
    if (NULL == x) return NAN;
    if (1 == n) return 0.0f;
    struct ffhip_engine *eng = flappie_hip_engine();
    float mad = NAN;
    if (NULL == eng) return NAN;
    if (0 != ffhip_mad(eng, x, n, med, &mad)) { warnx("madf: %s", ffhip_last_error()); return NAN; }
    return mad;
}

void medmad_normalise_array(float *x, size_t n) {
    if (NULL == x) return;
    if (1 == n) { x[0] = 0.0; return; }
    struct ffhip_engine *eng = flappie_hip_engine();
    if (NULL == eng) return;
    if (0 != ffhip_medmad_normalise(eng, x, n, NULL, NULL)) warnx("medmad_normalise_array: %s", ffhip_last_error());
}

void shift_scale_array(float *x, size_t n, float shift, float scale) {
    if (NULL == x || 0 == n) return;
    struct ffhip_engine *eng = flappie_hip_engine();
    if (NULL != eng && 0 != ffhip_array_transform(eng, x, n, FFHIP_PREP_SHIFT_SCALE, shift, scale)) warnx("shift_scale_array: %s", ffhip_last_error());
}

void difference_array(float *x, size_t n) {
    if (NULL == x || 0 == n) return;
    struct ffhip_engine *eng = flappie_hip_engine();
    if (NULL != eng && 0 != ffhip_array_transform(eng, x, n, FFHIP_PREP_DIFFERENCE, 0.0f, 1.0f)) warnx("difference_array: %s", ffhip_last_error());
}

/* util.c:429-438: string bookkeeping */
void reverse_char_array(char *x, size_t n) {
    if (NULL == x || 0 == n) return;
    for (size_t i = 0; i < n / 2; i++) {
        const size_t ri = n - i - 1;
        const char tmp = x[i];
        x[i] = x[ri];
        x[ri] = tmp;
    }
}

/* ranges from one GPU pass over one read; samples are not modified */
static int trim_ranges(const raw_table *rt, size_t trim_start, size_t trim_end, size_t chunk, float perc, size_t *start, size_t *end) {
    struct ffhip_engine *eng = flappie_hip_engine();
    if (NULL == eng) return -1;
    ffhip_prep *p = ffhip_prep_create(eng, rt, 1, trim_start, trim_end, chunk, perc, FFHIP_PREP_NONE, 0.0f);
    if (NULL == p) { warnx("signal trimming: %s", ffhip_last_error()); return -1; }
    const int rc = ffhip_prep_range(p, 0, start, end);
    ffhip_prep_destroy(p);
    return rc;
}

raw_table trim_raw_by_mad(raw_table rt, size_t chunk_size, float perc) {
    if (NULL == rt.raw) return (raw_table){ 0 };
    size_t start = 0, end = 0;
    if (0 != trim_ranges(&rt, 0, 0, chunk_size, perc, &start, &end)) return (raw_table){ 0 };
    rt.start = start;
    rt.end = end;
    return rt;
}

raw_table trim_and_segment_raw(raw_table rt, size_t trim_start, size_t trim_end, size_t varseg_chunk, float varseg_thresh) {
    if (NULL == rt.raw) return (raw_table){ 0 };
    size_t start = 0, end = 0;
    if (0 != trim_ranges(&rt, trim_start, trim_end, varseg_chunk, varseg_thresh, &start, &end) || start >= end) {
        free(rt.raw);
        return (raw_table){ 0 };
    }
    rt.start = start;
    rt.end = end;
    return rt;
}
