/*  signal_prep.c -- host signal preparation (include/flappie_common.h).
 *  Behaviour of /root/reference/src/util.c:74-223,278-287,416-427 and flappie_common.c:13-81.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/flappie_common.h"

/* util.c:74-80: equal elements compare as "less", like the reference */
static int floatcmp(const void *x, const void *y) {
    const float d = *(const float *)x - *(const float *)y;
    return (d > 0) ? 1 : -1;
}

void quantilef(const float *x, size_t nx, float *p, size_t np) {
    if (NULL == p) return;
    float *space = (NULL == x) ? NULL : malloc(nx * sizeof(float));
    if (NULL == space) {
        for (size_t i = 0; i < np; i++) p[i] = NAN;
        return;
    }
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

float medianf(const float *x, size_t n) {
    float p = 0.5;
    quantilef(x, n, &p, 1);
    return p;
}

float madf(const float *x, size_t n, const float *med) {
    const float mad_scaling_factor = 1.4826;
    if (NULL == x) return NAN;
    if (1 == n) return 0.0f;
    float *absdiff = malloc(n * sizeof(float));
    if (NULL == absdiff) return NAN;
    const float _med = (NULL == med) ? medianf(x, n) : *med;
    for (size_t i = 0; i < n; i++) absdiff[i] = fabsf(x[i] - _med);
    const float mad = medianf(absdiff, n);
    free(absdiff);
    return mad * mad_scaling_factor;
}

void medmad_normalise_array(float *x, size_t n) {
    if (NULL == x) return;
    if (1 == n) { x[0] = 0.0; return; }
    const float xmed = medianf(x, n);
    const float xmad = madf(x, n, &xmed);
    for (size_t i = 0; i < n; i++) x[i] = (x[i] - xmed) / xmad;
}

void shift_scale_array(float *x, size_t n, float shift, float scale) {
    if (NULL == x) return;
    for (size_t i = 0; i < n; i++) x[i] = (x[i] - shift) / scale;
}

void difference_array(float *x, size_t n) {
    if (NULL == x || 0 == n) return;
    for (size_t i = 1; i < n; i++) x[i - 1] = x[i] - x[i - 1];
    x[n - 1] = 0.0f;
}

void reverse_char_array(char *x, size_t n) {
    if (NULL == x || 0 == n) return;
    for (size_t i = 0; i < n / 2; i++) {
        const size_t ri = n - i - 1;
        const char tmp = x[i];
        x[i] = x[ri];
        x[ri] = tmp;
    }
}

raw_table trim_raw_by_mad(raw_table rt, size_t chunk_size, float perc) {
    const size_t nsample = rt.end - rt.start;
    const size_t nchunk = nsample / chunk_size;
    rt.end = nchunk * chunk_size;                    /* truncation, flappie_common.c:53-54 */
    float *madarr = malloc((nchunk ? nchunk : 1) * sizeof(float));
    if (NULL == madarr) return (raw_table){ 0 };
    for (size_t i = 0; i < nchunk; i++) madarr[i] = madf(rt.raw + rt.start + i * chunk_size, chunk_size, NULL);
    quantilef(madarr, nchunk, &perc, 1);
    const float thresh = perc;
    for (size_t i = 0; i < nchunk; i++) {
        if (madarr[i] > thresh) break;
        rt.start += chunk_size;
    }
    for (size_t i = nchunk; i > 0; i--) {
        if (madarr[i - 1] > thresh) break;
        rt.end -= chunk_size;
    }
    free(madarr);
    return rt;
}

raw_table trim_and_segment_raw(raw_table rt, size_t trim_start, size_t trim_end, size_t varseg_chunk, float varseg_thresh) {
    if (NULL == rt.raw) return (raw_table){ 0 };
    rt = trim_raw_by_mad(rt, varseg_chunk, varseg_thresh);
    if (NULL == rt.raw) return (raw_table){ 0 };
    rt.start = (rt.n - rt.start) > trim_start ? rt.start + trim_start : rt.n;
    rt.end = (rt.end > trim_end) ? rt.end - trim_end : 0;
    if (rt.start >= rt.end) {
        free(rt.raw);
        return (raw_table){ 0 };
    }
    return rt;
}
