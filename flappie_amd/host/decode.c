/*  decode.c -- flip-flop decoding wrappers (include/decode.h) over the HIP engine.
 *  Signatures and return conventions of /root/reference/src/decode.c:39-79,119-204,377-543.
 */
#include <err.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/decode.h"
#include "../../include/networks.h"
#include "../../include/ffhip.h"

/* layers.c:1029-1032 */
size_t nbase_from_flipflop_nparam(size_t nparam) {
    return (size_t)roundf((-1.0f + sqrtf(1 + 2 * nparam)) / 2.0f);
}

/* decode.c:39-63 */
char *collapse_repeats(int const *path, size_t npos, int modbase) {
    if (NULL == path || modbase <= 0 || 0 == npos) return NULL;
    size_t nbase = 1;
    for (size_t pos = 1; pos < npos; pos++) if (path[pos] != path[pos - 1]) nbase += 1;
    char *basecall = calloc(nbase + 1, sizeof(char));
    if (NULL == basecall) return NULL;
    basecall[0] = base_lookup[path[0] % modbase];
    for (size_t pos = 1, bpos = 1; pos < npos; pos++)
        if (path[pos] != path[pos - 1]) basecall[bpos++] = basechar(path[pos] % modbase);
    return basecall;
}

/* decode.c:66-79 */
size_t change_positions(int const *path, size_t npos, int *chpos) {
    if (NULL == path || NULL == chpos) return 0;
    size_t nch = 0;
    for (size_t pos = 1; pos < npos; pos++) {
        if (path[pos] == path[pos - 1]) continue;
        chpos[nch++] = (int)pos;
    }
    return nch;
}

/* decode.c:119-204 */
float decode_crf_flipflop(const_flappie_matrix trans, bool combine_stays, int *path, float *qpath) {
    if (NULL == trans || NULL == path || NULL == qpath) return NAN;
    struct ffhip_engine *eng = flappie_hip_engine();
    if (NULL == eng) return NAN;
    float score = NAN;
    if (0 != ffhip_viterbi(eng, trans->data.f, trans->nc, trans->nr, trans->stride, combine_stays, path, qpath, &score)) {
        warnx("%s", ffhip_last_error());
        return NAN;
    }
    return score;
}

/* decode.c:377-497 */
flappie_matrix transpost_crf_flipflop(const_flappie_matrix trans, bool return_log) {
    if (NULL == trans) return NULL;
    struct ffhip_engine *eng = flappie_hip_engine();
    if (NULL == eng) return NULL;
    flappie_matrix tpost = make_flappie_matrix(trans->nr, trans->nc);
    if (NULL == tpost) return NULL;
    if (0 != ffhip_transpost(eng, trans->data.f, trans->nc, trans->nr, trans->stride, return_log, tpost->data.f)) {
        warnx("%s", ffhip_last_error());
        return free_flappie_matrix(tpost);
    }
    return tpost;
}

/* decode.c:499-543 */
flappie_imatrix trace_from_posterior(flappie_matrix tpost) {
    if (NULL == tpost) return NULL;
    struct ffhip_engine *eng = flappie_hip_engine();
    if (NULL == eng) return NULL;
    const size_t nbase = nbase_from_flipflop_nparam(tpost->nr), nstate = 2 * nbase;
    flappie_imatrix trace = make_flappie_imatrix(nstate, tpost->nc + 1);
    int32_t *tmp = malloc((tpost->nc + 1) * nstate * sizeof(int32_t));
    if (NULL == trace || NULL == tmp ||
        0 != ffhip_trace(eng, tpost->data.f, tpost->nc, tpost->nr, tpost->stride, tmp)) {
        free(tmp);
        return free_flappie_imatrix(trace);
    }
    for (size_t c = 0; c <= tpost->nc; c++) memcpy(trace->data.f + c * trace->stride, tmp + c * nstate, nstate * sizeof(int32_t));
    free(tmp);
    return trace;
}

/* layers.c:56-66 with the cephes exp of sse_mathfun.h:225-301, every stored element (pads too) */
static float exp_cephes_host(float x) {
    x = (x < 88.3762626647949f) ? x : 88.3762626647949f;
    x = (x > -88.3762626647949f) ? x : -88.3762626647949f;
    float fx = x * 1.44269504088896341f + 0.5f;
    float tmp = (float)(int)fx;
    fx = tmp - ((tmp > fx) ? 1.0f : 0.0f);
    x = x - fx * 0.693359375f;
    x = x - fx * -2.12194440e-4f;
    const float z = x * x;
    float y = 1.9875691500E-4f;
    y = y * x + 1.3981999507E-3f;
    y = y * x + 8.3334519073E-3f;
    y = y * x + 4.1665795894E-2f;
    y = y * x + 1.6666665459E-1f;
    y = y * x + 5.0000001201E-1f;
    y = y * z + x + 1.0f;
    union { int i; float f; } p2 = { .i = ((int)fx + 0x7f) << 23 };
    return y * p2.f;
}

void exp_activation_inplace(flappie_matrix C) {
    if (NULL == C) return;
    const size_t n = C->stride * C->nc;
    for (size_t i = 0; i < n; i++) C->data.f[i] = exp_cephes_host(C->data.f[i]);
}
