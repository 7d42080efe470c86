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

static ffhip_mat mview(const_flappie_matrix m) {
    ffhip_mat v = { m->data.f, m->nr, m->nc, m->stride, (void **)&((flappie_matrix)m)->dev, (int *)&((flappie_matrix)m)->dev_state };
    return v;
}

/* decode.c:17-36 */
float argmax_decoder(const_flappie_matrix logpost, int *seq) {
    if (NULL == logpost || NULL == seq) return NAN;
    struct ffhip_engine *eng = flappie_hip_engine();
    float score = NAN;
    if (NULL == eng) return NAN;
    if (0 != ffhip_op_argmax_decoder(eng, mview(logpost), seq, &score)) { warnx("%s: %s", __func__, ffhip_last_error()); return NAN; }
    return score;
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
    if (0 != ffhip_op_viterbi(eng, mview(trans), combine_stays, path, qpath, &score)) {
        warnx("%s", ffhip_last_error());
        return NAN;
    }
    return score;
}

/* decode.c:209-270 */
float constrained_crf_flipflop(const_flappie_matrix post, int *path) {
    if (NULL == post || NULL == path) return NAN;
    struct ffhip_engine *eng = flappie_hip_engine();
    float score = NAN;
    if (NULL == eng) return NAN;
    if (0 != ffhip_op_constrained_flipflop(eng, mview(post), path, &score)) { warnx("%s: %s", __func__, ffhip_last_error()); return NAN; }
    return score;
}

/* decode.c:275-372 */
flappie_matrix posterior_crf_flipflop(const_flappie_matrix trans, bool return_log) {
    if (NULL == trans) return NULL;
    struct ffhip_engine *eng = flappie_hip_engine();
    if (NULL == eng) return NULL;
    const size_t nbase = nbase_from_flipflop_nparam(trans->nr);
    flappie_matrix fwd = make_flappie_matrix(2 * nbase, trans->nc + 1);
    if (NULL == fwd) return NULL;
    if (0 != ffhip_op_posterior_flipflop(eng, mview(trans), mview(fwd))) { warnx("%s: %s", __func__, ffhip_last_error()); return free_flappie_matrix(fwd); }
    if (!return_log) {                                   /* :365-368 */
        exp_activation_inplace(fwd);
        row_normalise_inplace(fwd);
    }
    return fwd;
}

/* decode.c:377-497 */
flappie_matrix transpost_crf_flipflop(const_flappie_matrix trans, bool return_log) {
    if (NULL == trans) return NULL;
    struct ffhip_engine *eng = flappie_hip_engine();
    if (NULL == eng) return NULL;
    flappie_matrix tpost = make_flappie_matrix(trans->nr, trans->nc);
    if (NULL == tpost) return NULL;
    if (0 != ffhip_op_transpost(eng, mview(trans), return_log, mview(tpost))) {      /* scores on the device: the posterior stays there too */
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
        0 != ffhip_op_trace(eng, mview(tpost), tmp)) {
        free(tmp);
        return free_flappie_imatrix(trace);
    }
    for (size_t c = 0; c <= tpost->nc; c++) memcpy(trace->data.f + c * trace->stride, tmp + c * nstate, nstate * sizeof(int32_t));
    free(tmp);
    return trace;
}

/* ---- run-length decoders (decode.c:927-1159) ---- */

float decode_crf_runlength(const_flappie_matrix param, int *path) {
    if (NULL == param || NULL == path) return NAN;
    struct ffhip_engine *eng = flappie_hip_engine();
    float score = NAN;
    if (NULL == eng) return NAN;
    if (0 != ffhip_runlength_viterbi(eng, mview(param), path, &score)) { warnx("%s: %s", __func__, ffhip_last_error()); return NAN; }
    return score;
}

flappie_matrix transpost_crf_runlength(const_flappie_matrix param) {
    if (NULL == param) return NULL;
    struct ffhip_engine *eng = flappie_hip_engine();
    if (NULL == eng) return NULL;
    flappie_matrix post = make_flappie_matrix(param->nr, param->nc);
    if (NULL == post) return NULL;
    if (0 != ffhip_runlength_transpost(eng, mview(param), mview(post))) { warnx("%s: %s", __func__, ffhip_last_error()); return free_flappie_matrix(post); }
    flappie_matrix_sync(post);          /* (runnie.c:294-307 reads this matrix's data.f directly: it is returned with a current host image) */
    return post;
}

/* ---- decoders of the first-generation run-length head (decode.c:552-892) ---- */

float decode_runlength(const_flappie_matrix param, int *path) {
    if (NULL == param || NULL == path) return NAN;
    struct ffhip_engine *eng = flappie_hip_engine();
    float score = NAN;
    if (NULL == eng) return NAN;
    if (0 != ffhip_runlength_v1_viterbi(eng, mview(param), path, &score)) { warnx("%s: %s", __func__, ffhip_last_error()); return NAN; }
    return score;
}

flappie_matrix posterior_runlength(const_flappie_matrix param) {
    if (NULL == param) return NULL;
    struct ffhip_engine *eng = flappie_hip_engine();
    if (NULL == eng) return NULL;
    flappie_matrix post = make_flappie_matrix(param->nr, param->nc + 1);
    if (NULL == post) return NULL;
    if (0 != ffhip_runlength_v1_posterior(eng, mview(param), mview(post))) { warnx("%s: %s", __func__, ffhip_last_error()); return free_flappie_matrix(post); }
    flappie_matrix_sync(post);
    return post;
}

size_t runlengths_mean(const_flappie_matrix param, const int *path, int *runlength) {
    if (NULL == param || NULL == path || NULL == runlength) return 0;
    struct ffhip_engine *eng = flappie_hip_engine();
    size_t seqlen = 0;
    if (NULL == eng) return 0;
    if (0 != ffhip_runlength_v1_mean(eng, mview(param), path, runlength, &seqlen)) { warnx("%s: %s", __func__, ffhip_last_error()); return 0; }
    return seqlen;
}

/* decode.c:616-635: bookkeeping on the caller's arrays, no arithmetic */
size_t runlengths_unit(const_flappie_matrix param, const int *path, int *runlength) {
    if (NULL == param || NULL == path || NULL == runlength) return 0;
    size_t seqlen = 0;
    for (size_t blk = 0; blk < param->nc; blk++) {
        runlength[blk] = (path[blk] >= 0) ? 1 : 0;
        seqlen += (size_t)runlength[blk];
    }
    return seqlen;
}

/* decode.c:646-672 */
char *runlength_to_basecall(const int *path, const int *runlength, size_t nblk) {
    if (NULL == path || NULL == runlength) return NULL;
    size_t seqlen = 0;
    for (size_t blk = 0; blk < nblk; blk++) seqlen += (size_t)runlength[blk];
    char *seq = calloc(seqlen + 1, sizeof(char));
    if (NULL == seq) return NULL;
    size_t at = 0;
    for (size_t blk = 0; blk < nblk; blk++) {
        if (path[blk] < 0) continue;
        memset(seq + at, base_lookup[path[blk]], (size_t)runlength[blk]);
        at += (size_t)runlength[blk];
    }
    return seq;
}
