/*  decode.h -- flip-flop decoding entry points of the drop-in boundary.
 *  Same signatures and semantics as /root/reference/src/decode.h:16-38: the flip-flop functions, runnie's run-length
 *  decoders and the first-generation run-length decoders no registry entry reaches.
 */
#ifndef FFHIP_DECODE_H
#define FFHIP_DECODE_H
#include <stdbool.h>
#include "flappie_matrix.h"
#include "layers.h"      /* exp_activation_inplace, nbase_from_flipflop_nparam (flappie.c:299, decode.c:131) */
#include "flappie_structures.h"

#ifdef __cplusplus
extern "C" {
#endif

/* decode.h:16-19 */
static const char base_lookup[5] = { 'A', 'C', 'G', 'T', 'Z' };
static inline char basechar(int b) { return base_lookup[b]; }

/* decode.c:17-36: per block the row of the maximum (-1 for the last row, the blank); returns the sum of the maxima, NAN on NULL */
float argmax_decoder(const_flappie_matrix logpost, int *seq);
/* decode.c:39-63: NUL-terminated string owned by the caller, NULL on failure */
char *collapse_repeats(int const *path, size_t npos, int modbase);
/* decode.c:66-79: positions pos in [1, npos) with path[pos] != path[pos-1]; 0 if an argument is NULL */
size_t change_positions(int const *path, size_t npos, int *chpos);

/* decode.c:119-204: Viterbi; path needs nblock+1 ints, qpath nblock+1 floats (qpath[0] = NAN);
 * returns the best score or NAN on failure */
float decode_crf_flipflop(const_flappie_matrix trans, bool combine_stays, int *path, float *qpath);
/* decode.c:209-270: Viterbi over per-state scores [nstate x nblock] under the flip-flop transition constraint; path needs
 * nblock+1 ints; returns the best score, NAN on failure */
float constrained_crf_flipflop(const_flappie_matrix post, int *path);
/* decode.c:275-372: per-state posteriors [nstate x nblock+1] (forward + backward); probabilities normalised per column when
 * return_log is false */
flappie_matrix posterior_crf_flipflop(const_flappie_matrix trans, bool return_log);
/* decode.c:377-497: forward/backward transition posteriors, log-normalised per block */
flappie_matrix transpost_crf_flipflop(const_flappie_matrix trans, bool return_log);
/* decode.c:499-543: tpost holds probabilities; returns [nstate x nblock+1] int32 */
flappie_imatrix trace_from_posterior(flappie_matrix tpost);

/* ---- run-length model (runnie) ----
 * decode.c:927-1013: Viterbi over nbase move + nbase stay states; path needs nblock ints (states 0..2*nbase-1,
 * < nbase marks a newly emitted base); returns the best score or NAN */
float decode_crf_runlength(const_flappie_matrix transparam, int *path);
/* decode.c:1037-1159: transition posteriors (not normalised per block); shape/scale rows copied through */
flappie_matrix transpost_crf_runlength(const_flappie_matrix trans);

/* ---- decoders of the first-generation run-length head (param: [4 nbase x nblock], globalnorm_runlength's output) ----
 * decode.c:694-767: Viterbi; path needs nblock ints: the base entered in a block, -1 while staying; returns the best score or NAN */
float decode_runlength(const_flappie_matrix param, int *path);
/* decode.c:576-603: runlength[blk] = 1 + round(mean of the entered base's discrete Weibull), 0 where path is -1; returns the sum */
size_t runlengths_mean(const_flappie_matrix param, const int *path, int *runlength);
/* decode.c:616-635: runlength[blk] = 1 where a base is entered; returns their number */
size_t runlengths_unit(const_flappie_matrix param, const int *path, int *runlength);
/* decode.c:646-672: the base of every entered block repeated runlength[blk] times; NUL-terminated, owned by the caller */
char *runlength_to_basecall(const int *path, const int *runlength, size_t nblk);
/* decode.c:793-892: log posteriors of the move and stay weights, [4 nbase x nblock + 1] (other entries zero) */
flappie_matrix posterior_runlength(const_flappie_matrix param);


#ifdef __cplusplus
}
#endif
#endif
