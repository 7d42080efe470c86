/*  ffhip.h -- the C-ABI between flappie's C host code and the MI355X (gfx950) HIP engine.
 *
 *  This is the thin shim the north-star asks for: plain C, plain pointers and sizes, no C++ or
 *  torch types.  It is what the reference's host code would bind instead of calling into
 *  layers.c/decode.c one read at a time.  Each entry point names the reference interface it
 *  replaces (paths relative to /root/reference/src).
 *
 *  The reference has no batching (flappie.c:364-385 loops over files; one read = one forward
 *  pass).  A GPU cannot be filled one read at a time, so the unit of work here is a BATCH of reads
 *  that share one trimmed length (SURVEY.md section 8b "what the replacement must add").  Reads are
 *  never split or padded in time: the CRF normaliser and the recurrent state are whole-read
 *  quantities (layers.c:1089, :902-916), so results are identical to per-read evaluation.
 *
 *  Error convention (flappie_stdlib.h:37-45 RETURN_NULL_IF): functions returning pointers return
 *  NULL on failure; functions returning int return 0 on success and a negative FFHIP_E* code
 *  otherwise; nothing throws, nothing aborts.  ffhip_last_error() gives a static message.
 *
 *  Threading: an engine and everything created from it belong to one host thread (the reference
 *  is single threaded).  Use one engine per GPU.
 */
#ifndef FFHIP_H
#define FFHIP_H

#include <stddef.h>
#include <stdint.h>
#include "flappie_matrix.h"
#include "flappie_structures.h"

#ifdef __cplusplus
extern "C" {
#endif

#define FFHIP_OK          0
#define FFHIP_EINVAL     -1   /* bad argument / unsupported model shape */
#define FFHIP_ENOMEM     -2   /* host or device allocation failed      */
#define FFHIP_EHIP       -3   /* a HIP runtime call failed             */
#define FFHIP_ENODEV     -4   /* no usable gfx950 device               */
#define FFHIP_ETIMEOUT   -5   /* an in-kernel wait gave up (persistent recurrent kernel) */

typedef struct ffhip_engine ffhip_engine;
typedef struct ffhip_model ffhip_model;
typedef struct ffhip_batch ffhip_batch;

enum ffhip_net_kind {
    FFHIP_NET_LSTM5 = 0,    /* flipflop5_guppy_transitions, networks.c:539-586 */
    FFHIP_NET_GRUMOD5 = 1,  /* flipflop_guppy_transitions,  networks.c:450-489 */
    FFHIP_NET_LSTM5_RLE = 2 /* runlength5_guppy_transitions, networks.c:672-725: LSTM5 trunk, globalnorm_runlengthV2 head;
                             * decode = transpost_crf_runlength + decode_crf_runlength (runnie.c:262-276): results are the
                             * path (ffhip_batch_get_path, states 0..2*nbase-1), score and parameter/posterior matrices */
};

/* Host-side weight bundle.  Mirrors `guppy_stride5_model` (networks.c:181-215) and `guppy_model`
 * (networks.c:150-178): pointers to matrices in the .mdl layout, never owned by the engine. */
typedef struct {
    int kind;                         /* enum ffhip_net_kind */
    int nconv;                        /* 3 (LSTM5: swish after each) or 1 (GRUMOD5: tanh) */
    const_flappie_matrix conv_W[3];
    const_flappie_matrix conv_b[3];
    int conv_stride[3];
    const_flappie_matrix rnn_iW[5];   /* layer order B1,F2,B3,F4,B5 */
    const_flappie_matrix rnn_sW[5];
    const_flappie_matrix rnn_b[5];
    const_flappie_matrix FF_W;
    const_flappie_matrix FF_b;
} ffhip_model_desc;

/* flags for ffhip_batch_run */
#define FFHIP_RUN_VITERBI_ONLY   1u   /* `--viterbi`: decode the transitions, skip fwd/bwd (flappie.c:277-282) */
#define FFHIP_RUN_NO_TRACE       2u   /* skip exp + trace_from_posterior (flappie.c:299-300)                  */
#define FFHIP_RUN_NO_DECODE      4u   /* stop after calculate_transitions (networks.c:108-111)                */
#define FFHIP_RUN_STEPWISE_RNN   8u   /* force the launch-per-step recurrent kernels (debug / cross-check)    */
#define FFHIP_RUN_UNFUSED_RNN   32u   /* separate input-projection GEMM + recurrent kernel (cross-check)      */
#define FFHIP_RUN_F32_RNN       64u   /* f32-input MFMA recurrent kernel instead of the split-operand (two fp16 slices) one (cross-check) */
#define FFHIP_RUN_KEEP_ACTS     16u   /* keep every layer's activations for ffhip_batch_get_activation        */
/* (A read that left the default kernels' operand range and was evaluated again on the f32 kernels -- ffhip_batch_f32_reruns() -- returns
 * the f32 run's scores, path and calls; its KEPT ACTIVATIONS are not refreshed: they stay those of the first, discarded evaluation.) */
/* Gate activations of the split layer kernels (logistic, tanh; layers.c:979-1026 through util.h:319-337, sse_mathfun.h:225-301).  Since round 6 the DEFAULT is
 * the hardware form of FFHIP_RUN_FAST_GATES2: decided by measurement (profiles/r06_gates_*.txt: 8192 reads at the headline shape, 2048 at the two H = 256 shapes --
 * worst |dtrans| against the oracle 2.1e-5 with either form, reads called apart from the oracle 5 against the exact form's 6, and two evaluations of the reference's
 * own algorithm in two summation orders part on 8) and worth +4.4 % (112.6 against 107.8 Msamples/s on one box).  FFHIP_RUN_EXACT_GATES, or FFHIP_FAST_GATES=0 in
 * the environment, brings back the operation-for-operation replay of the reference's exp_ps and division. */
#define FFHIP_RUN_FAST_GATES   128u   /* gate activations through v_exp_f32 / v_rcp_f32 as they are (1 ulp each, the exponent rounded once) */
#define FFHIP_RUN_FAST_GATES2  256u   /* the same with the exponent of v_exp_f32 carried in two words and a Newton step behind v_rcp_f32: exp and the
                                       * reciprocal to ~1 ulp at every argument (the reference's cephes replay is no closer to the true functions); the default */
#define FFHIP_RUN_EXACT_GATES  512u   /* the reference's exp_ps polynomial and its division replayed bit for bit (the default of rounds 1-5) */

const char *ffhip_last_error(void);
const char *ffhip_version(void);

/* ---- engine: one per GPU ------------------------------------------------------------------ */
int ffhip_device_count(void);
ffhip_engine *ffhip_engine_create(int device);
void ffhip_engine_destroy(ffhip_engine *eng);
int ffhip_engine_synchronize(ffhip_engine *eng);
/* name, CU count, clock in kHz of the engine's device */
int ffhip_engine_info(const ffhip_engine *eng, char *name, size_t name_len, int *ncu, int *clock_khz);

/* ---- model: weights re-packed into MFMA fragment order and kept resident in HBM ------------ */
/* replaces the static model instances of networks.c:218-399 */
ffhip_model *ffhip_model_upload(ffhip_engine *eng, const ffhip_model_desc *desc);
void ffhip_model_free(ffhip_model *mdl);
size_t ffhip_model_hidden(const ffhip_model *mdl);
size_t ffhip_model_nparam(const ffhip_model *mdl);       /* nstate * (nbase + 1), rows of `trans` */
size_t ffhip_model_nbase(const ffhip_model *mdl);
size_t ffhip_model_launch_reads(const ffhip_model *mdl);  /* reads per batch that keep every layer launch of this model full on this device (MI355X: 1024 at 256 hidden units, 512 at 384, else 256) */
size_t ffhip_model_nblock(const ffhip_model *mdl, size_t nsample);   /* iceil chain, layers.c:204 */

/* ---- batch: `nread` reads of `nsample` samples each, workspace resident in HBM -------------- */
ffhip_batch *ffhip_batch_create(ffhip_engine *eng, const ffhip_model *mdl, int nread, size_t nsample);
void ffhip_batch_destroy(ffhip_batch *b);
size_t ffhip_batch_nblock(const ffhip_batch *b);

/* replaces features_from_raw (nnfeatures.c:15-28): copies raw[start..end) of every read to HBM.
 * `reads[i].end - reads[i].start` may be anything from the convolution window up to nsample: `nsample` is the
 * batch's CAPACITY.  When the lengths differ the batch is RAGGED: every read is evaluated whole and exactly as
 * if it were alone (its own right-edge convolution windows, recurrent state started at its own end for the
 * backward layers, its own block count in the CRF normaliser and the decoders); a read tile of 16 reads costs
 * what its longest member costs, so callers should sort reads by length.  Equal lengths take the uniform path. */
int ffhip_batch_set_reads(ffhip_batch *b, const raw_table *reads);
/* same from a packed host array signals[nread][ld], all reads nsample long */
int ffhip_batch_set_signals(ffhip_batch *b, const float *signals, size_t ld);
/* same, read r uses the first nsample[r] <= capacity samples of its row; nsample[r] == 0 leaves slot r empty
 * (no work, no results) so that one batch object can serve groups of fewer reads */
int ffhip_batch_set_signals_ragged(ffhip_batch *b, const float *signals, size_t ld, const size_t *nsample);
/* blocks of one read of the batch (ffhip_batch_nblock is the capacity's); sizes of that read's results:
 * path/qpath nblock+1, transitions/posterior nblock x nparam, trace (nblock+1) x nstate */
size_t ffhip_batch_read_nblock(const ffhip_batch *b, int read);

/* replaces calculate_transitions (networks.c:108) + transpost_crf_flipflop (decode.c:377) +
 * decode_crf_flipflop (decode.c:119) + change_positions / base+quality assembly
 * (flappie.c:284-292) + exp_activation_inplace / trace_from_posterior (flappie.c:299-300)
 * for the whole batch.  Asynchronous on the batch's stream; results stay in HBM. */
int ffhip_batch_run(ffhip_batch *b, float temperature, unsigned flags);
/* The same for TWO batches of one model and one shape (reads per batch, capacity), with the recurrent layers of both as ONE launch
 * per layer: at H = 384 a 256-read batch fills half of what the layer kernel's dense form carries (two workgroups per CU, 512
 * reads), so a pair runs its five layers in about the time one batch needs alone.  Each batch is still finished and read on its
 * own (ffhip_batch_finish).  Shapes or flags the paired launch does not take run one batch after the other, as two ffhip_batch_run calls. */
int ffhip_batch_run_pair(ffhip_batch *b0, ffhip_batch *b1, float temperature, unsigned flags);
/* 1 if the last run's layer launches were shared with another batch (ffhip_batch_run_pair took the paired path) */
int ffhip_batch_paired(const ffhip_batch *b);
/* copy the small results (calls, qualities, lengths, scores) to pinned host memory and wait */
int ffhip_batch_finish(ffhip_batch *b);

/* ---- results (valid after ffhip_batch_finish) ------------------------------------------------ */
/* basecall/quality: NUL-terminated, at most nblock chars.  Returned pointers stay owned by the batch. */
const char *ffhip_batch_basecall(const ffhip_batch *b, int read, size_t *length);
const char *ffhip_batch_quality(const ffhip_batch *b, int read);
float ffhip_batch_score(const ffhip_batch *b, int read);             /* decode_crf_flipflop return value */
/* on-demand device-to-host copies; out buffers are caller owned */
int ffhip_batch_get_path(ffhip_batch *b, int read, int *path /*[nblock+1]*/, float *qpath /*[nblock+1]*/);
int ffhip_batch_get_transitions(ffhip_batch *b, int read, float *out /*[nblock][nparam] packed*/);
int ffhip_batch_get_posterior(ffhip_batch *b, int read, float *out /*[nblock][nparam] packed, log*/);
int ffhip_batch_get_trace(ffhip_batch *b, int read, int32_t *out /*[nblock+1][nstate] packed*/);
/* debug taps used by the parity tests: activations after conv stack (layer -1) or after RNN layer l
 * (0..4) as dense [nblock][hidden] */
int ffhip_batch_get_activation(ffhip_batch *b, int layer, int read, float *out);
/* which recurrent implementation the last ffhip_batch_run used: 0 = one launch per step, 1 = persistent recurrence behind a
 * projection GEMM, 2 = fused f32-MFMA layer kernel, 3 = split-operand layer kernel (two fp16 slices per operand; hidden 128/256/384/512), 4 = split-operand projection GEMM +
 * recurrence-only split-operand layer kernel (LSTM, hidden 256/512 under FFHIP_RUN_UNFUSED_RNN) */
int ffhip_batch_rnn_path(const ffhip_batch *b);
/* debug tap: `ntile` tiles of 16 reads x `hidden` values (hidden % 128 == 0) through the split activation layout of
 * the recurrent layer kernel and back; two fp16 slices of value * 2^12 hold 22 bits: |out - in| <= 2^-22 for |in| <= 1 (the bf16x3 build: bit for bit) */
int ffhip_debug_split_round_trip(ffhip_engine *eng, const float *in, float *out, size_t ntile, int hidden);
/* debug tap: the gate math of the persistent layer kernels (reciprocal by Newton steps instead of the division expansion,
 * floor instead of truncate/compare/subtract) against the reference-order arithmetic, on every fp32 mantissa at binary
 * exponent `exponent` (0..125); `steps` = Newton steps before the closing step (the kernels use 1).  Counts mismatching bit
 * patterns: 0 means bit-identical */
int ffhip_debug_lean_math_check(ffhip_engine *eng, int exponent, int steps, unsigned long long *mismatches);

/* debug tap: every op_sel / op_sel_hi form of the packed-fp32 VALU instructions checked against the scalar
 * instructions in a loop on a stream of its own, so that it can run beside a batch's kernels; counts[4][16][2][4] mismatches by
 * instruction (add, mul, fma, v_pk_mov_b32: forms 0, 4, 8, 12 only), form (op_sel[0], op_sel[1], op_sel_hi[0], op_sel_hi[1] as a 4-bit number), result half, wave quarter */
int ffhip_debug_pk_probe(ffhip_engine *eng, int iters, int nwg, int ballast, unsigned *counts);

/* ---- flappie matrices with a device image -------------------------------------------------------
 * An ffhip_mat describes one flappie matrix (flappie_matrix.h:18-24: column-major, `nc` columns of `stride` = 4*ceil(nr/4) floats):
 * its HOST image `data` and, through `dev` / `dev_state`, the device image the matrix owns (the two members include/flappie_matrix.h
 * appends to `_Mat`; both may be NULL for a plain host array).  *dev_state: 0 = host only, 1 = host and device equal, 2 = the DEVICE
 * image is the current one (the host image is stale until flappie_matrix_sync()).  The operators below read the device image when
 * there is one (no upload), and leave their result on the device (state 2, no download) when the matrix they are given was produced
 * there -- see INTEGRATION.md section 3 for the rules and FLAPPIE_HOST_MATRICES=1 for the reference's host-only behaviour. */
typedef struct { float *data; size_t nr, nc, stride; void **dev; int *dev_state; } ffhip_mat;
/* buffers of device images come from (and go back to) a pool per device: no hipMalloc / hipFree per matrix */
void ffhip_dev_release(void *dev);
/* A matrix whose struct the caller may release with a plain free() (the reference's flappie.c:281 does that to the transition matrix):
 * `owner` (the struct's address) is remembered with its image; ffhip_dev_forget(owner) returns the image still recorded for that address
 * (NULL if none) and drops the record -- make_flappie_matrix calls it on every new struct, so an address handed out again returns the
 * orphaned buffer.  Any release of an image (ffhip_dev_release, an operator replacing a stale image) drops its record as well.  Thread-safe. */
void ffhip_dev_remember(const void *owner, void *dev);
void *ffhip_dev_forget(const void *owner);
/* the pool's books, for tests: {buffers it allocated and has not freed, of those in its free lists, remembered owners, buffers that sit
 * TWICE in a free list (must be 0: such a buffer would be handed to two matrices)} */
void ffhip_debug_pool_state(unsigned long long out[4]);
/* device image -> host image (the whole [nc][stride] image); synchronous */
int ffhip_dev_download(const void *dev, float *host, size_t nfloat);
/* host image -> a new device image (pool buffer); NULL on failure */
void *ffhip_dev_upload(const float *host, size_t nfloat);
/* 0: never keep device images (every operator uploads its inputs and downloads its result: the reference's semantics, FLAPPIE_HOST_MATRICES=1);
 * 1 (default): as described above */
void ffhip_set_matrix_policy(int device_images);
int ffhip_matrix_policy(void);
/* calls and bytes of host<->device copies this library has made since the last reset: {h2d calls, h2d bytes, d2h calls, d2h bytes,
 * largest single d2h in bytes} -- what tests/test_host_layer.py counts for the relinked flappie.c */
void ffhip_copy_counts(unsigned long long out[5], int reset);
/* the transition matrix of read `read` of a finished batch as a device image owned by `out` (device-to-device; out.dev / out.dev_state set) */
int ffhip_batch_transitions_to(ffhip_batch *b, int read, ffhip_mat out);
/* transpost_crf_flipflop / decode_crf_flipflop / trace_from_posterior on matrices with device images (decode.c:377-497, :119-204, :499-543) */
int ffhip_op_transpost(ffhip_engine *eng, ffhip_mat trans, int return_log, ffhip_mat post);
int ffhip_op_viterbi(ffhip_engine *eng, ffhip_mat scores, int combine_stays, int *path, float *qpath, float *score);
int ffhip_op_trace(ffhip_engine *eng, ffhip_mat post, int32_t *out);

/* ---- single-matrix decode entry points -------------------------------------------------------
 * Used by the reference-compatible wrappers in include/decode.h.  `trans` / `scores` / `post` are
 * host arrays in the reference's flappie_matrix image: `nblock` columns of `stride` floats, the
 * first `nparam` = nstate*(nbase+1) of each column meaningful. */
/* transpost_crf_flipflop (decode.c:377-497): log posterior (return_log != 0) or probabilities */
int ffhip_transpost(ffhip_engine *eng, const float *trans, size_t nblock, size_t nparam, size_t stride,
                    int return_log, float *post_out);
/* decode_crf_flipflop (decode.c:119-204): path[nblock+1], qpath[nblock+1] (qpath[0] = NAN) */
int ffhip_viterbi(ffhip_engine *eng, const float *scores, size_t nblock, size_t nparam, size_t stride,
                  int combine_stays, int *path, float *qpath, float *score);
/* trace_from_posterior (decode.c:499-543): `post` holds PROBABILITIES; out[nblock+1][nstate] packed */
int ffhip_trace(ffhip_engine *eng, const float *post, size_t nblock, size_t nparam, size_t stride, int32_t *out);

/* ---- layer operators on single matrices ---------------------------------------------------------
 * The reference's per-layer interface (layers.h:15-100, flappie_matrix.h:67-73) on the GPU, used by the
 * wrappers in include/layers.h, on ffhip_mat views (above).  Outputs must be allocated by the caller with the shape the
 * reference function would return; every call is synchronous.  Batch-of-one use of the same kernels the
 * batched pipeline runs. */
/* the flip-flop decoders of decode.h that flappie.c does not use (one wave per call, reference order):
 * argmax_decoder (decode.c:17-36): seq[nc] = row of the column maximum (-1 for the last row), *score = their sum in block order */
int ffhip_op_argmax_decoder(ffhip_engine *eng, ffhip_mat logpost, int *seq, float *score);
/* constrained_crf_flipflop (decode.c:209-270): post [nstate x nblock] per-state scores, path[nblock+1] */
int ffhip_op_constrained_flipflop(ffhip_engine *eng, ffhip_mat post, int *path, float *score);
/* posterior_crf_flipflop (decode.c:275-372) in log space: out [nstate x nblock+1] per-state forward + backward */
int ffhip_op_posterior_flipflop(ffhip_engine *eng, ffhip_mat trans, ffhip_mat out);

enum ffhip_activation {
    FFHIP_ACT_NONE = 0,
    FFHIP_ACT_SWISH = 1,        /* swish_activation_inplace, layers.c:24-33    */
    FFHIP_ACT_TANH = 2,         /* tanh_activation_inplace, layers.c:40-49     */
    FFHIP_ACT_EXP = 3,          /* exp_activation_inplace, layers.c:56-66      */
    FFHIP_ACT_LOG = 4,          /* log_activation_inplace, layers.c:73-81      */
    FFHIP_ACT_ELU = 5,          /* elu_activation_inplace, layers.c:88-96      */
    FFHIP_ACT_ROBUSTLOG = 6,    /* robustlog_activation_inplace, layers.c:109-124: log(p0 + p1*x) */
    FFHIP_ACT_SHIFT_SCALE = 7   /* shift_scale_matrix_inplace, flappie_matrix.c:625-633: (x - p0)/p1 */
};
/* element-wise over the whole image, pad lanes included (as the reference's SSE loops do) */
int ffhip_op_activation(ffhip_engine *eng, ffhip_mat C, int act, float p0, float p1);
/* residual_inplace (layers.c:338-353): Y += X */
int ffhip_op_add_inplace(ffhip_engine *eng, ffhip_mat Y, ffhip_mat X);
/* row_normalise_inplace / log_row_normalise_inplace (flappie_matrix.c:425-467) */
int ffhip_op_row_normalise(ffhip_engine *eng, ffhip_mat C, int log_space);
/* convolution (layers.c:189-276), including its strided right-edge behaviour; C is [W.nc x ceil(X.nc/stride)] */
int ffhip_op_convolution(ffhip_engine *eng, ffhip_mat X, ffhip_mat W, ffhip_mat b, size_t conv_stride, ffhip_mat C);
/* affine_map / affine_map2 (flappie_matrix.c:361-419): C = Wf^T Xf (+ Wb^T Xb) + b; pass Xb.data == NULL for one input */
int ffhip_op_affine(ffhip_engine *eng, ffhip_mat Xf, ffhip_mat Wf, ffhip_mat Xb, ffhip_mat Wb, ffhip_mat b, ffhip_mat C);
/* lstm_forward/backward (layers.c:877-976), grumod_forward/backward (layers.c:571-660); kind = enum ffhip_net_kind */
int ffhip_op_recurrent(ffhip_engine *eng, int kind, ffhip_mat Xa, ffhip_mat sW, int backward, ffhip_mat out);
/* lstm_step (layers.c:979-1026) / grumod_step (layers.c:664-715); `state` = LSTM cell state, updated in place */
int ffhip_op_recurrent_step(ffhip_engine *eng, int kind, ffhip_mat x, ffhip_mat h_prev, ffhip_mat sW, ffhip_mat state, ffhip_mat h_out);
/* sloika GRU: gru_forward/backward (layers.c:412-510; relu = 0) and gru_relu_forward/backward (layers.c:718-816; relu = 1);
 * X [3H x T] projected input, sW [H x 2H], sW2 [H x H].  One workgroup per call: no registered model uses these layers. */
int ffhip_op_gru(ffhip_engine *eng, int relu, ffhip_mat X, ffhip_mat sW, ffhip_mat sW2, int backward, ffhip_mat out);
/* gru_step (layers.c:513-568) / gru_relu_step (layers.c:819-874) */
int ffhip_op_gru_step(ffhip_engine *eng, int relu, ffhip_mat x, ffhip_mat istate, ffhip_mat sW, ffhip_mat sW2, ffhip_mat ostate);
/* crf_manystay_partition_function (layers.c:1035-1079) */
int ffhip_op_partition_function(ffhip_engine *eng, ffhip_mat S, double *logZ);
/* the same quantity by the batched pipeline's scaled linear-space recursion; requires |S| <= bound everywhere */
int ffhip_op_partition_function_scaled(ffhip_engine *eng, ffhip_mat S, float bound, double *logZ);
/* globalnorm_flipflop (layers.c:1082-1106) */
int ffhip_op_globalnorm_flipflop(ffhip_engine *eng, ffhip_mat X, ffhip_mat W, ffhip_mat b, float temperature, ffhip_mat C);

/* ---- run-length (runnie) head and decoders on single matrices ------------------------------------------
 * globalnorm_runlengthV2 (layers.c:1325-1358), runlengthV2_partition_function (layers.c:1255-1302),
 * transpost_crf_runlength (decode.c:1037-1159), decode_crf_runlength (decode.c:927-1013). */
int ffhip_op_globalnorm_runlength(ffhip_engine *eng, ffhip_mat X, ffhip_mat W, ffhip_mat b, float temperature, ffhip_mat C);
int ffhip_op_runlength_partition_function(ffhip_engine *eng, ffhip_mat S, double *logZ);
/* first-generation head: globalnorm_runlength (layers.c:1197-1228), runlength_partition_function (layers.c:1127-1174) */
int ffhip_op_globalnorm_runlength_v1(ffhip_engine *eng, ffhip_mat X, ffhip_mat W, ffhip_mat b, float temperature, ffhip_mat C);
int ffhip_op_runlength_partition_function_v1(ffhip_engine *eng, ffhip_mat S, double *logZ);
int ffhip_runlength_transpost(ffhip_engine *eng, ffhip_mat param, ffhip_mat post);
int ffhip_runlength_viterbi(ffhip_engine *eng, ffhip_mat param, int *path /* nblock */, float *score);
/* decoders of the first-generation head on [4 nbase x nblock] matrices: decode_runlength (decode.c:694-767), posterior_runlength
 * (decode.c:793-892; post is [4 nbase x nblock + 1]), runlengths_mean (decode.c:576-603) */
int ffhip_runlength_v1_viterbi(ffhip_engine *eng, ffhip_mat param, int *path /* nblock: base entered, -1 = stay */, float *score);
int ffhip_runlength_v1_posterior(ffhip_engine *eng, ffhip_mat param, ffhip_mat post);
int ffhip_runlength_v1_mean(ffhip_engine *eng, ffhip_mat param, const int *path, int *runlength /* nblock */, size_t *seqlen);

/* ---- signal preparation on the GPU ------------------------------------------------------------------
 * trim_and_segment_raw (flappie_common.c:13-81) followed by medmad_normalise_array (util.c:198-212) or the
 * --delta transform (flappie.c:259-262) for a set of raw reads of any lengths, one workgroup per read; every
 * order statistic is an exact selection, so ranges and signals equal the reference's qsort-based ones.
 * The prepared signals stay in HBM; ffhip_batch_set_prepared feeds equal-length ones to a batch. */
typedef struct ffhip_prep ffhip_prep;
#define FFHIP_PREP_MEDMAD 0   /* medmad_normalise_array                                  */
#define FFHIP_PREP_DELTA  1   /* difference_array + shift_scale_array(0, delta)          */
#define FFHIP_PREP_NONE   2   /* trim only, samples copied                               */
#define FFHIP_PREP_DIFFERENCE  3   /* difference_array (util.c:416-427), ffhip_array_transform only  */
#define FFHIP_PREP_SHIFT_SCALE 4   /* shift_scale_array (util.c:214-223), ffhip_array_transform only */
ffhip_prep *ffhip_prep_create(ffhip_engine *eng, const raw_table *reads, int nread, size_t trim_start, size_t trim_end,
                              size_t varseg_chunk, float varseg_thresh, int mode, float delta);
/* the same in two halves: begin enqueues (uploads, kernel, the ranges' copies) and returns, finish waits -- a pipeline begins chunk k + 1 before it submits chunk k's
 * batches, and the preparation runs beside their convolutions.  One preparation may be pending at a time; ranges, statistics and signals are valid after finish. */
ffhip_prep *ffhip_prep_begin(ffhip_engine *eng, const raw_table *reads, int nread, size_t trim_start, size_t trim_end,
                             size_t varseg_chunk, float varseg_thresh, int mode, float delta);
int ffhip_prep_finish(ffhip_prep *p);
void ffhip_prep_destroy(ffhip_prep *p);      /* waits for the copies ffhip_batch_set_prepared enqueued from it, not for the batches */
/* start >= end: the read was rejected (trim_and_segment_raw would have returned a NULL table) */
int ffhip_prep_range(const ffhip_prep *p, int read, size_t *start, size_t *end);
int ffhip_prep_stats(const ffhip_prep *p, int read, float *median, float *mad);      /* MEDMAD mode */
int ffhip_prep_get_signal(const ffhip_prep *p, int read, float *out /* end-start floats */);
/* device-to-device and asynchronous on the batch's stream (one gather launch; the call does not wait for the GPU) */
int ffhip_batch_set_prepared(ffhip_batch *b, const ffhip_prep *prep, const int *reads /* batch nread indices into prep (-1 = empty slot); lengths <= capacity */);

/* ---- packed batches: reads of ANY lengths, several to a row ---------------------------------------------------------------
 * The reference takes reads of any length one at a time (flappie.c:245-262, 334-385).  A batch above costs what its longest read costs -- a launch per
 * layer runs as many steps as that read has blocks whatever the others' lengths -- so a nanopore-like length mix (a long tail of 100 000-sample reads among
 * 5000-sample ones) fills a fraction of it.  A PACKED batch is `nslot` rows of `nsample` samples; a row holds one or more reads one behind the other, each
 * starting at a block offset of the caller's choice with at least ffhip_model_pack_gap() free blocks behind it.  Every read is still evaluated whole and
 * exactly as if it were alone (bit for bit what the one-read-a-row batch gives): the convolutions see zero padding either side of it, the recurrent
 * layers start from a zero state at its first block and, in the reverse layers, at its last; partition function, posterior, Viterbi, strings and trace
 * are per read.  Results are indexed by READ, 0 .. nread - 1 in the order of the set call.  The default path only (flip-flop models with 128 .. 512 hidden units;
 * no FFHIP_RUN_KEEP_ACTS / _F32_RNN / _STEPWISE_RNN / _UNFUSED_RNN): ffhip_batch_run says so otherwise.  ffhip_batch_run_pair takes packed batches too. */
int ffhip_model_packable(const ffhip_model *mdl);         /* 1: this model's default path takes packed batches on this device */
size_t ffhip_model_pack_gap(const ffhip_model *mdl);      /* free blocks a read of a packed row needs behind it */
/* plan of a packed batch: slot[i] / block_off[i] for every read (slot -1: it did not fit into nslot rows of nsample_cap samples); returns the reads placed.  Longest read first,
 * each into the row that holds least so far: a launch runs as long as its longest row, and this rule leaves all rows within a short read of total / nslot (first fit -- the
 * rule before round 6's third session, FFHIP_DEBUG=pack_first_fit -- fills row after row to nsample_cap: profiles/r06_pack_bench.txt, fill 0.93 -> 0.98) */
int ffhip_pack_plan(const ffhip_model *mdl, int nslot, size_t nsample_cap, int nread, const size_t *nsample, int *slot, int *block_off);
/* rows (a multiple of 16, <= want_rows) of a packed batch of nsample-sample rows whose workspace fits 36 % of the device's memory (two such objects are
 * alive in a pipeline): 512 rows of 200 000 samples are ~90 GB at 384 hidden units */
int ffhip_pack_rows(const ffhip_model *mdl, int want_rows, size_t nsample);
int ffhip_pack_rows_for(const ffhip_model *mdl, int want_rows, size_t nsample, int nobjects);      /* the same for a pipeline of nobjects objects (72 % / nobjects each) */
ffhip_batch *ffhip_batch_create_packed(ffhip_engine *eng, const ffhip_model *mdl, int nslot, size_t nsample, int max_reads);
int ffhip_batch_set_prepared_packed(ffhip_batch *b, const ffhip_prep *prep, int nread, const int *reads /* indices into prep */, const int *slot, const int *block_off);
int ffhip_batch_set_signals_packed(ffhip_batch *b, int nread, const float *const *signals, const size_t *nsample, const int *slot, const int *block_off);
int ffhip_batch_nreads(const ffhip_batch *b);             /* reads of the last set call (a packed batch: its reads, not its rows) */
/* quantilef (util.c:100-139): p[] in, quantiles out */
int ffhip_quantiles(ffhip_engine *eng, const float *x, size_t n, float *p, size_t np);
/* difference_array / shift_scale_array / both (FFHIP_PREP_DELTA) on one host array, in place */
int ffhip_array_transform(ffhip_engine *eng, float *x, size_t n, int mode, float shift, float scale);
/* madf (util.c:164-187): 1.4826 * median(|x - med|); med == NULL: about the array's own median */
int ffhip_mad(ffhip_engine *eng, const float *x, size_t n, const float *med, float *mad);
/* medmad_normalise_array (util.c:198-212) in place; optionally returns the median and MAD used */
int ffhip_medmad_normalise(ffhip_engine *eng, float *x, size_t n, float *median, float *mad);

/* ---- measurement ----------------------------------------------------------------------------- */
/* HIP-event timing of the kernel groups of one batch_run, on the stream they are launched on.
 * groups: 0 conv, 1 in-projection GEMMs, 2 recurrent, 3 head+CRF norm, 4 posterior, 5 viterbi+assembly */
#define FFHIP_NGROUP 6
int ffhip_engine_set_profiling(ffhip_engine *eng, int on);
int ffhip_batch_profile(const ffhip_batch *b, float ms[FFHIP_NGROUP], int launches[FFHIP_NGROUP]);
/* how often a batch of this engine was re-run on the launch-per-step kernels because a persistent layer launch timed out waiting for
 * its peer workgroups (another tenant on the GPU); each occurrence also warns on stderr once per process and sends the next 64 runs
 * to those kernels directly */
int ffhip_debug_fallback_count(const ffhip_engine *eng);
/* Reads whose swish-convolution outputs left the range of the default path's operand format (two fp16 slices of value * 2^4: +-4094;
 * the reference's swish_activation_inplace, layers.c:24-33, has no bound): the producing kernels flag them, ffhip_batch_finish runs
 * them again through the all-f32 kernels (FFHIP_RUN_F32_RNN, no bound) and puts those results in place -- no read is returned
 * clamped.  The count of the batch's last run / of the engine's lifetime (the flappie binary prints the latter in its summary). */
int ffhip_batch_f32_reruns(const ffhip_batch *b);
unsigned long long ffhip_engine_f32_reruns(const ffhip_engine *eng);

#ifdef __cplusplus
}
#endif
#endif
