/*  ff_oracle.h -- CPU oracle for the flip-flop basecalling hot path.
 *
 *  TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the algorithm of
 *  nanoporetech/flappie v2.1.3 for the path SURVEY.md section 8 names.  Only
 *  tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *  The product (flappie_amd/, include/) never links, imports or calls it.
 *
 *  PINNING STATUS (tests/test_ref_pins.py, tests/test_oracle_cpu.py):
 *    - scalar math (exp_ps / log_ps, logistic, tanh, elu, logsumexpf, logsumexp, qscoref / phredf): BIT FOR BIT against the
 *      reference's own header-inline code compiled by oracle/ref_inline.c (oracle/_ref/libflappie_inlref.so), > 10^7 inputs
 *      per function;
 *    - the decode half (features_from_raw, transpost_crf_flipflop, decode_crf_flipflop, change_positions, exp +
 *      trace_from_posterior, decode_crf_runlength, transpost_crf_runlength): BIT FOR BIT against the reference's decode.c +
 *      util.c + nnfeatures.c compiled unchanged (oracle/_ref/libflappie_decref.so) with the declared glue of
 *      oracle/ref_decode_glue.c for the ten allocation / normalisation symbols of the unbuildable flappie_matrix.c / layers.c;
 *    - signal preparation (N1): the reference's own fixtures raw_signal / trimmed_signal / normalised_signal (tests/golden/)
 *      AND the reference's util.c / flappie_common.c compiled unchanged (oracle/_ref/libflappie_sigref.so), bit for bit;
 *    - ELU, row_normalise, median, identity convolution: the known-answer values of the reference's CUnit tests;
 *    - strided convolution (A3), affine maps, LSTM / GRUmod cells, globalnorm_flipflop and the partition-function recursion
 *      (A5-A9): NO golden vector in the reference, and layers.c / flappie_matrix.c need an external BLAS header (<cblas.h>,
 *      absent from this image) so they cannot be built here:  **parity unpinned** for those rows (cross-checked against
 *      PyTorch, autograd and brute-force enumeration: tests/test_oracle_vs_torch.py).
 *
 *  Every function cites the reference file:line it restates (paths relative to
 *  /root/reference/src).
 */
#ifndef FF_ORACLE_H
#define FF_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Column-major fp32 matrix, rows padded to a multiple of 4 (flappie_matrix.h:18-24).
 * Field order of the first five members mirrors `_Mat` so that .mdl initialisers map 1:1. */
typedef struct {
    size_t nr, nrq, nc, stride;
    float *f;
} fo_mat;

typedef struct {
    size_t nr, nrq, nc, stride;
    int32_t *f;
} fo_imat;

enum { FO_NET_LSTM5 = 0,      /* flipflop5_guppy_transitions  networks.c:539-586 */
       FO_NET_GRUMOD5 = 1,    /* flipflop_guppy_transitions   networks.c:450-489 */
       FO_NET_LSTM5_RLE = 2 };/* runlength5_guppy_transitions networks.c:672-725: LSTM5 trunk, globalnorm_runlengthV2 head */

typedef struct {
    int kind;
    int nconv;                  /* 3 for LSTM5 (swish), 1 for GRUMOD5 (tanh) */
    const fo_mat *conv_W[3];
    const fo_mat *conv_b[3];
    int conv_stride[3];
    const fo_mat *rnn_iW[5];    /* directions fixed B,F,B,F,B */
    const fo_mat *rnn_sW[5];
    const fo_mat *rnn_b[5];
    const fo_mat *FF_W;
    const fo_mat *FF_b;
} fo_model;

/* matrix helpers (flappie_matrix.c:20-61,142-148) */
fo_mat *fo_make_mat(size_t nr, size_t nc);
fo_mat *fo_free_mat(fo_mat *m);
fo_imat *fo_make_imat(size_t nr, size_t nc);
fo_imat *fo_free_imat(fo_imat *m);
fo_mat *fo_mat_from_array(const float *x, size_t nr, size_t nc);

/* vector math (util.h:319-356, sse_mathfun.h:211-301) -- scalar, lane-exact */
float fo_expf_cephes(float x);
float fo_logisticf(float x);
float fo_tanhf(float x);
float fo_eluf(float x);
float fo_logsumexpf(float x, float y);
double fo_logsumexp(double x, double y);
char fo_phredf(float p);

/* array forms for the sweeps against the reference's inline code; kind: 0 exp, 1 log, 2 logistic, 3 tanh, 4 elu */
int fo_map_array(int kind, const float *in, float *out, size_t n);
void fo_logsumexpf_array(const float *x, const float *y, float *out, size_t n);
void fo_logsumexp_array(const double *x, const double *y, double *out, size_t n);
void fo_phredf_array(const float *p, char *out, size_t n);

/* summation of dot products: 0 reference order in float (default, the oracle), 1 double accumulator (yardstick),
 * 2 vectorised kernels of cpu_ref.c (cpu_baseline timing only) */
void fo_set_dot_mode(int mode);
int fo_get_dot_mode(void);

/* layers */
void fo_swish_inplace(fo_mat *C);
void fo_tanh_inplace(fo_mat *C);
void fo_exp_inplace(fo_mat *C);
float fo_logf_cephes(float x);
void fo_log_inplace(fo_mat *C);
void fo_elu_inplace(fo_mat *C);
void fo_robustlog_inplace(fo_mat *C, float min_prob);
fo_mat *fo_affine_map2(const fo_mat *Xf, const fo_mat *Xb, const fo_mat *Wf, const fo_mat *Wb, const fo_mat *b);
void fo_row_normalise_inplace(fo_mat *C);
void fo_log_row_normalise_inplace(fo_mat *C);
fo_mat *fo_features_from_raw(const float *raw, size_t start, size_t end);
fo_mat *fo_convolution(const fo_mat *X, const fo_mat *W, const fo_mat *b, size_t stride);
fo_mat *fo_affine_map(const fo_mat *X, const fo_mat *W, const fo_mat *b);
fo_mat *fo_lstm(const fo_mat *Xaffine, const fo_mat *sW, int backward);
fo_mat *fo_grumod(const fo_mat *X, const fo_mat *sW, int backward);
fo_mat *fo_gru(const fo_mat *X, const fo_mat *sW, const fo_mat *sW2, int backward, int relu, const float *h0);
double fo_runlength_partition_function(const fo_mat *C);
fo_mat *fo_globalnorm_runlength(const fo_mat *X, const fo_mat *W, const fo_mat *b, float temperature);
double fo_partition_function(const fo_mat *C);
fo_mat *fo_globalnorm_flipflop(const fo_mat *X, const fo_mat *W, const fo_mat *b, float temperature);
fo_mat *fo_transitions(const float *raw, size_t start, size_t end, float temperature,
                       const fo_model *net);

/* decode */
size_t fo_nbase_from_nparam(size_t nparam);
/* run-length (runnie) head and decoders: layers.c:1241-1358, decode.c:927-1159, runnie.c:282-313 */
float fo_softplusf(float x);
double fo_runlengthV2_partition_function(const fo_mat *C);
fo_mat *fo_globalnorm_runlengthV2(const fo_mat *X, const fo_mat *W, const fo_mat *b, float temperature);
float fo_decode_crf_runlength(const fo_mat *param, int *path);
fo_mat *fo_transpost_crf_runlength(const fo_mat *param);
/* first-generation run-length decoders (decode.c:552-892) */
float fo_dwmean(float shape, float scale, int maxval);
size_t fo_runlengths_mean(const fo_mat *param, const int *path, int *runlength);
size_t fo_runlengths_unit(const fo_mat *param, const int *path, int *runlength);
char *fo_runlength_to_basecall(const int *path, const int *runlength, size_t nblk);
float fo_decode_runlength(const fo_mat *param, int *path);
fo_mat *fo_posterior_runlength(const fo_mat *param);
size_t fo_runlength_records(const int *path, size_t nblock, size_t nbase, int *base, int *block, int *dwell);
fo_mat *fo_transpost(const fo_mat *trans, int return_log);
float fo_decode_viterbi(const fo_mat *trans, int combine_stays, int *path, float *qpath);
size_t fo_change_positions(const int *path, size_t npos, int *chpos);
fo_imat *fo_trace_from_posterior(const fo_mat *tpost);

/* whole read: calculate_post (flappie.c:245-316) minus file I/O and signal prep.
 * Buffers: path/qpath nblock+1, basecall/quality nblock+1 chars (NUL terminated),
 * trace nstate*(nblock+1) int32 packed (no padding), trans/post P*nblock packed or NULL. */
typedef struct {
    size_t nblock, nbase, nstate, nparam;
    float score;
    size_t basecall_length;
} fo_read_result;
int fo_basecall_read(const float *raw, size_t start, size_t end, float temperature,
                     const fo_model *net, int viterbi_only,
                     fo_read_result *res, int *path, float *qpath,
                     char *basecall, char *quality, int32_t *trace,
                     float *trans_out, float *post_out);
size_t fo_nblock_for(const fo_model *net, size_t nsample);

/* signal preparation (util.c:100-212, flappie_common.c:13-81) */
void fo_quantilef(const float *x, size_t nx, float *p, size_t np);
float fo_medianf(const float *x, size_t n);
float fo_madf(const float *x, size_t n, const float *med);
void fo_medmad_normalise_array(float *x, size_t n);
/* returns 0 on success; start/end in-out */
int fo_trim_raw_by_mad(const float *raw, size_t *start, size_t *end, size_t chunk_size, float perc);
int fo_trim_and_segment_raw(const float *raw, size_t n, size_t *start, size_t *end,
                            size_t trim_start, size_t trim_end, size_t varseg_chunk, float varseg_thresh);

#ifdef __cplusplus
}
#endif
#endif
