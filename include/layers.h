/*  layers.h -- per-layer operators of the drop-in boundary, computed on the MI355X.
 *
 *  Same names, argument meaning and error behaviour as /root/reference/src/layers.h:15-100: every function
 *  that returns a matrix takes an optional output `C` that is reused iff it already has the right shape,
 *  else freed and reallocated (flappie_matrix.c:54-61); NULL input propagates as NULL output; the in-place
 *  activations touch the pad lanes of each column as the reference's SSE loops do.  Each call runs the
 *  batch-of-one form of the kernels behind calculate_transitions (ffhip_op_*, include/ffhip.h) and is
 *  synchronous.  On a machine without a usable gfx950 device the functions warn and return NULL.
 *
 *  The sloika GRU layers (gru_*, gru_relu_*) and the first-generation globalnorm_runlength head, which no model in the
 *  reference's registry uses (networks.c:85-99), are provided as correctness-level operators (one workgroup per call).
 */
#ifndef FFHIP_LAYERS_H
#define FFHIP_LAYERS_H

#include "flappie_matrix.h"

#ifdef __cplusplus
extern "C" {
#endif

/* layers.c:24-124 */
void swish_activation_inplace(flappie_matrix C);
void tanh_activation_inplace(flappie_matrix C);
void exp_activation_inplace(flappie_matrix C);
void log_activation_inplace(flappie_matrix C);
void elu_activation_inplace(flappie_matrix C);
void robustlog_activation_inplace(flappie_matrix C, float min_prob);

/* layers.c:127-187: pure data movement, done on the host */
flappie_matrix embedding(int const *index, size_t n, const_flappie_matrix E, flappie_matrix C);
flappie_matrix window(const_flappie_matrix input, size_t w, size_t stride);

/* layers.c:189-276 */
flappie_matrix convolution(const_flappie_matrix X, const_flappie_matrix W, const_flappie_matrix b, size_t stride,
                           flappie_matrix C);
/* layers.c:279-310 */
flappie_matrix feedforward_linear(const_flappie_matrix X, const_flappie_matrix W, const_flappie_matrix b, flappie_matrix C);
flappie_matrix feedforward_tanh(const_flappie_matrix X, const_flappie_matrix W, const_flappie_matrix b, flappie_matrix C);
flappie_matrix feedforward_exp(const_flappie_matrix X, const_flappie_matrix W, const_flappie_matrix b, flappie_matrix C);
/* layers.c:313-353 */
flappie_matrix residual(const_flappie_matrix X, const_flappie_matrix fX, flappie_matrix C);
void residual_inplace(const_flappie_matrix X, flappie_matrix fX);
/* layers.c:356-395 */
flappie_matrix softmax(const_flappie_matrix X, const_flappie_matrix W, const_flappie_matrix b, flappie_matrix C);
flappie_matrix softmax_with_temperature(flappie_matrix X, const_flappie_matrix W, const_flappie_matrix b, float tempW,
                                        float tempb, flappie_matrix C);
/* layers.c:398-410 */
flappie_matrix feedforward2_tanh(const_flappie_matrix Xf, const_flappie_matrix Xb, const_flappie_matrix Wf,
                                 const_flappie_matrix Wb, const_flappie_matrix b, flappie_matrix C);

/* layers.c:412-568 and 718-874: the sloika GRU; X is the projected input [3H x T], sW [H x 2H], sW2 [H x H] */
flappie_matrix gru_forward(const_flappie_matrix X, const_flappie_matrix sW, const_flappie_matrix sW2, flappie_matrix res);
flappie_matrix gru_backward(const_flappie_matrix X, const_flappie_matrix sW, const_flappie_matrix sW2, flappie_matrix res);
void gru_step(const_flappie_matrix x, const_flappie_matrix istate, const_flappie_matrix sW, const_flappie_matrix sW2,
              flappie_matrix xF, flappie_matrix ostate);
flappie_matrix gru_relu_forward(const_flappie_matrix X, const_flappie_matrix sW, const_flappie_matrix sW2, flappie_matrix res);
flappie_matrix gru_relu_backward(const_flappie_matrix X, const_flappie_matrix sW, const_flappie_matrix sW2, flappie_matrix res);
void gru_relu_step(const_flappie_matrix x, const_flappie_matrix istate, const_flappie_matrix sW, const_flappie_matrix sW2,
                   flappie_matrix xF, flappie_matrix ostate);

/* layers.c:571-715: X is the projected input [3H x T], sW [H x 3H] */
flappie_matrix grumod_forward(const_flappie_matrix X, const_flappie_matrix sW, flappie_matrix res);
flappie_matrix grumod_backward(const_flappie_matrix X, const_flappie_matrix sW, flappie_matrix res);
void grumod_step(const_flappie_matrix x, const_flappie_matrix istate, const_flappie_matrix sW, flappie_matrix xF,
                 flappie_matrix ostate);

/* layers.c:877-1026: X is the projected input [4H x T], sW [H x 4H] */
flappie_matrix lstm_forward(const_flappie_matrix X, const_flappie_matrix sW, flappie_matrix output);
flappie_matrix lstm_backward(const_flappie_matrix X, const_flappie_matrix sW, flappie_matrix output);
void lstm_step(const_flappie_matrix x, const_flappie_matrix out_prev, const_flappie_matrix sW, flappie_matrix xF,
               flappie_matrix state, flappie_matrix output);

/* layers.c:1029-1106 */
double crf_manystay_partition_function(const_flappie_matrix C);
flappie_matrix globalnorm_manystay(const_flappie_matrix X, const_flappie_matrix W, const_flappie_matrix b, float temperature,
                                   flappie_matrix C);
size_t nbase_from_flipflop_nparam(size_t nparam);
flappie_matrix globalnorm_flipflop(const_flappie_matrix X, const_flappie_matrix W, const_flappie_matrix b, float temperature,
                                   flappie_matrix C);
/* layers.c:1115-1228: the first-generation run-length head (rows shape, scale, move, stay; scale = 0.1 + softplus) */
size_t nbase_from_runlength_nparam(size_t nparam);
double runlength_partition_function(const_flappie_matrix C);
flappie_matrix globalnorm_runlength(const_flappie_matrix X, const_flappie_matrix W, const_flappie_matrix b, float temperature,
                                    flappie_matrix C);
/* layers.c:1230-1358: the run-length head of runnie's model (shape = 1 + softplus, scale = 1e-8 + softplus,
 * transitions 5 tanh / temperature, globally normalised) and its fp64 partition function */
size_t nbase_from_crf_runlength_nparam(size_t nparam);
double runlengthV2_partition_function(const_flappie_matrix C);
flappie_matrix globalnorm_runlengthV2(const_flappie_matrix X, const_flappie_matrix W, const_flappie_matrix b, float temperature,
                                      flappie_matrix C);

#ifdef __cplusplus
}
#endif
#endif
