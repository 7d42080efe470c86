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

/* A shorthand for this header only: every operand below is a read-only matrix handle. */
#define FM_IN const_flappie_matrix

/* ---- element-wise, in place, pad lanes included (layers.c:24-124) ---- */
void swish_activation_inplace(flappie_matrix acts);
void tanh_activation_inplace(flappie_matrix acts);
void exp_activation_inplace(flappie_matrix acts);
void log_activation_inplace(flappie_matrix acts);
void elu_activation_inplace(flappie_matrix acts);
void robustlog_activation_inplace(flappie_matrix probs, float min_prob);

/* ---- pure data movement, done on the host (layers.c:127-187) ---- */
flappie_matrix embedding(int const *index, size_t nindex, FM_IN table, flappie_matrix out);
flappie_matrix window(FM_IN signal, size_t winlen, size_t stride);

/* ---- strided convolution with the reference's edge behaviour (layers.c:189-276); out is [nfilter x ceil(T / stride)] ---- */
flappie_matrix convolution(FM_IN signal, FM_IN filters, FM_IN bias, size_t stride, flappie_matrix out);

/* ---- out = weights^T input + bias, then nothing / tanh / exp (layers.c:279-310) ---- */
flappie_matrix feedforward_linear(FM_IN input, FM_IN weights, FM_IN bias, flappie_matrix out);
flappie_matrix feedforward_tanh(FM_IN input, FM_IN weights, FM_IN bias, flappie_matrix out);
flappie_matrix feedforward_exp(FM_IN input, FM_IN weights, FM_IN bias, flappie_matrix out);
/* two inputs, two weight matrices, one bias (layers.c:398-410) */
flappie_matrix feedforward2_tanh(FM_IN input_f, FM_IN input_b, FM_IN weights_f, FM_IN weights_b, FM_IN bias, flappie_matrix out);

/* ---- out = skip + branch (layers.c:313-353) ---- */
flappie_matrix residual(FM_IN skip, FM_IN branch, flappie_matrix out);
void residual_inplace(FM_IN skip, flappie_matrix branch);

/* ---- affine map, exp, row normalisation (layers.c:356-395) ---- */
flappie_matrix softmax(FM_IN input, FM_IN weights, FM_IN bias, flappie_matrix out);
flappie_matrix softmax_with_temperature(flappie_matrix input, FM_IN weights, FM_IN bias, float temp_weights, float temp_bias,
                                        flappie_matrix out);

/* ---- recurrent layers: `projected` is the already-projected input [G*H x T], `recurrent` the state weights [H x G*H], the result is
 *      [H x T].  `scratch` of the step functions is the reference's gate buffer; it is not touched here. ---- */
/* sloika GRU (layers.c:412-568) and its ReLU variant (layers.c:718-874): G = 3, recurrent [H x 2H], recurrent2 [H x H] */
flappie_matrix gru_forward(FM_IN projected, FM_IN recurrent, FM_IN recurrent2, flappie_matrix out);
flappie_matrix gru_backward(FM_IN projected, FM_IN recurrent, FM_IN recurrent2, flappie_matrix out);
void gru_step(FM_IN projected_t, FM_IN state_in, FM_IN recurrent, FM_IN recurrent2, flappie_matrix scratch, flappie_matrix state_out);
flappie_matrix gru_relu_forward(FM_IN projected, FM_IN recurrent, FM_IN recurrent2, flappie_matrix out);
flappie_matrix gru_relu_backward(FM_IN projected, FM_IN recurrent, FM_IN recurrent2, flappie_matrix out);
void gru_relu_step(FM_IN projected_t, FM_IN state_in, FM_IN recurrent, FM_IN recurrent2, flappie_matrix scratch,
                   flappie_matrix state_out);
/* GRUmod of the 5mC model (layers.c:571-715): G = 3 */
flappie_matrix grumod_forward(FM_IN projected, FM_IN recurrent, flappie_matrix out);
flappie_matrix grumod_backward(FM_IN projected, FM_IN recurrent, flappie_matrix out);
void grumod_step(FM_IN projected_t, FM_IN state_in, FM_IN recurrent, flappie_matrix scratch, flappie_matrix state_out);
/* LSTM (layers.c:877-1026): G = 4; `cell` is the cell state, updated in place */
flappie_matrix lstm_forward(FM_IN projected, FM_IN recurrent, flappie_matrix out);
flappie_matrix lstm_backward(FM_IN projected, FM_IN recurrent, flappie_matrix out);
void lstm_step(FM_IN projected_t, FM_IN hidden_in, FM_IN recurrent, flappie_matrix scratch, flappie_matrix cell, flappie_matrix hidden_out);

/* ---- flip-flop CRF head: tanh, scale by 5 / temperature, global normalisation (layers.c:1029-1106) ---- */
size_t nbase_from_flipflop_nparam(size_t nparam);
double crf_manystay_partition_function(FM_IN scores);
flappie_matrix globalnorm_manystay(FM_IN input, FM_IN weights, FM_IN bias, float temperature, flappie_matrix out);
flappie_matrix globalnorm_flipflop(FM_IN input, FM_IN weights, FM_IN bias, float temperature, flappie_matrix out);

/* ---- run-length heads.  First generation (layers.c:1115-1228): rows shape, scale, move, stay; scale = 0.1 + softplus ---- */
size_t nbase_from_runlength_nparam(size_t nparam);
double runlength_partition_function(FM_IN params);
flappie_matrix globalnorm_runlength(FM_IN input, FM_IN weights, FM_IN bias, float temperature, flappie_matrix out);
/* runnie's model (layers.c:1230-1358): shape = 1 + softplus, scale = 1e-8 + softplus, transitions 5 tanh / temperature,
 * globally normalised with the fp64 partition function below */
size_t nbase_from_crf_runlength_nparam(size_t nparam);
double runlengthV2_partition_function(FM_IN params);
flappie_matrix globalnorm_runlengthV2(FM_IN input, FM_IN weights, FM_IN bias, float temperature, flappie_matrix out);

#undef FM_IN

#ifdef __cplusplus
}
#endif
#endif
