/*  layers.c -- the reference's per-layer operators (include/layers.h) over the HIP engine.
 *  Shape handling, output reuse and NULL propagation follow /root/reference/src/layers.c; the arithmetic
 *  runs on the GPU through ffhip_op_* (flappie_amd/csrc/ffhip_layers.hip).  Shape errors that the
 *  reference only asserts (compiled out under NDEBUG) are reported with a warning and a NULL return.
 */
#include <err.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/ffhip.h"
#include "../../include/layers.h"
#include "../../include/networks.h"

static ffhip_mat view(const_flappie_matrix m) {
    ffhip_mat v = { NULL, 0, 0, 0, NULL, NULL };
    if (NULL != m) {
        v.data = m->data.f; v.nr = m->nr; v.nc = m->nc; v.stride = m->stride;
        v.dev = (void **)&((flappie_matrix)m)->dev;             /* the device image the matrix owns (a cache: logically const) */
        v.dev_state = (int *)&((flappie_matrix)m)->dev_state;
    }
    return v;
}

/* runs `rc = call` if an engine is available; 0 on success, warns otherwise */
static int check(int rc, const char *what) {
    if (0 != rc) warnx("%s: %s", what, ffhip_last_error());
    return rc;
}

static void activation(flappie_matrix C, int act, float p0, float p1, const char *what) {
    if (NULL == C) return;
    struct ffhip_engine *eng = flappie_hip_engine();
    if (NULL == eng) return;
    (void)check(ffhip_op_activation(eng, view(C), act, p0, p1), what);
}

void swish_activation_inplace(flappie_matrix C) { activation(C, FFHIP_ACT_SWISH, 0, 0, __func__); }
void tanh_activation_inplace(flappie_matrix C) { activation(C, FFHIP_ACT_TANH, 0, 0, __func__); }
void exp_activation_inplace(flappie_matrix C) { activation(C, FFHIP_ACT_EXP, 0, 0, __func__); }
void log_activation_inplace(flappie_matrix C) { activation(C, FFHIP_ACT_LOG, 0, 0, __func__); }
void elu_activation_inplace(flappie_matrix C) { activation(C, FFHIP_ACT_ELU, 0, 0, __func__); }
void robustlog_activation_inplace(flappie_matrix C, float min_prob) {
    activation(C, FFHIP_ACT_ROBUSTLOG, min_prob, 1.0f - min_prob, __func__);
}

/* flappie_matrix.c:625-633 */
void shift_scale_matrix_inplace(flappie_matrix C, float shift, float scale) {
    activation(C, FFHIP_ACT_SHIFT_SCALE, shift, scale, __func__);
}

/* layers.c:127-147 */
flappie_matrix embedding(int const *index, size_t n, const_flappie_matrix E, flappie_matrix C) {
    if (NULL == index || NULL == E || 0 == n) return NULL;
    C = remake_flappie_matrix(C, E->nr, n);
    if (NULL == C) return NULL;
    flappie_matrix_sync(E);                     /* a host-side copy loop: it reads E's host image ... */
    flappie_matrix_host_changed(C);             /* ... and writes C's: a device image C kept from an earlier use is stale */
    for (size_t c = 0; c < n; c++) {
        if (index[c] < 0 || (size_t)index[c] >= E->nc) { warnx("embedding: index %d out of range", index[c]); return free_flappie_matrix(C); }
        memcpy(C->data.f + c * C->stride, E->data.f + (size_t)index[c] * E->stride, E->stride * sizeof(float));
    }
    return C;
}

/* layers.c:150-176 */
flappie_matrix window(const_flappie_matrix input, size_t w, size_t stride) {
    if (NULL == input || 0 == w || 0 == stride) return NULL;
    const size_t wh = (w + 1) / 2;
    flappie_matrix output = make_flappie_matrix(input->nr * w, (size_t)ceilf(input->nc / (float)stride));
    if (NULL == output) return NULL;
    flappie_matrix_sync(input);                 /* a host-side copy loop: it reads the host image */
    for (size_t col = 0; col < output->nc; col++) {
        const size_t out_offset = col * output->stride;
        const int icol = (int)(col * stride);
        int i = 0;
        for (int w1 = icol - (int)wh + 1; w1 <= icol + (int)wh; w1++) {
            if (w1 < 0 || (size_t)w1 >= input->nc) { i += (int)input->nr; continue; }
            const size_t in_offset = (size_t)w1 * input->stride;
            /* the reference writes up to nr*(2*wh) entries per column when w is odd; keep inside the column */
            for (size_t row = 0; row < input->nr; row++, i++)
                if ((size_t)i < output->stride) output->data.f[out_offset + i] = input->data.f[in_offset + row];
        }
    }
    return output;
}

flappie_matrix convolution(const_flappie_matrix X, const_flappie_matrix W, const_flappie_matrix b, size_t stride, flappie_matrix C) {
    if (NULL == X) return NULL;
    if (NULL == W || NULL == b || 0 == stride) { warnx("convolution: missing filter, bias or stride"); return NULL; }
    struct ffhip_engine *eng = flappie_hip_engine();
    if (NULL == eng) return NULL;
    C = remake_flappie_matrix(C, W->nc, (X->nc + stride - 1) / stride);
    if (NULL == C) return NULL;
    if (0 != check(ffhip_op_convolution(eng, view(X), view(W), view(b), stride, view(C)), __func__)) return free_flappie_matrix(C);
    return C;
}

/* flappie_matrix.c:361-419 */
flappie_matrix affine_map2(const_flappie_matrix Xf, const_flappie_matrix Xb, const_flappie_matrix Wf, const_flappie_matrix Wb,
                           const_flappie_matrix b, flappie_matrix C) {
    if (NULL == Xf || NULL == Xb) return NULL;
    if (NULL == Wf || NULL == Wb || NULL == b) { warnx("affine_map2: missing weights or bias"); return NULL; }
    struct ffhip_engine *eng = flappie_hip_engine();
    if (NULL == eng) return NULL;
    C = remake_flappie_matrix(C, Wf->nc, Xf->nc);
    if (NULL == C) return NULL;
    if (0 != check(ffhip_op_affine(eng, view(Xf), view(Wf), view(Xb), view(Wb), view(b), view(C)), __func__)) return free_flappie_matrix(C);
    return C;
}

flappie_matrix affine_map(const_flappie_matrix X, const_flappie_matrix W, const_flappie_matrix b, flappie_matrix C) {
    if (NULL == X) return NULL;
    if (NULL == W || NULL == b) { warnx("affine_map: missing weights or bias"); return NULL; }
    struct ffhip_engine *eng = flappie_hip_engine();
    if (NULL == eng) return NULL;
    C = remake_flappie_matrix(C, W->nc, X->nc);
    if (NULL == C) return NULL;
    if (0 != check(ffhip_op_affine(eng, view(X), view(W), view(NULL), view(NULL), view(b), view(C)), __func__)) return free_flappie_matrix(C);
    return C;
}

void row_normalise_inplace(flappie_matrix C) {
    struct ffhip_engine *eng = (NULL != C) ? flappie_hip_engine() : NULL;
    if (NULL != eng) (void)check(ffhip_op_row_normalise(eng, view(C), 0), __func__);
}

void log_row_normalise_inplace(flappie_matrix C) {
    struct ffhip_engine *eng = (NULL != C) ? flappie_hip_engine() : NULL;
    if (NULL != eng) (void)check(ffhip_op_row_normalise(eng, view(C), 1), __func__);
}

flappie_matrix feedforward_linear(const_flappie_matrix X, const_flappie_matrix W, const_flappie_matrix b, flappie_matrix C) {
    return affine_map(X, W, b, C);
}

flappie_matrix feedforward_tanh(const_flappie_matrix X, const_flappie_matrix W, const_flappie_matrix b, flappie_matrix C) {
    C = affine_map(X, W, b, C);
    tanh_activation_inplace(C);
    return C;
}

flappie_matrix feedforward_exp(const_flappie_matrix X, const_flappie_matrix W, const_flappie_matrix b, flappie_matrix C) {
    C = affine_map(X, W, b, C);
    exp_activation_inplace(C);
    return C;
}

void residual_inplace(const_flappie_matrix X, flappie_matrix fX) {
    if (NULL == X || NULL == fX) return;
    struct ffhip_engine *eng = flappie_hip_engine();
    if (NULL != eng) (void)check(ffhip_op_add_inplace(eng, view(fX), view(X)), __func__);
}

flappie_matrix residual(const_flappie_matrix X, const_flappie_matrix fX, flappie_matrix C) {
    if (NULL == X || NULL == fX) return NULL;
    if (X->nr != fX->nr || X->nc != fX->nc) { warnx("residual: shapes differ"); return NULL; }
    if (C == fX) { residual_inplace(X, C); return C; }
    if (C == X) { residual_inplace(fX, C); return C; }          /* output aliases the first input: X += fX (the sum is symmetric) */
    C = remake_flappie_matrix(C, X->nr, X->nc);
    if (NULL == C) return NULL;
    flappie_matrix_sync(fX);
    flappie_matrix_host_changed(C);
    memcpy(C->data.f, fX->data.f, fX->stride * fX->nc * sizeof(float));
    residual_inplace(X, C);
    return C;
}

flappie_matrix softmax(const_flappie_matrix X, const_flappie_matrix W, const_flappie_matrix b, flappie_matrix C) {
    C = feedforward_exp(X, W, b, C);
    row_normalise_inplace(C);
    return C;
}

flappie_matrix softmax_with_temperature(flappie_matrix X, const_flappie_matrix W, const_flappie_matrix b, float tempW, float tempb,
                                        flappie_matrix C) {
    if (NULL == X) return NULL;
    shift_scale_matrix_inplace(X, 0.0f, tempW / tempb);
    C = feedforward_linear(X, W, b, C);
    if (NULL == C) return NULL;
    shift_scale_matrix_inplace(C, 0.0f, tempb);
    exp_activation_inplace(C);
    row_normalise_inplace(C);
    return C;
}

flappie_matrix feedforward2_tanh(const_flappie_matrix Xf, const_flappie_matrix Xb, const_flappie_matrix Wf, const_flappie_matrix Wb,
                                 const_flappie_matrix b, flappie_matrix C) {
    C = affine_map2(Xf, Xb, Wf, Wb, b, C);
    tanh_activation_inplace(C);
    return C;
}

static flappie_matrix recurrent(int kind, const_flappie_matrix X, const_flappie_matrix sW, int backward, flappie_matrix out, const char *what) {
    if (NULL == X) return NULL;
    if (NULL == sW) { warnx("%s: missing recurrent weights", what); return NULL; }
    struct ffhip_engine *eng = flappie_hip_engine();
    if (NULL == eng) return NULL;
    out = remake_flappie_matrix(out, sW->nr, X->nc);
    if (NULL == out) return NULL;
    if (0 != check(ffhip_op_recurrent(eng, kind, view(X), view(sW), backward, view(out)), what)) return free_flappie_matrix(out);
    return out;
}

flappie_matrix lstm_forward(const_flappie_matrix X, const_flappie_matrix sW, flappie_matrix output) {
    return recurrent(FFHIP_NET_LSTM5, X, sW, 0, output, __func__);
}
flappie_matrix lstm_backward(const_flappie_matrix X, const_flappie_matrix sW, flappie_matrix output) {
    return recurrent(FFHIP_NET_LSTM5, X, sW, 1, output, __func__);
}
flappie_matrix grumod_forward(const_flappie_matrix X, const_flappie_matrix sW, flappie_matrix res) {
    return recurrent(FFHIP_NET_GRUMOD5, X, sW, 0, res, __func__);
}
flappie_matrix grumod_backward(const_flappie_matrix X, const_flappie_matrix sW, flappie_matrix res) {
    return recurrent(FFHIP_NET_GRUMOD5, X, sW, 1, res, __func__);
}

/* layers.c:412-510, 718-816 */
static flappie_matrix gru_layer(int relu, const_flappie_matrix X, const_flappie_matrix sW, const_flappie_matrix sW2, int backward,
                                flappie_matrix out, const char *what) {
    if (NULL == X) return NULL;
    if (NULL == sW || NULL == sW2) { warnx("%s: missing recurrent weights", what); return NULL; }
    struct ffhip_engine *eng = flappie_hip_engine();
    if (NULL == eng) return NULL;
    out = remake_flappie_matrix(out, sW2->nc, X->nc);
    if (NULL == out) return NULL;
    if (0 != check(ffhip_op_gru(eng, relu, view(X), view(sW), view(sW2), backward, view(out)), what)) return free_flappie_matrix(out);
    return out;
}
flappie_matrix gru_forward(const_flappie_matrix X, const_flappie_matrix sW, const_flappie_matrix sW2, flappie_matrix res) {
    return gru_layer(0, X, sW, sW2, 0, res, __func__);
}
flappie_matrix gru_backward(const_flappie_matrix X, const_flappie_matrix sW, const_flappie_matrix sW2, flappie_matrix res) {
    return gru_layer(0, X, sW, sW2, 1, res, __func__);
}
flappie_matrix gru_relu_forward(const_flappie_matrix X, const_flappie_matrix sW, const_flappie_matrix sW2, flappie_matrix res) {
    return gru_layer(1, X, sW, sW2, 0, res, __func__);
}
flappie_matrix gru_relu_backward(const_flappie_matrix X, const_flappie_matrix sW, const_flappie_matrix sW2, flappie_matrix res) {
    return gru_layer(1, X, sW, sW2, 1, res, __func__);
}

/* layers.c:513-568, 819-874; xF (the reference's scratch) is left untouched */
static void gru_one_step(int relu, const_flappie_matrix x, const_flappie_matrix istate, const_flappie_matrix sW, const_flappie_matrix sW2,
                         flappie_matrix ostate, const char *what) {
    if (NULL == x || NULL == istate || NULL == sW || NULL == sW2 || NULL == ostate) return;
    struct ffhip_engine *eng = flappie_hip_engine();
    if (NULL != eng) (void)check(ffhip_op_gru_step(eng, relu, view(x), view(istate), view(sW), view(sW2), view(ostate)), what);
}
void gru_step(const_flappie_matrix x, const_flappie_matrix istate, const_flappie_matrix sW, const_flappie_matrix sW2, flappie_matrix xF,
              flappie_matrix ostate) {
    (void)xF;
    gru_one_step(0, x, istate, sW, sW2, ostate, __func__);
}
void gru_relu_step(const_flappie_matrix x, const_flappie_matrix istate, const_flappie_matrix sW, const_flappie_matrix sW2, flappie_matrix xF,
                   flappie_matrix ostate) {
    (void)xF;
    gru_one_step(1, x, istate, sW, sW2, ostate, __func__);
}

/* layers.c:979-1026.  xF is the reference's scratch for the gate pre-activations; it is left untouched. */
void lstm_step(const_flappie_matrix x, const_flappie_matrix out_prev, const_flappie_matrix sW, flappie_matrix xF,
               flappie_matrix state, flappie_matrix output) {
    if (NULL == x || NULL == out_prev || NULL == sW || NULL == state || NULL == output) return;
    struct ffhip_engine *eng = flappie_hip_engine();
    if (NULL != eng)
        (void)check(ffhip_op_recurrent_step(eng, FFHIP_NET_LSTM5, view(x), view(out_prev), view(sW), view(state), view(output)), __func__);
}

/* layers.c:664-715 */
void grumod_step(const_flappie_matrix x, const_flappie_matrix istate, const_flappie_matrix sW, flappie_matrix xF, flappie_matrix ostate) {
    if (NULL == x || NULL == istate || NULL == sW || NULL == ostate) return;
    struct ffhip_engine *eng = flappie_hip_engine();
    if (NULL != eng)
        (void)check(ffhip_op_recurrent_step(eng, FFHIP_NET_GRUMOD5, view(x), view(istate), view(sW), view(NULL), view(ostate)), __func__);
}

/* layers.c:1029-1032 */
size_t nbase_from_flipflop_nparam(size_t nparam) {
    return (size_t)roundf((-1.0f + sqrtf(1 + 2 * nparam)) / 2.0f);
}

/* layers.c:1035-1079; NAN on failure */
double crf_manystay_partition_function(const_flappie_matrix C) {
    if (NULL == C) return NAN;
    struct ffhip_engine *eng = flappie_hip_engine();
    double logZ = NAN;
    if (NULL == eng || 0 != check(ffhip_op_partition_function(eng, view(C), &logZ), __func__)) return NAN;
    return logZ;
}

/* layers.c:1082-1106 */
flappie_matrix globalnorm_manystay(const_flappie_matrix X, const_flappie_matrix W, const_flappie_matrix b, float temperature, flappie_matrix C) {
    if (NULL == X) return NULL;
    if (NULL == W || NULL == b) { warnx("globalnorm: missing weights or bias"); return NULL; }
    struct ffhip_engine *eng = flappie_hip_engine();
    if (NULL == eng) return NULL;
    C = remake_flappie_matrix(C, W->nc, X->nc);
    if (NULL == C) return NULL;
    if (0 != check(ffhip_op_globalnorm_flipflop(eng, view(X), view(W), view(b), temperature, view(C)), __func__)) return free_flappie_matrix(C);
    return C;
}

flappie_matrix globalnorm_flipflop(const_flappie_matrix X, const_flappie_matrix W, const_flappie_matrix b, float temperature, flappie_matrix C) {
    return globalnorm_manystay(X, W, b, temperature, C);
}

/* ---- first-generation run-length head (layers.c:1115-1228) ---- */
size_t nbase_from_runlength_nparam(size_t nparam) { return nparam / 4; }

double runlength_partition_function(const_flappie_matrix C) {
    if (NULL == C) return NAN;
    struct ffhip_engine *eng = flappie_hip_engine();
    double logZ = NAN;
    if (NULL == eng || 0 != check(ffhip_op_runlength_partition_function_v1(eng, view(C), &logZ), __func__)) return NAN;
    return logZ;
}

flappie_matrix globalnorm_runlength(const_flappie_matrix X, const_flappie_matrix W, const_flappie_matrix b, float temperature, flappie_matrix C) {
    if (NULL == X) return NULL;
    if (NULL == W || NULL == b) { warnx("globalnorm_runlength: missing weights or bias"); return NULL; }
    struct ffhip_engine *eng = flappie_hip_engine();
    if (NULL == eng) return NULL;
    C = remake_flappie_matrix(C, W->nc, X->nc);
    if (NULL == C) return NULL;
    if (0 != check(ffhip_op_globalnorm_runlength_v1(eng, view(X), view(W), view(b), temperature, view(C)), __func__)) return free_flappie_matrix(C);
    return C;
}

/* ---- run-length head (layers.c:1230-1358) ---- */
size_t nbase_from_crf_runlength_nparam(size_t nparam) { return nbase_from_flipflop_nparam(nparam); }

double runlengthV2_partition_function(const_flappie_matrix C) {
    if (NULL == C) return NAN;
    struct ffhip_engine *eng = flappie_hip_engine();
    double logZ = NAN;
    if (NULL == eng || 0 != check(ffhip_op_runlength_partition_function(eng, view(C), &logZ), __func__)) return NAN;
    return logZ;
}

flappie_matrix globalnorm_runlengthV2(const_flappie_matrix X, const_flappie_matrix W, const_flappie_matrix b, float temperature, flappie_matrix C) {
    if (NULL == X) return NULL;
    if (NULL == W || NULL == b) { warnx("globalnorm_runlengthV2: missing weights or bias"); return NULL; }
    struct ffhip_engine *eng = flappie_hip_engine();
    if (NULL == eng) return NULL;
    C = remake_flappie_matrix(C, W->nc, X->nc);
    if (NULL == C) return NULL;
    if (0 != check(ffhip_op_globalnorm_runlength(eng, view(X), view(W), view(b), temperature, view(C)), __func__)) return free_flappie_matrix(C);
    return C;
}
