/*  networks.c -- model registry and single-read network entry points (include/networks.h).
 *
 *  Mirrors /root/reference/src/networks.c:21-111,725-743 name for name.  The network bodies
 *  (networks.c:450-489 and :539-586: convolution -> 5 recurrent layers -> globalnorm_flipflop) are not
 *  evaluated here: this file only resolves the model and hands a batch of ONE read to the HIP engine
 *  (ffhip.h), which is what the reference's per-read loop becomes at batch size 1.
 */
#include <err.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/networks.h"
#include "../../include/ffhip.h"
#include "mdl_loader.h"

void flappie_matrix_remember_device_image(const_flappie_matrix mat);      /* flappie_matrix.c */

/* registry slot -> what the reference includes (networks.c:10-14) and the tensor name stems
 * (networks.c:218-361) */
typedef struct {
    const char *name;
    const char *description;
    int kind;                 /* ffhip_net_kind */
    const char *header;       /* file under $FLAPPIE_MODEL_DIR */
    const char *family;       /* "flipflop5" | "flipflop" */
    const char *ident;
} registry_entry;

static const registry_entry registry[] = {
    [FLAPPIE_MODEL_R941_NATIVE] = { "r941_native", "R9.4.1 model for MinION.  Trained from native DNA library",
                                    FFHIP_NET_LSTM5, "flipflop5_r941native.h", "flipflop5", "r941native" },
    [FLAPPIE_MODEL_R941_RNA002] = { "r941_rna002", "R9.4.1 dRNA model for MinION.  Trained from native and synthetic RNA library",
                                    FFHIP_NET_LSTM5, "flipflop5_r941rna002.h", "flipflop5", "r941rna002" },
    [FLAPPIE_MODEL_R941_5mC] = { "r941_5mC", "R9.4.1 model for PromethION; 5mC aware.  Trained from native NA12878 library",
                                 FFHIP_NET_GRUMOD5, "flipflop_r941native5mC.h", "flipflop", "r941native5mC" },
    [FLAPPIE_MODEL_R103_NATIVE] = { "r103_native", "R10.3 model for MinION.  Trained from native DNA library",
                                    FFHIP_NET_LSTM5, "flipflop5_r103native.h", "flipflop5", "r103native" },
    /* runnie's model (networks.c:14,364-399): the LSTM5 trunk with the run-length head */
    [RUNNIE_MODEL_R941_NATIVE] = { "rle_r941_native", "R9.4.1 run-length encoded model for MinION.  Trained from native DNA library",
                                   FFHIP_NET_LSTM5_RLE, "runlength5_r941native.h", "rle5", "r941native" },
};
#define NSLOT ((int)RUNNIE_MODEL_INVALID)
static int valid_model(int model) { return model >= 0 && model < NSLOT && NULL != registry[model].name; }

static struct {
    ffhip_engine *engine;
    ffhip_model *model[RUNNIE_MODEL_INVALID];
    int tried[RUNNIE_MODEL_INVALID];
} g;

/* networks.c:21-39 */
enum model_type get_flappie_model_type(const char *modelstr) {
    if (NULL == modelstr) return FLAPPIE_MODEL_INVALID;
    for (int i = 0; i < NSLOT; i++)
        if (valid_model(i) && 0 == strcmp(modelstr, registry[i].name)) return (enum model_type)i;
    return FLAPPIE_MODEL_INVALID;
}

/* networks.c:42-61 */
const char *flappie_model_string(const enum model_type model) {
    if (valid_model((int)model)) return registry[model].name;
    if (model == FLAPPIE_MODEL_INVALID || model == RUNNIE_MODEL_INVALID) errx(EXIT_FAILURE, "Invalid model  %s:%d", __FILE__, __LINE__);
    errx(EXIT_FAILURE, "Flappie enum failure -- report as bug. %s:%d \n", __FILE__, __LINE__);
    return NULL;
}

/* networks.c:64-83 */
const char *flappie_model_description(const enum model_type model) {
    if (valid_model((int)model)) return registry[model].description;
    if (model == FLAPPIE_MODEL_INVALID || model == RUNNIE_MODEL_INVALID) errx(EXIT_FAILURE, "Invalid Flappie model  %s:%d", __FILE__, __LINE__);
    errx(EXIT_FAILURE, "Flappie enum failure -- report as bug. %s:%d \n", __FILE__, __LINE__);
    return NULL;
}

/* networks.c:86-105 */
transition_function_ptr get_transition_function(const enum model_type model) {
    switch (model) {
    case FLAPPIE_MODEL_R941_NATIVE: return flipflop5_transitions_r941native;
    case FLAPPIE_MODEL_R941_RNA002: return flipflop5_transitions_r941rna002;
    case FLAPPIE_MODEL_R941_5mC: return flipflop_transitions_r941native5mC;
    case FLAPPIE_MODEL_R103_NATIVE: return flipflop5_transitions_r103native;
    case RUNNIE_MODEL_R941_NATIVE: return runlength5_transitions_r941native;
    case FLAPPIE_MODEL_INVALID:
    case RUNNIE_MODEL_INVALID:
        errx(EXIT_FAILURE, "Invalid Flappie model  %s:%d", __FILE__, __LINE__);
    default:
        errx(EXIT_FAILURE, "Flappie enum failure -- report as bug. %s:%d \n", __FILE__, __LINE__);
    }
    return NULL;
}

/* networks.c:108-111 */
flappie_matrix calculate_transitions(const raw_table signal, float temperature, enum model_type model) {
    transition_function_ptr transfun = get_transition_function(model);
    return transfun(signal, temperature);
}

struct ffhip_engine *flappie_hip_engine(void) {
    if (NULL == g.engine) {
        const char *dev = getenv("FLAPPIE_HIP_DEVICE");
        g.engine = ffhip_engine_create(dev ? atoi(dev) : 0);
        if (NULL == g.engine) warnx("HIP engine unavailable: %s", ffhip_last_error());
    }
    return g.engine;
}

static const_flappie_matrix need(const mdl_file *m, const char *fmt, const char *a, const char *b, const char *c, int *ok) {
    char name[MDL_NAME_MAX];
    snprintf(name, sizeof(name), fmt, a, b, c);
    const_flappie_matrix x = mdl_matrix(m, name);
    if (NULL == x) { warnx("model file lacks tensor %s", name); *ok = 0; }
    return x;
}

/* tensor names: networks.c:218-253 (flipflop5 / LSTM), :294-323 (flipflop / GRU), :364-399 (rle5 / LSTM) */
int flappie_hip_load_model(enum model_type model, const char *mdl_path) {
    if (!valid_model((int)model) || NULL == mdl_path) return -1;
    ffhip_engine *eng = flappie_hip_engine();
    if (NULL == eng) return -1;
    const registry_entry *r = &registry[model];
    mdl_file *m = mdl_load(mdl_path);
    if (NULL == m) return -1;
    ffhip_model_desc d;
    memset(&d, 0, sizeof(d));
    d.kind = r->kind;
    int ok = 1;
    char def[MDL_NAME_MAX];
    static const char *tags[5] = { "B1", "F2", "B3", "F4", "B5" };
    if (r->kind != FFHIP_NET_GRUMOD5) {
        d.nconv = 3;
        static const char *cn[3] = { "conv1", "conv2", "conv3" };
        for (int i = 0; i < 3; i++) {
            snprintf(def, sizeof(def), "%s_rnnrf_%s_%s_W", cn[i], r->family, r->ident);
            d.conv_W[i] = mdl_matrix(m, def);
            snprintf(def, sizeof(def), "%s_rnnrf_%s_%s_b", cn[i], r->family, r->ident);
            d.conv_b[i] = mdl_matrix(m, def);
            snprintf(def, sizeof(def), "%s_rnnrf_%s_%s_stride", cn[i], r->family, r->ident);
            d.conv_stride[i] = mdl_define(m, def, 0);
        }
        for (int i = 0; i < 3; i++) if (!d.conv_W[i] || !d.conv_b[i] || d.conv_stride[i] <= 0) { warnx("%s: convolution %d incomplete", mdl_path, i + 1); ok = 0; }
    } else {
        d.nconv = 1;
        snprintf(def, sizeof(def), "conv_rnnrf_%s_%s_W", r->family, r->ident);
        d.conv_W[0] = mdl_matrix(m, def);
        snprintf(def, sizeof(def), "conv_rnnrf_%s_%s_b", r->family, r->ident);
        d.conv_b[0] = mdl_matrix(m, def);
        snprintf(def, sizeof(def), "conv_rnnrf_%s_%s_stride", r->family, r->ident);
        d.conv_stride[0] = mdl_define(m, def, 0);
        if (!d.conv_W[0] || !d.conv_b[0] || d.conv_stride[0] <= 0) { warnx("%s: convolution incomplete", mdl_path); ok = 0; }
    }
    const char *cell = (r->kind != FFHIP_NET_GRUMOD5) ? "lstm" : "gru";
    for (int i = 0; i < 5; i++) {
        char stem[MDL_NAME_MAX];
        snprintf(stem, sizeof(stem), "%s%s_rnnrf_%s_%s_", cell, tags[i], r->family, r->ident);
        d.rnn_iW[i] = need(m, "%s%s%s", stem, "iW", "", &ok);
        d.rnn_sW[i] = need(m, "%s%s%s", stem, "sW", "", &ok);
        d.rnn_b[i] = need(m, "%s%s%s", stem, "b", "", &ok);
    }
    {
        char stem[MDL_NAME_MAX];
        snprintf(stem, sizeof(stem), "FF_rnnrf_%s_%s_", r->family, r->ident);
        d.FF_W = need(m, "%s%s%s", stem, "W", "", &ok);
        d.FF_b = need(m, "%s%s%s", stem, "b", "", &ok);
    }
    ffhip_model *dev = NULL;
    if (ok) {
        dev = ffhip_model_upload(eng, &d);
        if (NULL == dev) warnx("%s: %s", mdl_path, ffhip_last_error());
    }
    mdl_free(m);                      /* weights now live in HBM */
    if (NULL == dev) return -1;
    if (g.model[model]) ffhip_model_free(g.model[model]);
    g.model[model] = dev;
    g.tried[model] = 1;
    return 0;
}

const struct ffhip_model *flappie_hip_model(enum model_type model) {
    if (!valid_model((int)model)) return NULL;
    if (NULL == g.model[model] && !g.tried[model]) {
        g.tried[model] = 1;
        const char *dir = getenv("FLAPPIE_MODEL_DIR");
        if (NULL == dir) {
            warnx("model %s: set FLAPPIE_MODEL_DIR to the directory holding %s (the reference's src/models)", registry[model].name, registry[model].header);
            return NULL;
        }
        char path[4096];
        snprintf(path, sizeof(path), "%s/%s", dir, registry[model].header);
        g.tried[model] = 0;
        if (0 != flappie_hip_load_model(model, path)) { g.tried[model] = 1; return NULL; }
    }
    return g.model[model];
}

/* FLAPPIE_REPORT_COPIES=1: the library's host<->device copies of the whole run on stderr at exit (tests/test_cli.py counts what a
 * relinked flappie.c moves per read: the signal up; path, qualities' scores and trace down; never a matrix) */
__attribute__((destructor)) static void report_copies(void) {
    if (NULL == getenv("FLAPPIE_REPORT_COPIES")) return;
    unsigned long long c[5];
    ffhip_copy_counts(c, 0);
    fprintf(stderr, "ffhip copies: h2d %llu calls %llu bytes, d2h %llu calls %llu bytes, largest d2h %llu bytes\n", c[0], c[1], c[2], c[3], c[4]);
}

void flappie_hip_shutdown(void) {
    for (int i = 0; i < NSLOT; i++) {
        if (g.model[i]) ffhip_model_free(g.model[i]);
        g.model[i] = NULL;
        g.tried[i] = 0;
    }
    if (g.engine) ffhip_engine_destroy(g.engine);
    g.engine = NULL;
}

/* the common body of networks.c:725-739: NULL for an empty signal (networks.c:540-541) */
static flappie_matrix transitions_for(const raw_table signal, float temperature, enum model_type model) {
    if (0 == signal.n || NULL == signal.raw || signal.end <= signal.start) return NULL;
    const ffhip_model *mdl = flappie_hip_model(model);
    if (NULL == mdl) return NULL;
    ffhip_batch *b = ffhip_batch_create(g.engine, mdl, 1, signal.end - signal.start);
    if (NULL == b) { warnx("%s", ffhip_last_error()); return NULL; }
    flappie_matrix trans = NULL;
    if (0 == ffhip_batch_set_reads(b, &signal) && 0 == ffhip_batch_run(b, temperature, FFHIP_RUN_NO_DECODE) &&
        0 == ffhip_batch_finish(b)) {
        const size_t P = ffhip_model_nparam(mdl), nblock = ffhip_batch_nblock(b);
        trans = make_flappie_matrix(P, nblock);
        if (trans && 0 != ffhip_matrix_policy()) {
            /* the scores stay in HBM: a device-to-device copy into the image the matrix owns (dev_state 2); transpost_crf_flipflop,
             * decode_crf_flipflop, exp_activation_inplace and trace_from_posterior (flappie.c:266-300) then never move a matrix */
            const ffhip_mat v = { trans->data.f, trans->nr, trans->nc, trans->stride, &trans->dev, &trans->dev_state };
            if (0 != ffhip_batch_transitions_to(b, 0, v)) { warnx("%s", ffhip_last_error()); trans = free_flappie_matrix(trans); }
            else flappie_matrix_remember_device_image(trans);      /* flappie.c:281 frees this matrix with a plain free() */
        } else if (trans) {
            float *tmp = malloc(P * nblock * sizeof(float));
            if (tmp && 0 == ffhip_batch_get_transitions(b, 0, tmp)) {
                for (size_t c = 0; c < nblock; c++) memcpy(trans->data.f + c * trans->stride, tmp + c * P, P * sizeof(float));
            } else {
                trans = free_flappie_matrix(trans);
            }
            free(tmp);
        }
    } else {
        warnx("%s", ffhip_last_error());
    }
    ffhip_batch_destroy(b);
    return trans;
}

flappie_matrix flipflop5_transitions_r941native(const raw_table signal, float temperature) {
    return transitions_for(signal, temperature, FLAPPIE_MODEL_R941_NATIVE);
}
flappie_matrix flipflop5_transitions_r941rna002(const raw_table signal, float temperature) {
    return transitions_for(signal, temperature, FLAPPIE_MODEL_R941_RNA002);
}
flappie_matrix flipflop_transitions_r941native5mC(const raw_table signal, float temperature) {
    return transitions_for(signal, temperature, FLAPPIE_MODEL_R941_5mC);
}
flappie_matrix flipflop5_transitions_r103native(const raw_table signal, float temperature) {
    return transitions_for(signal, temperature, FLAPPIE_MODEL_R103_NATIVE);
}
/* networks.c:740-743 */
flappie_matrix runlength5_transitions_r941native(const raw_table signal, float temperature) {
    return transitions_for(signal, temperature, RUNNIE_MODEL_R941_NATIVE);
}
