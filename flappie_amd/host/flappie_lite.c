/*  flappie_lite -- minimal C driver over the boundary: raw float32 signal files in, FASTQ out.
 *
 *  It is the reference's calculate_post (flappie.c:245-316) and fprintf_fastq
 *  (flappie_output.c:109-122) around the batched HIP engine: reads are prepared on the host exactly
 *  as flappie does (trim_and_segment_raw 200:10 / 100:0.0, medmad normalisation), grouped by trimmed
 *  length, and every group goes through ffhip as one batch.  fast5/HDF5 input, --trace output and
 *  the rest of the option table are the "next" rows N2/N3 of SURVEY.md section 8f.
 *
 *  usage: flappie_lite [--model NAME] [--temperature T] [--viterbi] [--no-trim] file.f32 ...
 *  (each file: little-endian float32 samples, already scaled to pA)
 */
#include <err.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/ffhip.h"
#include "../../include/flappie_common.h"
#include "../../include/networks.h"

typedef struct { char *name; raw_table rt; int done; } item;

static raw_table read_f32(const char *path) {
    raw_table rt = { 0 };
    FILE *fh = fopen(path, "rb");
    if (NULL == fh) { warn("%s", path); return rt; }
    fseek(fh, 0, SEEK_END);
    const long bytes = ftell(fh);
    fseek(fh, 0, SEEK_SET);
    const size_t n = bytes > 0 ? (size_t)bytes / sizeof(float) : 0;
    rt.raw = n ? malloc(n * sizeof(float)) : NULL;
    if (rt.raw && fread(rt.raw, sizeof(float), n, fh) == n) { rt.n = n; rt.start = 0; rt.end = n; }
    else { free(rt.raw); rt.raw = NULL; }
    fclose(fh);
    return rt;
}

int main(int argc, char *argv[]) {
    const char *model_name = "r941_native";
    float temperature = 1.0f;
    unsigned flags = 0;
    int trim = 1, first = 1;
    for (; first < argc && argv[first][0] == '-' && argv[first][1] == '-'; first++) {
        if (0 == strcmp(argv[first], "--model") && first + 1 < argc) model_name = argv[++first];
        else if (0 == strcmp(argv[first], "--temperature") && first + 1 < argc) temperature = atof(argv[++first]);
        else if (0 == strcmp(argv[first], "--viterbi")) flags |= FFHIP_RUN_VITERBI_ONLY;
        else if (0 == strcmp(argv[first], "--no-trim")) trim = 0;
        else errx(EXIT_FAILURE, "unknown option %s", argv[first]);
    }
    const enum model_type model = get_flappie_model_type(model_name);
    if (FLAPPIE_MODEL_INVALID == model || model >= FLAPPIE_MODEL_INVALID) errx(EXIT_FAILURE, "Invalid model \"%s\"", model_name);
    const struct ffhip_model *mdl = flappie_hip_model(model);
    if (NULL == mdl) errx(EXIT_FAILURE, "model %s is not available", model_name);
    struct ffhip_engine *eng = flappie_hip_engine();

    const int nfile = argc - first;
    item *items = calloc(nfile > 0 ? nfile : 1, sizeof(item));
    for (int i = 0; i < nfile; i++) {
        items[i].name = argv[first + i];
        raw_table rt = read_f32(argv[first + i]);
        if (rt.raw && trim) rt = trim_and_segment_raw(rt, 200, 10, 100, 0.0f);       /* flappie.c:105-108 defaults */
        if (rt.raw) medmad_normalise_array(rt.raw + rt.start, rt.end - rt.start);
        if (NULL == rt.raw) { warnx("No basecall returned for %s", items[i].name); items[i].done = 1; }
        items[i].rt = rt;
    }
    /* group by trimmed length: one batch per distinct length (results do not depend on batching) */
    raw_table *group = calloc(nfile > 0 ? nfile : 1, sizeof(raw_table));
    int *idx = calloc(nfile > 0 ? nfile : 1, sizeof(int));
    for (int i = 0; i < nfile; i++) {
        if (items[i].done) continue;
        const size_t len = items[i].rt.end - items[i].rt.start;
        int n = 0;
        for (int j = i; j < nfile; j++)
            if (!items[j].done && items[j].rt.end - items[j].rt.start == len) { group[n] = items[j].rt; idx[n++] = j; }
        ffhip_batch *b = ffhip_batch_create(eng, mdl, n, len);
        if (b && 0 == ffhip_batch_set_reads(b, group) && 0 == ffhip_batch_run(b, temperature, flags | FFHIP_RUN_NO_TRACE) &&
            0 == ffhip_batch_finish(b)) {
            const size_t nblock = ffhip_batch_nblock(b);
            for (int k = 0; k < n; k++) {
                size_t blen = 0;
                const char *bases = ffhip_batch_basecall(b, k, &blen);
                const raw_table *rt = &items[idx[k]].rt;
                /* flappie_output.c:112-116 header */
                printf("@%s  { \"filename\" : \"%s\", \"uuid\" : \"%s\", \"normalised_score\" : %f,  \"nblock\" : %zu,  \"sequence_length\" : %zu,  \"blocks_per_base\" : %f, \"nsample\" : %zu, \"trim\" : [ %zu, %zu ] }\n",
                       items[idx[k]].name, items[idx[k]].name, "", -ffhip_batch_score(b, k) / nblock, nblock, blen,
                       (float)nblock / (float)blen, rt->n, rt->start, rt->end);
                printf("%s\n+\n%s\n", bases, ffhip_batch_quality(b, k));
            }
        } else {
            warnx("batch failed: %s", ffhip_last_error());
        }
        if (b) ffhip_batch_destroy(b);
        for (int k = 0; k < n; k++) items[idx[k]].done = 1;
    }
    for (int i = 0; i < nfile; i++) free(items[i].rt.raw);
    free(items); free(group); free(idx);
    flappie_hip_shutdown();
    return EXIT_SUCCESS;
}
