/*  flappie -- command line of the MI355X flip-flop basecaller.
 *
 *  Same option table, defaults, file/directory globbing and per-read output as the reference's
 *  src/flappie.c (options :42-67, defaults :93-112, parse :127-239, main :319-399).  The body differs where
 *  it has to: the reference calls calculate_post() per file (flappie.c:371); here files are read and
 *  prepared on the host, grouped by trimmed length (the network never pads or splits a read) and sent to
 *  the HIP engine in batches; records are written in input order.
 */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE      /* sched_setaffinity, CPU_SET (bind_to_gpu_numa) */
#endif
#include <sched.h>
#include <argp.h>
#include <assert.h>
#include <dirent.h>
#include <err.h>
#include <sys/stat.h>
#include <glob.h>
#include <libgen.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <pthread.h>
#include <semaphore.h>
#include <string.h>
#include <time.h>
#include <strings.h>
#include <errno.h>
#include <fcntl.h>
#include <signal.h>
#include <sys/wait.h>
#include <unistd.h>
#include <execinfo.h>

#include "../../include/decode.h"
#include "../../include/fast5_interface.h"
#include "../../include/ffhip.h"
#include "../../include/flappie_common.h"
#include "../../include/flappie_output.h"
#include "../../include/networks.h"

const char *argp_program_version = "flappie (MI355X/HIP) 0.1, interface of flappie 2.1.3";
const char *argp_program_bug_address = "<this repository>";
#ifdef BUILD_RUNNIE
static char doc[] = "Runnie basecaller -- basecall from raw signal";
#else
static char doc[] = "Flappie basecaller -- basecall from raw signal";
#endif
static char args_doc[] = "fast5 [fast5 ...]";
static struct argp_option options[] = {
    {"delta", 'd', "factor", 0, "Using delta samples model with scaling factor"},
    {"format", 'f', "format", 0, "Format to output reads (FASTA or SAM)"},
    {"limit", 'l', "nreads", 0, "Maximum number of reads to call (0 is unlimited)"},
    {"model", 'm', "name", 0, "Model to use (\"help\" to list)"},
    {"output", 'o', "filename", 0, "Write to file rather than stdout"},
    {"prefix", 'p', "string", 0, "Prefix to append to name of each read"},
    {"reverse", 'r', 0, 0, "Reverse output base calls"},
    {"no-reverse", 6, 0, OPTION_ALIAS, "Don't reverse output base calls"},
    {"temperature", 7, "factor", 0, "Temperature for weights"},
    {"trim", 't', "start:end", 0, "Number of samples to trim, as start:end"},
    {"trace", 'T', "filename", 0, "Dump trace to HDF5 file"},
    {"licence", 10, 0, 0, "Print licensing information"},
    {"license", 11, 0, OPTION_ALIAS, "Print licensing information"},
    {"segmentation", 3, "chunk:percentile", 0, "Chunk size and percentile for variance based segmentation"},
    {"viterbi", 'v', 0, 0, "Use viterbi decoding only"},
    {"no-viterbi", 8, 0, OPTION_ALIAS, "Use forward-backward followed by viterbi"},
    {"fb", 9, 0, OPTION_ALIAS, "Use forward-backward followed by viterbi"},
    {"hdf5-compression", 12, "level", 0, "Gzip compression level for HDF5 output (0:off, 1: quickest, 9: best)"},
    {"hdf5-chunk", 13, "size", 0, "Chunk size for HDF5 output"},
    {"uuid", 14, 0, 0, "Output UUID"},
    {"no-uuid", 15, 0, OPTION_ALIAS, "Output read file"},
    {"batch", 16, "nreads", 0, "Reads per GPU batch (default: what one layer launch takes -- 1024 at up to 256 hidden units, 512 up to 384, else 256)"},
    {"shard", 18, "g/n", 0, "Call only files g, g+n, g+2n, ... of the sorted input list (one process per GPU: tools/flappie_multi_gpu.sh)"},
    {"readers", 17, "n", 0, "fast5 reader processes feeding the GPU (default 4: one keeps up with ~50 Msamples/s of files that need libhdf5 and > 200 of those host/fast5_raw.c reads; more only cost CPU; 0 reads in this process)"},
    {"shard-by-size", 19, 0, 0, "With --shard: deal the files to the n shards by size (largest first, each to the lightest shard) instead of by index"},
    {0}
};

#ifdef BUILD_RUNNIE
/* runnie.c:71: the run-length model; its --format/--model/--trace/--reverse options are commented out in the
 * reference (runnie.c:42-60) and are ignored here with a warning */
#define DEFAULT_MODEL RUNNIE_MODEL_R941_NATIVE
#else
#define DEFAULT_MODEL FLAPPIE_MODEL_R941_NATIVE
#endif

static struct {
    int compression_level, compression_chunk_size;
    float delta;
    char *trace;
    enum flappie_outformat_type outformat;
    int limit;
    enum model_type model;
    FILE *output;
    char *prefix;
    bool reverse;
    float temperature;
    int trim_start, trim_end, varseg_chunk;
    float varseg_thresh;
    bool viterbi_only;
    char **files;
    bool uuid;
    int batch;
    int readers;
    int shard, nshard;
    bool shard_by_size;
} args = { 1, 200, 0.0f, NULL, FLAPPIE_OUTFORMAT_FASTQ, 0, DEFAULT_MODEL, NULL, "", false, 1.0f, 200, 10, 100, 0.0f, false, NULL, true, 0, 4, 0, 0 };      /* batch 0: by model (below); nshard 0: --shard not given */

static void print_models(FILE *fh) {
    for (int mdl = 0; mdl < (int)flappie_nmodel; mdl++)
        fprintf(fh, "%10s : %s  %s\n", flappie_model_string(mdl), flappie_model_description(mdl), (DEFAULT_MODEL == mdl) ? "(default)" : "");
}

static error_t parse_arg(int key, char *arg, struct argp_state *state) {
    char *next_tok = NULL;
    switch (key) {
    case 'd': args.delta = atof(arg); break;
    case 'f':
        args.outformat = get_outformat(arg);
        if (FLAPPIE_OUTFORMAT_INVALID == args.outformat) errx(EXIT_FAILURE, "Unrecognised output format \"%s\".", arg);
        break;
    case 'l': args.limit = atoi(arg); break;
    case 'm':
#ifdef BUILD_RUNNIE
        /* runnie has one model (runnie.c:71); its --model option is commented out in the reference (runnie.c:42-60): a
         * flip-flop model name here must not reach the run-length record formatter */
        warnx("--model is ignored by runnie (always %s)", flappie_model_string(DEFAULT_MODEL));
        break;
#endif
        if (0 == strcasecmp(arg, "help")) { print_models(stdout); exit(EXIT_SUCCESS); }
        args.model = get_flappie_model_type(arg);
        if (FLAPPIE_MODEL_INVALID == args.model || args.model > FLAPPIE_MODEL_INVALID) {
            fprintf(stdout, "Invalid Flappie model \"%s\".\n", arg);
            print_models(stdout);
            exit(EXIT_FAILURE);
        }
        break;
    case 'o':
        args.output = fopen(arg, "w");
        if (NULL == args.output) errx(EXIT_FAILURE, "Failed to open \"%s\" for output.", arg);
        break;
    case 'p': args.prefix = arg; break;
    case 'r': args.reverse = true; break;
    case 't':
        args.trim_start = atoi(strtok(arg, ":"));
        next_tok = strtok(NULL, ":");
        args.trim_end = (NULL != next_tok) ? atoi(next_tok) : args.trim_start;
        if (args.trim_start < 0 || args.trim_end < 0) errx(EXIT_FAILURE, "--trim values must be non-negative");
        break;
    case 'T': args.trace = arg; break;
    case 'v': args.viterbi_only = true; break;
    case 3:
        args.varseg_chunk = atoi(strtok(arg, ":"));
        next_tok = strtok(NULL, ":");
        if (NULL == next_tok) errx(EXIT_FAILURE, "--segmentation should be of form chunk:percentile");
        args.varseg_thresh = atof(next_tok) / 100.0;
        if (args.varseg_chunk < 2 || !(args.varseg_thresh >= 0.0f && args.varseg_thresh < 1.0f)) errx(EXIT_FAILURE, "--segmentation out of range");
        break;
    case 6: args.reverse = false; break;
    case 7:
        args.temperature = atof(arg);
        if (!(isfinite(args.temperature) && args.temperature > 0.0f)) errx(EXIT_FAILURE, "--temperature must be positive");
        break;
    case 8:
    case 9: args.viterbi_only = false; break;
    case 10:
    case 11:
        /* The reference prints Oxford Nanopore's licence text (flappie_licence.h).  This program is an
         * independent implementation of the same interface; it points at the respective licence files. */
        puts("This MI355X/HIP basecaller is an independent implementation of the flappie command line interface.\n"
             "The flappie reference implementation and its models are (c) Oxford Nanopore Technologies, Ltd. and are\n"
             "distributed under the Oxford Nanopore Technologies, Ltd. Public License v1.0 (see LICENCE.txt of\n"
             "https://github.com/nanoporetech/flappie); models loaded by this program remain under that licence.");
        exit(EXIT_SUCCESS);
    case 12:
        args.compression_level = atoi(arg);
        if (args.compression_level < 0 || args.compression_level > 9) errx(EXIT_FAILURE, "--hdf5-compression must be 0..9");
        break;
    case 13:
        args.compression_chunk_size = atoi(arg);
        if (args.compression_chunk_size <= 0) errx(EXIT_FAILURE, "--hdf5-chunk must be positive");
        break;
    case 14: args.uuid = true; break;
    case 15: args.uuid = false; break;
    case 16:
        args.batch = atoi(arg);
        if (args.batch <= 0) errx(EXIT_FAILURE, "--batch must be positive");
        break;
    case 17:
        args.readers = atoi(arg);
        if (args.readers < 0 || args.readers > 64) errx(EXIT_FAILURE, "--readers must be between 0 and 64");
        break;
    case 18:
        if (2 != sscanf(arg, "%d/%d", &args.shard, &args.nshard) || args.nshard < 1 || args.shard < 0 || args.shard >= args.nshard)
            errx(EXIT_FAILURE, "--shard takes g/n with 0 <= g < n");
        break;
    case 19: args.shard_by_size = true; break;
    case ARGP_KEY_NO_ARGS: argp_usage(state); break;
    case ARGP_KEY_ARG:
        args.files = &state->argv[state->next - 1];
        state->next = state->argc;
        break;
    default: return ARGP_ERR_UNKNOWN;
    }
    return 0;
}

static struct argp argp = { options, parse_arg, args_doc, doc };

/* this libhdf5 is not built thread-safe: every HDF5 call of either thread is made under this lock */
static pthread_mutex_t hdf5_lock = PTHREAD_MUTEX_INITIALIZER;

/* FLAPPIE_CLI_TIMING=1: wall-clock split of the driver's phases on stderr at exit */
static double t_phase[8];
static const char *phase_name[8] = { "fast5 read", "signal preparation", "batch create/destroy", "upload+network+decode", "fetch results", "write output",
                                     "  of which set_prepared", "  of which batch_run" };
/* Development switches of the binary: FLAPPIE_DEBUG=token[,token=value ...] -- no_reader_thread, no_writer_thread,
 * list_only, kill_reader=k:f, no_numa_bind, sysfs_root=DIR.  NULL when the token is absent, its value ("" for a bare token) otherwise. */
static const char *cli_dbg(const char *token) {
    static char buf[256];
    const char *e = getenv("FLAPPIE_DEBUG");
    const size_t n = strlen(token);
    while (e && *e) {
        const char *end = e + strcspn(e, ", ;");
        if ((size_t)(end - e) >= n && 0 == strncmp(e, token, n) && (e + n == end || '=' == e[n])) {
            const size_t vl = (e + n == end) ? 0 : (size_t)(end - e - n - 1);
            if (vl >= sizeof buf) return NULL;
            memcpy(buf, e + n + 1, vl);
            buf[vl] = 0;
            return buf;
        }
        e = *end ? end + 1 : end;
    }
    return NULL;
}

/* This process, the reader children it is about to fork and the pinned staging they fill, on the CPUs of the GPU's NUMA node (eight
 * ranks of a node otherwise read their files wherever the scheduler puts them).  The node comes from sysfs -- the HIP runtime must not be up before the
 * fork --: the device-th render node of vendor 0x1002 in minor order, the device number first taken through the *_VISIBLE_DEVICES variables (physical_gpu_index).  Left alone when the
 * node is unknown or none of its CPUs is in this process's set; FLAPPIE_DEBUG=no_numa_bind switches it off, sysfs_root=DIR is for the tests. */
/* HIP device number of this process -> position among the node's GPUs in the kernel's order, through HIP_VISIBLE_DEVICES (or its synonym
 * CUDA_VISIBLE_DEVICES) and ROCR_VISIBLE_DEVICES below it; -1 when an entry is not a plain number (a UUID) or the index is beyond the list -- the process
 * is then left unbound: several ranks pinned to ONE socket by a wrong guess are worse off than unbound ones (shard.py physical_gpu_index) */
static int physical_gpu_index(int device) {
    const char *hip = getenv("HIP_VISIBLE_DEVICES");
    const char *names[2] = { (hip && hip[0]) ? "HIP_VISIBLE_DEVICES" : "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES" };
    for (int v = 0; v < 2; v++) {
        const char *text = getenv(names[v]);
        if (NULL == text || 0 == text[strspn(text, " ")]) continue;
        const char *p = text;
        for (int k = 0; k < device; k++) {
            p = strchr(p, ',');
            if (NULL == p) return -1;
            p++;
        }
        p += strspn(p, " ");
        char *end;
        const long val = strtol(p, &end, 10);
        if (end == p || (end[strspn(end, " ")] != ',' && end[strspn(end, " ")] != 0) || val < 0) return -1;
        device = (int)val;
    }
    return device;
}

static int bind_to_gpu_numa(int device) {
    device = physical_gpu_index(device);
    if (device < 0) return -1;
    const char *root = cli_dbg("sysfs_root");
    char sys[256], path[512], buf[4096];
    snprintf(sys, sizeof sys, "%s", (root && root[0]) ? root : "/sys");
    int minors[64], nodes[64], n = 0;
    snprintf(path, sizeof path, "%s/class/drm", sys);
    DIR *d = opendir(path);
    if (NULL == d) return -1;
    for (struct dirent *e; n < 64 && NULL != (e = readdir(d));) {
        int minor;
        if (1 != sscanf(e->d_name, "renderD%d", &minor)) continue;
        snprintf(path, sizeof path, "%s/class/drm/%s/device/vendor", sys, e->d_name);
        FILE *fh = fopen(path, "r");
        if (NULL == fh) continue;
        const int amd = (NULL != fgets(buf, sizeof buf, fh) && 0 == strncasecmp(buf, "0x1002", 6));
        fclose(fh);
        if (!amd) continue;
        snprintf(path, sizeof path, "%s/class/drm/%s/device/numa_node", sys, e->d_name);
        int node = -1;
        if (NULL != (fh = fopen(path, "r"))) { if (1 != fscanf(fh, "%d", &node)) node = -1; fclose(fh); }
        int k = n++;
        while (k > 0 && minors[k - 1] > minor) { minors[k] = minors[k - 1]; nodes[k] = nodes[k - 1]; k--; }
        minors[k] = minor; nodes[k] = node;
    }
    closedir(d);
    if (device < 0 || device >= n || nodes[device] < 0) return -1;
    snprintf(path, sizeof path, "%s/devices/system/node/node%d/cpulist", sys, nodes[device]);
    FILE *fh = fopen(path, "r");
    if (NULL == fh) return -1;
    const int got = NULL != fgets(buf, sizeof buf, fh);
    fclose(fh);
    if (!got) return -1;
    cpu_set_t mine, want;
    if (0 != sched_getaffinity(0, sizeof mine, &mine)) return -1;
    CPU_ZERO(&want);
    int nbound = 0;
    for (char *p = buf; *p && *p != '\n';) {
        char *end;
        const long lo = strtol(p, &end, 10);
        long hi = lo;
        if (end == p) break;
        if ('-' == *end) { p = end + 1; hi = strtol(p, &end, 10); }
        for (long c = lo; c <= hi && c < CPU_SETSIZE; c++) if (CPU_ISSET((int)c, &mine)) { CPU_SET((int)c, &want); nbound++; }
        p = (',' == *end) ? end + 1 : end;
    }
    if (0 == nbound || 0 != sched_setaffinity(0, sizeof want, &want)) return -1;
    if (getenv("FLAPPIE_CLI_TIMING")) fprintf(stderr, "bound to %d CPUs of NUMA node %d (GPU %d)\n", nbound, nodes[device], device);
    return nodes[device];
}

static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
static double t_program_start;          /* main's first statement (FLAPPIE_DEBUG=pack_log's time stamps) */
/* what the run basecalled: reads, samples of their trimmed ranges (the metric of SURVEY.md section 8d), samples read from the files */
static unsigned long long n_called_reads, n_called_samples, n_raw_samples;
/* what the batches cost: a batch (a launch per layer) takes as long as its longest read needs whatever the others' lengths, a read tile of 16 as long as
 * ITS longest read holds its workgroups -- samples submitted against samples x slots paid for, at both grains (FLAPPIE_CLI_TIMING prints them) */
static unsigned long long n_batches, n_packed_batches, n_batch_samples, n_batch_slot_samples, n_tile_slot_samples;
static int reader_failures;              /* reader children that ended abnormally: the exit status says so */

typedef struct {
    char *filename;                     /* owned */
    struct _raw_basecall_info res;      /* rt filled by read_raw, start/end by the preparation; basecall == NULL until called */
    int prepared;                       /* index into the chunk's ffhip_prep, or -1 */
    char *rle_text;                     /* runnie: the read's records, formatted (runnie.c:282-313) */
} item;

/* Batch objects own gigabytes of workspace; creating one per group costs more than running it.  Full-size groups
 * (--batch reads) share one cached object whose capacity grows when a longer read turns up. */
#define NINFLIGHT 3                    /* batches submitted and not collected, at most (FLAPPIE_INFLIGHT=3; default 2) */
static struct { ffhip_batch *b; int nread; size_t cap; } batch_cache[NINFLIGHT];

/* *nslot = reads the returned batch was created for: args.batch for the cached objects (groups of at least a quarter of
 * that are padded with empty slots), the group's own size for small groups */
static ffhip_batch *acquire_batch(struct ffhip_engine *eng, const struct ffhip_model *mdl, int n, size_t len, int slot, int *cached, int *nslot) {
    *cached = 0;
    *nslot = n;
    if (4 * n < args.batch) return ffhip_batch_create(eng, mdl, n, len);
    if (NULL == batch_cache[slot].b || batch_cache[slot].cap < len) {
        if (batch_cache[slot].b) ffhip_batch_destroy(batch_cache[slot].b);
        batch_cache[slot].cap = len + len / 8;
        batch_cache[slot].nread = args.batch;
        batch_cache[slot].b = ffhip_batch_create(eng, mdl, args.batch, batch_cache[slot].cap);
        if (NULL == batch_cache[slot].b) { batch_cache[slot].cap = 0; return NULL; }
    }
    *cached = 1;
    *nslot = args.batch;
    return batch_cache[slot].b;
}

/* Packed batches (ffhip.h "packed batches"): reads of mixed lengths, several to a row.  One cached object per pipeline slot, `--batch` rows of the chunk's row
 * capacity, created anew when a chunk needs longer rows. */
static struct { ffhip_batch *b; size_t cap; int max_reads, rows, full, single; } pack_cache[NINFLIGHT];
static ffhip_batch *acquire_packed(struct ffhip_engine *eng, const struct ffhip_model *mdl, int rows, int rows_full, size_t cap, int max_reads, int slot, int single) {
    /* (an object with MORE rows or longer rows than this batch needs serves it as it is: rows without a read cost nothing, and re-creating a 90 GB object is seconds) */
    if (NULL == pack_cache[slot].b || pack_cache[slot].cap < cap || pack_cache[slot].max_reads < max_reads || pack_cache[slot].rows < rows) {
        if (pack_cache[slot].b) ffhip_batch_destroy(pack_cache[slot].b);
        /* two shapes of object only, so that a slot's object is re-created when the reads get LONGER and for nothing else: a quarter of the rows for a run's first,
         * small chunk (created in a fraction of a second, replaced by the second chunk), all the rows the memory takes otherwise; capacities in pack_row_cap's steps */
        const int make_rows = (4 * rows <= rows_full) ? (rows_full / 4 + 15) / 16 * 16 : rows_full;
        pack_cache[slot].cap = cap;
        pack_cache[slot].max_reads = max_reads;
        pack_cache[slot].rows = make_rows;
        pack_cache[slot].full = (make_rows == rows_full);
        pack_cache[slot].single = single;
        const double tc0 = now_s();
        pack_cache[slot].b = cli_dbg("pack_fail") ? NULL : ffhip_batch_create_packed(eng, mdl, make_rows, cap, max_reads);      /* (pack_fail: tests -- as if the device had no memory for it) */
        if (getenv("FLAPPIE_CLI_TIMING")) fprintf(stderr, "packed batch object (slot %d): %d rows of %zu samples, up to %d reads%s, %.2f s\n", slot, make_rows, cap, max_reads,
                                                  pack_cache[slot].b ? "" : " -- FAILED", now_s() - tc0);
        if (NULL == pack_cache[slot].b) { pack_cache[slot].cap = 0; return NULL; }
    }
    return pack_cache[slot].b;
}
/* row capacities in steps of a quarter octave (a batch object is re-created -- seconds of hipMalloc at these sizes -- only when a chunk needs the next step) */
static size_t pack_row_cap(size_t want) {
    size_t c = 4096;
    while (c < want) c = (c + c / 4 + 1023) & ~(size_t)1023;
    return c;
}

/* A group of prepared reads in flight: submitted (upload + network + decode enqueued on the batch's stream), collected
 * later (flappie.c:264-316 after normalisation) -- so the host side of the next group overlaps the GPU side of this one. */
struct chunk_ctx;
typedef struct { ffhip_batch *b; int cached, n, *idx; item **its; const ffhip_prep *prep; struct chunk_ctx *owner; } pending_batch;

/* n reads in the rows of one packed batch: slot_of / off_of from ffhip_pack_plan, `cap` the row capacity it was made for */
static pending_batch submit_packed(struct ffhip_engine *eng, const struct ffhip_model *mdl, const ffhip_prep *prep, item **its, int n, const int *slot_of, const int *off_of,
                                   int rows, int rows_full, size_t cap, size_t cap_obj, int max_reads, int slot, int single) {
    pending_batch pb = { NULL, 1, n, malloc((n > 0 ? n : 1) * sizeof(int)), malloc((n > 0 ? n : 1) * sizeof(item *)), prep, NULL };
    memcpy(pb.its, its, n * sizeof(item *));
    size_t longest = 0;
    unsigned long long samples = 0;
    size_t *row_end = calloc(rows, sizeof(size_t));
    for (int i = 0; i < n; i++) {
        pb.idx[i] = its[i]->prepared;
        const size_t li = its[i]->res.rt.end - its[i]->res.rt.start;
        samples += li;
        if (row_end) { const size_t e = (size_t)off_of[i] + ffhip_model_nblock(mdl, li); if (e > row_end[slot_of[i]]) row_end[slot_of[i]] = e; }
    }
    /* what the batch costs: its longest ROW (in blocks -> samples of the trimmed signal), whatever the others hold */
    const size_t nb_cap = ffhip_model_nblock(mdl, cap);
    const size_t spb = nb_cap ? (cap + nb_cap / 2) / nb_cap : 1;      /* samples a block (the model's stride) */
    for (int r = 0; row_end && r < rows; r++) {
        if (row_end[r] > longest) longest = row_end[r];
    }
    n_batches++; n_packed_batches++;
    n_batch_samples += samples;
    n_batch_slot_samples += (unsigned long long)longest * spb * (unsigned long long)(16 * ((rows + 15) / 16));
    if (cli_dbg("pack_log"))             /* development: what every packed batch holds and pays for */
        fprintf(stderr, "packed batch %llu (submitted at %.3f s): %d reads, %llu samples in %d rows planned for %zu samples (object: %d rows of %zu), longest row %zu samples, longest read %zu: fill %.3f\n", n_packed_batches, now_s() - t_program_start, n, samples,
                rows, cap, rows_full, cap_obj, longest * spb, (size_t)(its[0]->res.rt.end - its[0]->res.rt.start), (double)samples / ((double)(longest * spb) * (double)(16 * ((rows + 15) / 16))));
    for (int r0 = 0; row_end && r0 < rows; r0 += 16) {
        size_t lt = 0;
        for (int r = r0; r < rows && r < r0 + 16; r++) if (row_end[r] > lt) lt = row_end[r];
        n_tile_slot_samples += 16ull * lt * spb;
    }
    free(row_end);
    double t0 = now_s();
    pb.b = acquire_packed(eng, mdl, rows, rows_full, cap_obj, max_reads, slot, single);      /* (the object's rows come in steps; the plan was made for rows of `cap` samples) */
    t_phase[2] += now_s() - t0; t0 = now_s();
    const unsigned flags = (args.viterbi_only ? FFHIP_RUN_VITERBI_ONLY : 0u) | (args.trace ? 0u : FFHIP_RUN_NO_TRACE);
    if (getenv("FLAPPIE_CLI_TIMING")) fprintf(stderr, "packed batch: %d reads, %.1f Msamples in %d rows planned for %zu samples (longest row %zu)\n", n, (double)samples / 1e6, rows, cap, longest * spb);
    int rc_sub = (NULL == pb.b) ? -1 : ffhip_batch_set_prepared_packed(pb.b, prep, n, pb.idx, slot_of, off_of);
    t_phase[6] += now_s() - t0;
    const double t1 = now_s();
    if (0 == rc_sub) rc_sub = ffhip_batch_run(pb.b, args.temperature, flags);
    t_phase[7] += now_s() - t1;
    if (0 != rc_sub) { warnx("%s", ffhip_last_error()); pb.b = NULL; }
    t_phase[3] += now_s() - t0;
    return pb;
}

static pending_batch submit_batch(struct ffhip_engine *eng, const struct ffhip_model *mdl, const ffhip_prep *prep, item **its, int n, int slot) {
    const int nmax = (n > args.batch) ? n : args.batch;
    pending_batch pb = { NULL, 0, n, malloc(nmax * sizeof(int)), malloc(n * sizeof(item *)), prep, NULL };
    memcpy(pb.its, its, n * sizeof(item *));
    for (int i = 0; i < nmax; i++) pb.idx[i] = (i < n) ? its[i]->prepared : -1;             /* -1: empty slot */
    size_t len = 0;                                          /* capacity = the longest read of the (sorted) group */
    for (int i = 0; i < n; i++) {
        const size_t li = its[i]->res.rt.end - its[i]->res.rt.start;
        if (li > len) len = li;
        n_batch_samples += li;
    }
    n_batches++;
    n_batch_slot_samples += (unsigned long long)len * (unsigned long long)(16 * ((nmax + 15) / 16));
    for (int i = 0; i < n; i += 16) {
        size_t lt = 0;
        for (int k = i; k < n && k < i + 16; k++) { const size_t lk = its[k]->res.rt.end - its[k]->res.rt.start; if (lk > lt) lt = lk; }
        n_tile_slot_samples += 16ull * lt;
    }
    double t0 = now_s();
    int nslot = n;
    pb.b = acquire_batch(eng, mdl, n, len, slot, &pb.cached, &nslot);
    (void)nslot;
    t_phase[2] += now_s() - t0; t0 = now_s();
    const unsigned flags = (args.viterbi_only ? FFHIP_RUN_VITERBI_ONLY : 0u) | (args.trace ? 0u : FFHIP_RUN_NO_TRACE);
    int rc_sub = (NULL == pb.b) ? -1 : ffhip_batch_set_prepared(pb.b, prep, pb.idx);
    t_phase[6] += now_s() - t0;
    const double t1 = now_s();
    if (0 == rc_sub) rc_sub = ffhip_batch_run(pb.b, args.temperature, flags);
    t_phase[7] += now_s() - t1;
    if (0 != rc_sub) {
        warnx("%s", ffhip_last_error());
        if (pb.b && !pb.cached) ffhip_batch_destroy(pb.b);
        pb.b = NULL;
    }
    t_phase[3] += now_s() - t0;
    return pb;
}

static void collect_batch(const struct ffhip_model *mdl, pending_batch *pb) {
    ffhip_batch *b = pb->b;
    item **its = pb->its;
    const int n = pb->n, cached = pb->cached, *idx = pb->idx;
    const ffhip_prep *prep = pb->prep;
    double t0 = now_s();
    if (NULL == b || 0 != ffhip_batch_finish(b)) {
        if (b) { warnx("%s", ffhip_last_error()); if (!cached) ffhip_batch_destroy(b); }
        free(pb->idx); free(pb->its);
        pb->b = NULL;
        return;
    }
    t_phase[3] += now_s() - t0; t0 = now_s();
    const size_t nblock_cap = ffhip_batch_nblock(b), nstate = 2 * ffhip_model_nbase(mdl);
    (void)nblock_cap; (void)nstate;
#ifdef BUILD_RUNNIE
    {   /* runnie.c:262-313: path from decode_crf_runlength, then one line per emitted base with the discrete-Weibull
         * shape and scale of the block that emitted it and the dwell in blocks */
        const size_t nbase = ffhip_model_nbase(mdl), P = ffhip_model_nparam(mdl);
        int *path = malloc((nblock_cap + 1) * sizeof(int));
        float *qp = malloc((nblock_cap + 1) * sizeof(float)), *mat = malloc(nblock_cap * P * sizeof(float));
        for (int i = 0; path && qp && mat && i < n; i++) {
            const size_t nblock = ffhip_batch_read_nblock(b, i);
            if (0 != ffhip_batch_get_path(b, i, path, qp)) continue;
            if (0 != (args.viterbi_only ? ffhip_batch_get_transitions(b, i, mat) : ffhip_batch_get_posterior(b, i, mat))) continue;
            size_t cap = 64 * (nblock + 1), len = 0;
            char *text = malloc(cap);
            if (NULL == text) continue;
            text[0] = 0;
            int dwell = 1, last_blk = -1;
            for (size_t blk = 0; blk <= nblock; blk++) {
                const int emit = (blk < nblock) ? (path[blk] < (int)nbase) : 1;      /* the final pass flushes the last run */
                if (!emit) { dwell += 1; continue; }
                if (last_blk >= 0) {
                    const int base = path[last_blk];
                    /* %f of a huge value prints ~47 characters: grow the buffer instead of trusting the 64-per-block estimate
                     * (snprintf returns the would-be length, so an unchecked `len +=` walks past `cap` after one truncation) */
                    for (;;) {
                        const int ret = snprintf(text + len, cap - len, "%c\t%f\t%f\t%d\n", basechar(base), mat[(size_t)last_blk * P + base],
                                                 mat[(size_t)last_blk * P + nbase + base], dwell);
                        if (ret >= 0 && (size_t)ret < cap - len) { len += (size_t)ret; break; }
                        char *bigger = (ret < 0) ? NULL : realloc(text, 2 * cap + (size_t)ret);
                        if (NULL == bigger) { text[len] = 0; blk = nblock; break; }      /* out of memory: keep what fits */
                        text = bigger;
                        cap = 2 * cap + (size_t)ret;
                    }
                }
                last_blk = (int)blk;
                dwell = 1;
            }
            its[i]->rle_text = text;
            its[i]->res.score = ffhip_batch_score(b, i);
            its[i]->res.nblock = nblock;
        }
        free(path); free(qp); free(mat);
        t_phase[4] += now_s() - t0; t0 = now_s();
        if (!cached) ffhip_batch_destroy(b);
        t_phase[2] += now_s() - t0;
        free(pb->idx); free(pb->its);
        pb->b = NULL;
        return;
    }
#endif
    for (int i = 0; i < n; i++) {
        struct _raw_basecall_info *r = &its[i]->res;
        const size_t nblock = ffhip_batch_read_nblock(b, i);
        n_called_reads++;
        n_called_samples += r->rt.end - r->rt.start;
        n_raw_samples += r->rt.n;
        size_t blen = 0;
        const char *bases = ffhip_batch_basecall(b, i, &blen);
        r->basecall = strdup(bases);
        r->quality = strdup(ffhip_batch_quality(b, i));
        r->basecall_length = blen;
        r->score = ffhip_batch_score(b, i);
        r->nblock = nblock;
        r->pos = calloc(nblock + 1, sizeof(int));
        if (args.reverse) {                                    /* flappie.c:294-297 */
            reverse_char_array(r->basecall, blen);
            reverse_char_array(r->quality, blen);
        }
        if (args.trace) {
            int32_t *tmp = malloc((nblock + 1) * nstate * sizeof(int32_t));
            r->trace = make_flappie_imatrix(nstate, nblock + 1);
            if (tmp && r->trace && 0 == ffhip_batch_get_trace(b, i, tmp)) {
                for (size_t c = 0; c <= nblock; c++) memcpy(r->trace->data.f + c * r->trace->stride, tmp + c * nstate, nstate * sizeof(int32_t));
            } else {
                r->trace = free_flappie_imatrix(r->trace);
            }
            free(tmp);
            /* write_summary stores the signal the network saw (fast5_interface.c:332-335) */
            if (0 != ffhip_prep_get_signal(prep, idx[i], r->rt.raw + r->rt.start)) warnx("%s", ffhip_last_error());
        }
    }
    t_phase[4] += now_s() - t0; t0 = now_s();
    if (!cached) ffhip_batch_destroy(b);
    t_phase[2] += now_s() - t0;
    free(pb->idx); free(pb->its);
    pb->b = NULL;
}

/* A chunk of reads on its way through the GPU: prepared in one device pass (flappie.c:248-262: trim/segment, then med-MAD or
 * --delta), cut into batches of similar trimmed length, written in input order (flappie.c:371-384).  The batch pipeline does
 * NOT drain between chunks: the first batch of chunk k+1 is submitted before the last batch of chunk k is collected, and chunk
 * k is written while chunk k+1 runs -- the signal preparation of k+1 and the output of k overlap GPU work. */
typedef struct chunk_ctx {
    item *items;
    int n, buf;                  /* buf: which of the reader's buffers holds the items (released when the chunk is written) */
    ffhip_prep *prep;
    raw_table *rts;
    item **group;
    int m2, submitted, collected, all_submitted, live;
} chunk_ctx;

static int by_length_desc(const void *x, const void *y) {
    const item *a = *(item *const *)x, *b = *(item *const *)y;
    const size_t la = a->res.rt.end - a->res.rt.start, lb = b->res.rt.end - b->res.rt.start;
    return (la < lb) - (la > lb);
}

/* The chunk's device pass in two halves (round 6, third session): begin enqueues it (wait = 0: ffhip_prep_begin returns at once), finish waits for it, takes the
 * ranges and sorts the reads.  pipe_chunk begins chunk k + 1 before it submits chunk k's batches when that chunk is a large one: its preparation -- one workgroup a
 * read, as long as the longest read's selection passes last, and held behind the engine's last layer launch -- then runs beside chunk k's convolutions instead of in
 * front of chunk k + 1's (profiles/r06_pack_trace.txt). */
static void chunk_begin(struct ffhip_engine *eng, chunk_ctx *c, item *items, int n, int buf, int wait) {
    memset(c, 0, sizeof(*c));
    c->items = items; c->n = n; c->buf = buf; c->live = 1;
    c->group = calloc(n > 0 ? n : 1, sizeof(item *));
    c->rts = calloc(n > 0 ? n : 1, sizeof(raw_table));
    int m = 0;
    for (int i = 0; i < n; i++) {
        items[i].prepared = -1;
        if (NULL != items[i].res.rt.raw) { c->rts[m] = items[i].res.rt; items[i].prepared = m++; }
    }
    double tp0 = now_s();
    c->prep = (m > 0) ? (wait ? ffhip_prep_create : ffhip_prep_begin)(eng, c->rts, m, args.trim_start, args.trim_end, args.varseg_chunk, args.varseg_thresh,
                                                                         (args.delta == 0.0f) ? FFHIP_PREP_MEDMAD : FFHIP_PREP_DELTA, args.delta) : NULL;
    t_phase[1] += now_s() - tp0;
    if (m > 0 && NULL == c->prep) warnx("%s", ffhip_last_error());
}

static void chunk_begin_finish(chunk_ctx *c) {
    item *items = c->items;
    const int n = c->n;
    double tp0 = now_s();
    if (NULL != c->prep && 0 != ffhip_prep_finish(c->prep)) { warnx("%s", ffhip_last_error()); ffhip_prep_destroy(c->prep); c->prep = NULL; }
    t_phase[1] += now_s() - tp0;
    for (int i = 0; i < n; i++) {
        if (items[i].prepared < 0) continue;
        size_t st = 0, en = 0;
        if (NULL == c->prep || 0 != ffhip_prep_range(c->prep, items[i].prepared, &st, &en) || st >= en) { items[i].prepared = -1; continue; }
        items[i].res.rt.start = st;
        items[i].res.rt.end = en;
    }
    /* ragged batches: reads sorted by trimmed length, a batch takes up to --batch consecutive ones as long as the
     * shortest is at least 3/4 of the longest (a read tile of 16 costs what its longest read costs) */
    for (int i = 0; i < n; i++) if (items[i].prepared >= 0) c->group[c->m2++] = &items[i];
    qsort(c->group, c->m2, sizeof(item *), by_length_desc);
}

/* --trace: the filters of the summary file (shuffle + deflate, 10-15 ms per 100 000-sample read) are the one expensive step of
 * the output side and libhdf5 runs them under its caller's lock; here they run in worker threads, per read, outside libhdf5
 * (summary_pack_create), and the writer below only hands finished chunks over */
typedef struct { chunk_ctx *c; summary_pack **packs; int k, nthread; } pack_job;
static void *pack_worker(void *arg) {
    pack_job *j = arg;
    for (int i = j->k; i < j->c->n; i += j->nthread) {
        item *it = &j->c->items[i];
        if (NULL != it->res.basecall) j->packs[i] = summary_pack_create(it->res, args.compression_chunk_size, args.compression_level);
    }
    return NULL;
}
static summary_pack **pack_chunk(chunk_ctx *c) {
    summary_pack **packs = calloc(c->n > 0 ? c->n : 1, sizeof(*packs));
    if (NULL == packs) return NULL;
    const char *e = getenv("FLAPPIE_TRACE_THREADS");
    int nthread = e ? atoi(e) : 16;
    if (nthread < 1) nthread = 1;
    if (nthread > 64) nthread = 64;
    if (nthread > c->n) nthread = c->n > 0 ? c->n : 1;
    pthread_t th[64];
    pack_job jobs[64];
    char started[64];
    for (int k = 0; k < nthread; k++) {                       /* the last share runs here; so does any share whose thread did not start */
        jobs[k] = (pack_job){ c, packs, k, nthread };
        started[k] = (k + 1 < nthread) && 0 == pthread_create(&th[k], NULL, pack_worker, &jobs[k]);
        if (!started[k]) pack_worker(&jobs[k]);
    }
    for (int k = 0; k < nthread; k++) if (started[k]) pthread_join(th[k], NULL);
    return packs;
}

/* output of a finished chunk, in input order; releases what the chunk owns */
static void chunk_finish(chunk_ctx *c, hid_t hdf5out) {
    const double to0 = now_s();
    summary_pack **packs = (hdf5out >= 0) ? pack_chunk(c) : NULL;
    for (int i = 0; i < c->n; i++) {
        item *it = &c->items[i];
#ifdef BUILD_RUNNIE
        if (NULL != it->rle_text) {
            fprintf(args.output, "# %s\n%s", it->res.rt.uuid ? it->res.rt.uuid : "", it->rle_text);      /* runnie.c:280 */
            free(it->rle_text);
            it->rle_text = NULL;
        } else {
            warnx("No basecall returned for %s", it->filename);
        }
        free_raw_basecall_info(&it->res);
        free(it->filename);
        continue;
#endif
        if (NULL == it->res.basecall) {
            warnx("No basecall returned for %s", it->filename);
        } else {
            char *fn = strdup(it->filename);
            const char *base = basename(fn);
            const char *uuid = it->res.rt.uuid ? it->res.rt.uuid : "";
            fprintf_format(args.outformat, args.output, uuid, base, args.uuid, args.prefix, it->res);
            if (hdf5out >= 0) {
                pthread_mutex_lock(&hdf5_lock);
                if (packs && packs[i]) summary_pack_write(hdf5out, args.uuid ? uuid : base, packs[i]);
                else write_summary(hdf5out, args.uuid ? uuid : base, it->res, args.compression_chunk_size, args.compression_level);
                pthread_mutex_unlock(&hdf5_lock);
                if (packs) { summary_pack_free(packs[i]); packs[i] = NULL; }
            }
            free(fn);
        }
        free_raw_basecall_info(&it->res);
        free(it->filename);
    }
    if (packs) { for (int i = 0; i < c->n; i++) summary_pack_free(packs[i]); free(packs); }
    t_phase[5] += now_s() - to0;
    free(c->rts);
    free(c->group);
    __atomic_store_n(&c->live, 0, __ATOMIC_RELEASE);
}

/* the pipeline's state: up to NINFLIGHT - 1 batches submitted and not collected (oldest first) while the next is submitted (by default
 * one: a batch runs while the next is set up); chunks finish (are written) strictly in order */
#define NCHUNKBUF 4
#define PACK_ROW_MAX ((size_t)1 << 18)      /* samples a row of a packed batch holds at most (262 144: workspace of a 512-row batch ~45 GB at 384 hidden units) */
static int rs_chunk_cap = 0;                 /* reads a chunk holds at most (= reads a packed batch must take) */
/* packed batches: models whose default path takes them (ffhip_model_packable), ordinary temperatures, the flip-flop caller (runnie's model has no packed form);
 * FLAPPIE_DEBUG=no_pack keeps the one-read-a-row batches */
static int pack_allowed(const struct ffhip_model *mdl) {
    static int v = -1;
    if (v < 0) v = (ffhip_model_packable(mdl) && args.temperature >= 0.2f && args.temperature <= 5.0f && !cli_dbg("no_pack")) ? 1 : 0;
    return v;
}
/* Reads are sorted by length inside a chunk and cut into batches there (a batch costs what its longest read costs): the more batches a chunk
 * holds, the narrower the spread of lengths inside one.  On 3500-5500-sample reads 8 instead of 4 takes 2-8 % off the GPU time of a run
 * (65 536 files: 4.07 -> 4.00 s at 384 hidden units, 2.51 -> 2.38 s at 256); 16 gains little more and delays the first and the last output. */
#define CHUNK_BATCHES 8
static struct {
    pending_batch fifo[NINFLIGHT - 1];
    int nfifo, slot;
    chunk_ctx ctx[NCHUNKBUF];
    long next_finish, nbegun;            /* chunk sequence numbers; chunk q lives in ctx[q % NCHUNKBUF] */
    void (*released)(int buf, void *arg);
    void *released_arg;
} pipe_state;

/* A WRITER THREAD takes the finished chunks, strictly in order: formatting, --trace compression and the HDF5 writes of chunk k run
 * while the main thread keeps submitting the batches of chunk k + 1 (with --trace on 100 000-sample reads the output side costs as
 * much time as the GPU side; done on the main thread it stood between two submissions).  chunk sequence numbers in
 * [writer.head, writer.tail) are ready to be written; FLAPPIE_DEBUG=no_writer_thread writes on the main thread. */
static struct { pthread_t th; pthread_mutex_t mu; pthread_cond_t cv; long head, tail; int started, stop; hid_t hdf5out; } writer =
    { 0, PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, 0, 0, 0, 0, -1 };

static void writer_do(long q) {
    chunk_ctx *c = &pipe_state.ctx[q % NCHUNKBUF];
    const int buf = c->buf;
    chunk_finish(c, writer.hdf5out);                  /* sets c->live = 0 */
    if (pipe_state.released) pipe_state.released(buf, pipe_state.released_arg);
}

static void *writer_main(void *arg) {
    (void)arg;
    pthread_mutex_lock(&writer.mu);
    for (;;) {
        while (writer.head == writer.tail && !writer.stop) pthread_cond_wait(&writer.cv, &writer.mu);
        if (writer.head == writer.tail) break;
        const long q = writer.head;
        pthread_mutex_unlock(&writer.mu);
        writer_do(q);
        pthread_mutex_lock(&writer.mu);
        writer.head++;
        pthread_cond_broadcast(&writer.cv);
    }
    pthread_mutex_unlock(&writer.mu);
    return NULL;
}

static void pipe_finish_ready(hid_t hdf5out) {
    while (pipe_state.next_finish < pipe_state.nbegun) {
        chunk_ctx *c = &pipe_state.ctx[pipe_state.next_finish % NCHUNKBUF];
        if (!(c->all_submitted && c->collected == c->submitted)) break;
        ffhip_prep_destroy(c->prep);                  /* (the engine's buffer pool belongs to this thread) */
        c->prep = NULL;
        writer.hdf5out = hdf5out;
        if (!writer.started && !cli_dbg("no_writer_thread")) writer.started = (0 == pthread_create(&writer.th, NULL, writer_main, NULL)) ? 1 : -1;
        if (writer.started == 1) {
            pthread_mutex_lock(&writer.mu);
            writer.tail = pipe_state.next_finish + 1;
            pthread_cond_broadcast(&writer.cv);
            pthread_mutex_unlock(&writer.mu);
        } else {
            writer_do(pipe_state.next_finish);
        }
        pipe_state.next_finish++;
    }
}

/* all chunks handed over so far are written (end of the run) */
static void writer_finish(void) {
    if (writer.started != 1) return;
    pthread_mutex_lock(&writer.mu);
    writer.stop = 1;
    pthread_cond_broadcast(&writer.cv);
    pthread_mutex_unlock(&writer.mu);
    pthread_join(writer.th, NULL);
}

/* collect the oldest batch in flight */
static void pipe_collect_oldest(const struct ffhip_model *mdl, hid_t hdf5out) {
    if (0 == pipe_state.nfifo) return;
    pending_batch old = pipe_state.fifo[0];
    for (int k = 1; k < pipe_state.nfifo; k++) pipe_state.fifo[k - 1] = pipe_state.fifo[k];
    pipe_state.nfifo--;
    chunk_ctx *owner = old.owner;
    collect_batch(mdl, &old);
    owner->collected++;
    pipe_finish_ready(hdf5out);
}
static void pipe_collect_all(const struct ffhip_model *mdl, hid_t hdf5out) {
    while (pipe_state.nfifo) pipe_collect_oldest(mdl, hdf5out);
}

/* reads sorted by trimmed length, longest first, in one-read-a-row batches: a batch takes up to --batch consecutive ones as long as the shortest is at least 3/4 of the
 * longest (a read tile of 16 costs what its longest read costs) */
static void run_groups(struct ffhip_engine *eng, const struct ffhip_model *mdl, struct chunk_ctx *c, item **group, int m2, hid_t hdf5out, int depth);
static int pack_failed = 0;                  /* a packed batch object could not be created (memory): the rest of the run goes one read a row */

/* the context of the next chunk to begin (chunk sequence number nbegun): its slot's previous chunk may still be with the writer */
static chunk_ctx *chunk_slot(void) {
    chunk_ctx *c = &pipe_state.ctx[pipe_state.nbegun % NCHUNKBUF];
    if (writer.started == 1) {
        pthread_mutex_lock(&writer.mu);
        while (__atomic_load_n(&c->live, __ATOMIC_ACQUIRE)) pthread_cond_wait(&writer.cv, &writer.mu);
        pthread_mutex_unlock(&writer.mu);
    }
    if (c->live) errx(EXIT_FAILURE, "internal error: chunk slot still in use");
    return c;
}
static chunk_ctx *chunk_ahead = NULL;        /* the chunk whose device pass was begun while its predecessor was handled (its items are the next pipe_chunk's) */

/* look_ahead (may be NULL): asked AFTER this chunk's own device pass has been waited for -- i.e. when the batch in flight has just ended -- whether the chunk that follows is in
 * the reader's buffers already and is a large one; its device pass is then begun here, in front of this chunk's batches, and runs beside their convolutions */
typedef int (*look_ahead_fn)(item **items, int *n, int *buf);
static void pipe_chunk(struct ffhip_engine *eng, const struct ffhip_model *mdl, item *items, int n, int buf, hid_t hdf5out, look_ahead_fn look_ahead) {
    chunk_ctx *c = chunk_ahead;
    if (NULL != c) {
        if (c->items != items || c->n != n) errx(EXIT_FAILURE, "internal error: the chunk prepared ahead is not the one that follows");
        chunk_ahead = NULL;
    } else {
        c = chunk_slot();
        chunk_begin(eng, c, items, n, buf, 1);     /* the device pass runs beside the batch still in flight */
        pipe_state.nbegun++;
    }
    chunk_begin_finish(c);
    item *next_items = NULL;
    int next_n = 0, next_buf = 0;
    if (NULL != look_ahead && look_ahead(&next_items, &next_n, &next_buf)) {
        chunk_ahead = chunk_slot();
        chunk_begin(eng, chunk_ahead, next_items, next_n, next_buf, 0);
        pipe_state.nbegun++;
    }
    static int depth = 0;
    if (0 == depth) { const char *e = getenv("FLAPPIE_INFLIGHT"); depth = (e && atoi(e) == 3) ? NINFLIGHT - 1 : 1; }
    /* ---- a chunk of MIXED lengths goes in packed batches: what the one-read-a-row grouping below would pay for (rows x the group's longest read, group by
     * group) against what the reads hold; below 0.85 the chunk's reads are placed several to a row (ffhip_pack_plan: longest first, each into the emptiest row) in as few batches of --batch rows
     * as hold them, rows just long enough -- a batch then costs what its samples cost (nanopore-like mix: 0.07 -> 0.9+, tools/length_mix.py) */
    int packed_chunk = 0;
    if (c->m2 > 0 && pack_allowed(mdl) && !pack_failed) {
        unsigned long long real = 0, paid = 0;
        for (int i = 0; i < c->m2; ) {
            const size_t longest = c->group[i]->res.rt.end - c->group[i]->res.rt.start;
            int g = 1;
            real += longest;
            while (i + g < c->m2 && g < args.batch && 4 * (c->group[i + g]->res.rt.end - c->group[i + g]->res.rt.start) >= 3 * longest) { real += c->group[i + g]->res.rt.end - c->group[i + g]->res.rt.start; g++; }
            paid += (unsigned long long)longest * (unsigned long long)(4 * g < args.batch ? 16 * ((g + 15) / 16) : args.batch);
            i += g;
        }
        packed_chunk = (double)real < 0.85 * (double)paid;
    }
    if (packed_chunk) {
        size_t *ns = malloc(c->m2 * sizeof(size_t));
        int *slot_of = malloc(c->m2 * sizeof(int)), *off_of = malloc(c->m2 * sizeof(int)), *sl2 = malloc(c->m2 * sizeof(int)), *of2 = malloc(c->m2 * sizeof(int));
        item **sel = malloc(c->m2 * sizeof(item *)), **rest = malloc(c->m2 * sizeof(item *));
        int nleft = c->m2;
        memcpy(rest, c->group, c->m2 * sizeof(item *));
        /* The SHAPE of this chunk's batch objects is fixed by the chunk as a whole -- rows long enough for its longest read and for everything in ONE batch of --batch rows
         * (5 % slack for the gaps and the fit; at most PACK_ROW_MAX samples unless a read is longer), capacities in pack_row_cap's steps, as many rows (<= --batch) as the
         * device's memory takes at that length (ffhip_pack_rows_for: 1024 rows of 237 568 samples at 256 hidden units are 160 GB) -- and every batch of the chunk is planned
         * into it: what one batch does not hold goes into the next, whose reads are shorter (longest first) and fit the same object. */
        unsigned long long total_all = 0;
        for (int i = 0; i < nleft; i++) total_all += rest[i]->res.rt.end - rest[i]->res.rt.start;
        const size_t longest_all = rest[0]->res.rt.end - rest[0]->res.rt.start;
        size_t cap_all = (size_t)((double)total_all / (double)args.batch * 1.05) + 64 * ffhip_model_pack_gap(mdl);
        if (cap_all < longest_all + 64) cap_all = longest_all + 64;
        if (cap_all > PACK_ROW_MAX && longest_all + 64 <= PACK_ROW_MAX) cap_all = PACK_ROW_MAX;
        size_t cap_obj = pack_row_cap(cap_all);
        /* Two objects of --batch rows do not fit (H = 256: 1024 rows of 237 568 samples are 160 GB each) but ONE does: the full launch is worth more than setting a batch up
         * beside the one that runs (k_lstm_pack at 1024 rows 200+ Msamples/s, the forms for 512 rows 155; a packed batch's set-up is milliseconds) -- this chunk's batches
         * go through one object, each collected before the next is set up */
        int single = 0;
        int rows_mem = ffhip_pack_rows_for(mdl, args.batch, cap_obj, 2);
        if (rows_mem < args.batch) {
            /* (... or at least half again as many rows as two objects would have: the r941_5mC shape -- stride 2, 2.5 x the blocks a sample -- on 200 000-sample reads takes
             * 272 rows twice or 544 once; rows are what fills the chip) */
            const int rows_one = ffhip_pack_rows_for(mdl, args.batch, cap_obj, 1);
            if (rows_one >= args.batch || 2 * rows_one >= 3 * rows_mem) { single = 1; rows_mem = rows_one; }
        }
        if (rows_mem < args.batch && rows_mem >= args.batch / 2) rows_mem = args.batch / 2;      /* (half a launch's rows keep a form of the layer kernel that fills the chip; 7/8 of one does not) */
        /* ... unless a full-size object with rows long enough is there already: its shape stands (a shorter chunk could take more rows of shorter rows -- and pay seconds of
         * hipMalloc for them, chunk after chunk) */
        for (int k = 0; k < NINFLIGHT; k++)
            if (pack_cache[k].b && pack_cache[k].full && pack_cache[k].cap >= cap_obj) { cap_obj = pack_cache[k].cap; rows_mem = pack_cache[k].rows; single = pack_cache[k].single; break; }
        while (nleft > 0 && ns && slot_of && off_of && sl2 && of2 && sel && rest) {
            unsigned long long total = 0;
            for (int i = 0; i < nleft; i++) { ns[i] = rest[i]->res.rt.end - rest[i]->res.rt.start; total += ns[i]; }
            /* this batch: rows just long enough for what is left (a launch runs as long as its longest row), and no more rows than the samples left fill */
            size_t cap = (size_t)((double)total / (double)rows_mem * 1.05) + 64 * ffhip_model_pack_gap(mdl);
            if (cap < ns[0] + 64) cap = ns[0] + 64;
            cap = (cap + 1023) & ~(size_t)1023;
            if (cap > cap_obj) cap = cap_obj;
            int rows = (int)(((double)total * 1.10 / (double)cap) / 16.0 + 2.0) * 16;
            if (rows > rows_mem) rows = rows_mem;
            int placed = 0;
            for (int tries = 0; tries < 6; tries++) {
                placed = ffhip_pack_plan(mdl, rows, cap, nleft, ns, slot_of, off_of);
                if (placed == nleft) break;
                if (rows < rows_mem) { rows = (rows + 32 < rows_mem) ? rows + 32 : rows_mem; continue; }      /* the fit left reads over: more rows, ... */
                if (cap >= cap_obj) break;                                                                    /* ... or longer ones, as far as the object goes */
                cap = ((size_t)((double)cap * 1.06) + 1023) & ~(size_t)1023;
                if (cap > cap_obj) cap = cap_obj;
            }
            if (placed <= 0) { warnx("packed batch: no read fits a row of %zu samples", cap); break; }
            int nsel = 0, nrest = 0;
            for (int i = 0; i < nleft; i++) {
                if (slot_of[i] >= 0) { sel[nsel] = rest[i]; sl2[nsel] = slot_of[i]; of2[nsel] = off_of[i]; nsel++; }
                else rest[nrest++] = rest[i];
            }
            if (single) {                                  /* one object: nothing in flight while it is set up, and no second object alive */
                pipe_collect_all(mdl, hdf5out);
                pipe_state.slot = 0;
                for (int k = 1; k < NINFLIGHT; k++) if (pack_cache[k].b) { ffhip_batch_destroy(pack_cache[k].b); pack_cache[k].b = NULL; pack_cache[k].cap = 0; }
            }
            pending_batch cur = submit_packed(eng, mdl, c->prep, sel, nsel, sl2, of2, rows, rows_mem, cap, cap_obj, rs_chunk_cap, pipe_state.slot, single);
            if (NULL == cur.b && NULL == pack_cache[pipe_state.slot].b) {
                /* the object could not be created (device memory): no read is dropped for that -- this batch's reads and the chunk's remaining ones go one read a row,
                 * and so does the rest of the run */
                warnx("packed batches are off for the rest of this run (no memory for %d rows of %zu samples)", rows_mem, cap_obj);
                pack_failed = 1;
                free(cur.idx); free(cur.its);
                pipe_collect_all(mdl, hdf5out);                /* (nothing in flight any more: the packed objects that were made give their memory back) */
                for (int k = 0; k < NINFLIGHT; k++) if (pack_cache[k].b) { ffhip_batch_destroy(pack_cache[k].b); pack_cache[k].b = NULL; pack_cache[k].cap = 0; }
                memcpy(sel + nsel, rest, nrest * sizeof(item *));
                qsort(sel, nsel + nrest, sizeof(item *), by_length_desc);
                run_groups(eng, mdl, c, sel, nsel + nrest, hdf5out, depth);
                break;
            }
            cur.owner = c;
            c->submitted++;
            while (pipe_state.nfifo >= depth) pipe_collect_oldest(mdl, hdf5out);
            pipe_state.fifo[pipe_state.nfifo++] = cur;
            pipe_state.slot = single ? 0 : (pipe_state.slot + 1) % (depth + 1);
            nleft = nrest;
        }
        free(ns); free(slot_of); free(off_of); free(sl2); free(of2); free(sel); free(rest);
    } else run_groups(eng, mdl, c, c->group, c->m2, hdf5out, depth);
    c->all_submitted = 1;
    /* A chunk without a single batch (every read failed) collects nothing on its own account: the batch still in flight belongs
     * to an OLDER chunk, whose reader buffer is only released when it is written -- two such chunks in a row and the reader
     * thread would wait for that buffer while this thread waits for the reader.  Collect it now. */
    if (0 == c->m2) pipe_collect_all(mdl, hdf5out);
    pipe_finish_ready(hdf5out);
}

static void run_groups(struct ffhip_engine *eng, const struct ffhip_model *mdl, struct chunk_ctx *c, item **group, int m2, hid_t hdf5out, int depth) {
    for (int i = 0; i < m2; ) {
        const size_t longest = group[i]->res.rt.end - group[i]->res.rt.start;
        int g = 1;
        while (i + g < m2 && g < args.batch && 4 * (group[i + g]->res.rt.end - group[i + g]->res.rt.start) >= 3 * longest) g++;
        pending_batch cur = submit_batch(eng, mdl, c->prep, group + i, g, pipe_state.slot);
        cur.owner = c;
        c->submitted++;
        /* default: one batch runs while the next is set up.  FLAPPIE_INFLIGHT=3 keeps two on the GPU while the third is set up: no
         * measurable gain on 4000-sample reads (80-92 against 84-87 Msamples/s, run-to-run noise), and a third batch object costs
         * long reads another 10+ GB of workspace and ~1 s of start-up */
        while (pipe_state.nfifo >= depth) pipe_collect_oldest(mdl, hdf5out);           /* may complete and write an earlier chunk */
        pipe_state.fifo[pipe_state.nfifo++] = cur;
        pipe_state.slot = (pipe_state.slot + 1) % (depth + 1);       /* depth in flight + the one being set up: no more batch objects than that */
        i += g;
    }
}

/* Without the reader thread (FLAPPIE_DEBUG=no_reader_thread) this thread fills the reader buffers itself: buffer k may only be
 * overwritten once the chunk that last used it has been WRITTEN -- by the writer thread, possibly still at work. */
static void pipe_wait_slot_written(void) {
    chunk_ctx *c = &pipe_state.ctx[pipe_state.nbegun % NCHUNKBUF];
    if (writer.started != 1) return;               /* no writer thread: chunks are written by this thread, in pipe_finish_ready */
    pthread_mutex_lock(&writer.mu);
    while (__atomic_load_n(&c->live, __ATOMIC_ACQUIRE)) pthread_cond_wait(&writer.cv, &writer.mu);
    pthread_mutex_unlock(&writer.mu);
}

static void pipe_drain(const struct ffhip_model *mdl, hid_t hdf5out) {
    pipe_collect_all(mdl, hdf5out);
    pipe_finish_ready(hdf5out);
    writer_finish();
}

/* ---- input side: the list of files (flappie.c:336-358), read one chunk ahead of the GPU by a reader thread ---- */
typedef struct { char **path; size_t n, cap; } file_list;

static int by_path(const void *a, const void *b) { return strcmp(*(char *const *)a, *(char *const *)b); }

/* --shard-by-size: the shard of every file of the sorted list, balanced by bytes (a single-read fast5 file's size follows its sample
 * count): files in order of decreasing size, ties by position, each to the shard that holds the fewest bytes so far (lowest index on a
 * tie) -- flappie_amd/shard.py::partition_reads, SURVEY.md section 8e.  Every process of a sharded run computes the same table. */
typedef struct { off_t size; size_t idx; } sized_file;
static int by_size_desc(const void *a, const void *b) {
    const sized_file *x = a, *y = b;
    if (x->size != y->size) return x->size < y->size ? 1 : -1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
}
static int *shards_by_size(char **path, size_t n, int nshard) {
    sized_file *sf = malloc((n ? n : 1) * sizeof(*sf));
    int *shard = malloc((n ? n : 1) * sizeof(int));
    unsigned long long *load = calloc((size_t)nshard, sizeof(*load));
    if (NULL == sf || NULL == shard || NULL == load) errx(EXIT_FAILURE, "out of memory");
    for (size_t f = 0; f < n; f++) {
        struct stat st;
        sf[f].size = (0 == stat(path[f], &st)) ? st.st_size : 0;
        sf[f].idx = f;
    }
    qsort(sf, n, sizeof(*sf), by_size_desc);
    for (size_t k = 0; k < n; k++) {
        int best = 0;
        for (int g = 1; g < nshard; g++) if (load[g] < load[best]) best = g;
        shard[sf[k].idx] = best;
        load[best] += (unsigned long long)sf[k].size;
    }
    free(sf); free(load);
    return shard;
}

static void list_files(file_list *fl) {
    const int reads_limit = (args.nshard >= 1) ? 0 : args.limit;      /* a shard is cut from the whole list, then limited */
    for (int fn = 0; args.files && args.files[fn]; fn++) {
        if (reads_limit > 0 && (int)fl->n >= reads_limit) break;
        glob_t globbuf;
        /* a directory means every .fast5 file inside it (flappie.c:341-353) */
        const size_t rootlen = strlen(args.files[fn]);
        char *globpath = calloc(rootlen + 9, sizeof(char));
        memcpy(globpath, args.files[fn], rootlen);
        DIR *dirp = opendir(args.files[fn]);
        if (NULL != dirp) { memcpy(globpath + rootlen, "/*.fast5", 8); closedir(dirp); }
        const int globret = glob(globpath, GLOB_NOSORT, NULL, &globbuf);
        free(globpath);
        if (0 != globret) {
            if (GLOB_NOMATCH == globret) warnx("File or directory \"%s\" does not exist or no fast5 files found.", args.files[fn]);
            globfree(&globbuf);
            continue;
        }
        for (size_t f2 = 0; f2 < globbuf.gl_pathc; f2++) {
            if (reads_limit > 0 && (int)fl->n >= reads_limit) break;
            if (fl->n == fl->cap) { fl->cap = fl->cap ? 2 * fl->cap : 1024; fl->path = realloc(fl->path, fl->cap * sizeof(char *)); }
            fl->path[fl->n++] = strdup(globbuf.gl_pathv[f2]);
        }
        globfree(&globbuf);
    }
    if (args.nshard >= 1) {
        /* every process of a sharded run must see the same order: sort, keep every n-th file starting at g */
        qsort(fl->path, fl->n, sizeof(char *), by_path);
        int *of = args.shard_by_size ? shards_by_size(fl->path, fl->n, args.nshard) : NULL;
        size_t kept = 0;
        for (size_t f = 0; f < fl->n; f++) {
            const int g = of ? of[f] : (int)(f % (size_t)args.nshard);
            if (g == args.shard && (args.limit <= 0 || (int)kept < args.limit)) fl->path[kept++] = fl->path[f];
            else free(fl->path[f]);
        }
        free(of);
        fl->n = kept;
    }
}

typedef struct {
    const file_list *fl;
    int chunk_cap;
    item *items[NCHUNKBUF];              /* one more than the chunks that can be unfinished at a time (one per batch in flight) */
    int nitem[NCHUNKBUF];
    sem_t filled[NCHUNKBUF], empty[NCHUNKBUF];
} reader_state;

/* ---- reader PROCESSES.  Opening and reading a single-read fast5 costs ~0.1-0.15 ms of CPU in libhdf5 (~6800 files/s on one
 * core; since round 6 read_raw walks the files it knows itself, host/fast5_raw.c: ~10 us, and libhdf5 is the fall-back), this libhdf5 is not thread-safe, and one MI355X consumes 15-20 000 reads of 4000 samples per second: one reader
 * thread caps the binary at less than half of what the kernels deliver.  So the files are read by `--readers` child
 * processes, forked BEFORE the HIP runtime starts (a fork of a process with a live GPU context is not supported): child k
 * reads files k, k + R, k + 2R ... in order (read_raw, fast5_interface.c:231-318, with the pA scaling of flappie.c:248) and
 * streams {nsample, uuid, samples} records down its own pipe; the parent's reader thread takes file f from pipe f % R, so
 * the input order is kept.  Children run ahead by what a pipe holds (1 MiB: ~50 reads each). */
typedef struct { pid_t pid; int fd; int dead; } reader_proc;
static reader_proc *rprocs = NULL;
static int nrproc = 0;

static int write_all(int fd, const void *buf, size_t n) {
    const char *p = buf;
    while (n > 0) {
        const ssize_t w = write(fd, p, n);
        if (w < 0) { if (EINTR == errno) continue; return -1; }
        p += w; n -= (size_t)w;
    }
    return 0;
}
static int read_all(int fd, void *buf, size_t n) {
    char *p = buf;
    while (n > 0) {
        const ssize_t r = read(fd, p, n);
        if (r < 0) { if (EINTR == errno) continue; return -1; }
        if (0 == r) return -1;
        p += r; n -= (size_t)r;
    }
    return 0;
}

static void reader_child(const file_list *fl, int k, int R, int fd) {
    signal(SIGPIPE, SIG_IGN);            /* a parent that went away is a failed write, not a signal (the PARENT keeps the default: `flappie ... | head` ends) */
    long kill_k = -1, kill_f = -1;       /* tests: FLAPPIE_DEBUG=kill_reader=k:f -- child k dies (SIGKILL) when it reaches file index f */
    const char *kill_env = cli_dbg("kill_reader");
    if (kill_env && 2 != sscanf(kill_env, "%ld:%ld", &kill_k, &kill_f)) kill_k = -1;
    for (size_t f = (size_t)k; f < fl->n; f += (size_t)R) {
        if (kill_k == k && (long)f >= kill_f) raise(SIGKILL);
        raw_table rt = read_raw(fl->path[f], true);
        uint64_t hdr[2] = { (NULL != rt.raw) ? (uint64_t)rt.n : 0, (NULL != rt.uuid && NULL != rt.raw) ? (uint64_t)strlen(rt.uuid) : 0 };
        if (0 != write_all(fd, hdr, sizeof(hdr)) || (hdr[1] && 0 != write_all(fd, rt.uuid, hdr[1])) ||
            (hdr[0] && 0 != write_all(fd, rt.raw, hdr[0] * sizeof(float)))) _exit(1);      /* the parent went away */
        free(rt.raw);
        free(rt.uuid);
    }
    _exit(0);
}

static void start_reader_procs(const file_list *fl, int R) {
    if (R <= 0 || 0 == fl->n) return;
    if ((size_t)R > fl->n) R = (int)fl->n;
    rprocs = calloc((size_t)R, sizeof(reader_proc));
    fflush(NULL);
    for (int k = 0; k < R; k++) {
        int pfd[2];
        if (0 != pipe(pfd)) errx(EXIT_FAILURE, "could not create a pipe for reader %d", k);
        const pid_t pid = fork();
        if (pid < 0) errx(EXIT_FAILURE, "could not fork reader %d", k);
        if (0 == pid) {
            close(pfd[0]);
            for (int j = 0; j < k; j++) close(rprocs[j].fd);
            reader_child(fl, k, R, pfd[1]);
        }
        close(pfd[1]);
#ifdef F_SETPIPE_SZ
        (void)fcntl(pfd[0], F_SETPIPE_SZ, 1 << 20);
#endif
        rprocs[k].pid = pid;
        rprocs[k].fd = pfd[0];
        nrproc = k + 1;
    }
}

/* A reader child that dies (libhdf5 crashing on a corrupt file, an OOM kill) takes its stripe with it: the file it died on is
 * reported unreadable, every LATER file of the stripe is read in this process instead (as with --readers 0), and the exit status
 * of the run says that a reader failed.  A stream cut in the middle of a record is out of sync and handled the same way. */
static raw_table reader_lost(size_t f, const char *path) {
    raw_table rt = { NULL, 0, 0, 0, NULL };
    reader_proc *rp = &rprocs[f % (size_t)nrproc];
    if (!rp->dead) {
        rp->dead = 1;
        reader_failures++;
        warnx("reader process %zu ended early at %s; the rest of its files are read in-process", f % (size_t)nrproc, path);
        return rt;                                                         /* this file: whatever killed the child stays unread */
    }
    pthread_mutex_lock(&hdf5_lock);
    rt = read_raw(path, true);
    pthread_mutex_unlock(&hdf5_lock);
    return rt;
}

static raw_table read_from_proc(size_t f, const char *path) {
    raw_table rt = { NULL, 0, 0, 0, NULL };
    if (rprocs[f % (size_t)nrproc].dead) return reader_lost(f, path);
    const int fd = rprocs[f % (size_t)nrproc].fd;
    uint64_t hdr[2];
    if (0 != read_all(fd, hdr, sizeof(hdr))) return reader_lost(f, path);
    char *uuid = hdr[1] ? calloc(hdr[1] + 1, 1) : NULL;
    if (hdr[1] && (NULL == uuid || 0 != read_all(fd, uuid, hdr[1]))) { free(uuid); return reader_lost(f, path); }
    if (0 == hdr[0]) { free(uuid); return rt; }                        /* the child could not read the file (it said why) */
    float *raw = malloc(hdr[0] * sizeof(float));
    if (NULL == raw || 0 != read_all(fd, raw, hdr[0] * sizeof(float))) { free(raw); free(uuid); return reader_lost(f, path); }
    rt = (raw_table){ uuid, hdr[0], 0, hdr[0], raw };
    return rt;
}

static void stop_reader_procs(void) {
    for (int k = 0; k < nrproc; k++) {
        close(rprocs[k].fd);
        int st = 0;
        if (waitpid(rprocs[k].pid, &st, 0) == rprocs[k].pid && !rprocs[k].dead) {
            /* exit status 1 = its pipe closed under it (this process stopped reading early: --limit, an error) -- not a failure of the child */
            if (WIFSIGNALED(st)) { warnx("reader process %d was killed by signal %d", k, WTERMSIG(st)); reader_failures++; }
            else if (WIFEXITED(st) && WEXITSTATUS(st) > 1) { warnx("reader process %d exited with status %d", k, WEXITSTATUS(st)); reader_failures++; }
        }
    }
    free(rprocs);
    rprocs = NULL;
    nrproc = 0;
}

/* Mixed lengths (packed batches): a read of L samples is L / stride dependent steps of every layer, whatever else the GPU holds meanwhile -- a chunk whose longest read
 * is L should hold about --batch rows of L samples of work, or its one batch runs as long as that read with most rows empty (tools/length_mix.py: 8192 files of a
 * log-normal mix, 33 M-sample chunks: 105 of 512 rows in use).  The chunk's sample budget therefore grows with the longest read seen while it fills, up to
 * PACK_WINDOW_MAX samples and PACK_WINDOW_READS x the usual read count (the reader's buffers are that large).  Uniform reads never get there. */
#define PACK_WINDOW_MAX ((size_t)224 << 20)
#define PACK_WINDOW_READS 4
static int g_pack_window = 0;                /* set once the model is known: pack_allowed() */
static void read_chunk(const file_list *fl, size_t first, int chunk_cap, item *items, int *nitem) {
    /* a chunk is up to chunk_cap reads (CHUNK_BATCHES batches of the usual 4-8 k-sample reads) -- or, with long reads, what holds about as
     * many SAMPLES but at least one batch: records leave when their chunk is done, and a 1024-read chunk of 100 000-sample reads
     * would be four seconds of GPU work with nothing written (and nothing for the writer thread to overlap) */
    size_t sample_budget = (size_t)chunk_cap * 8192, longest = 0;
    size_t nsamp = 0;
    int n = 0, read_cap = chunk_cap;
    for (size_t f = first; f < fl->n && n < read_cap && !(n >= args.batch && nsamp >= sample_budget); f++, n++) {
        item *it = &items[n];
        memset(it, 0, sizeof(*it));
        it->filename = fl->path[f];                                   /* ownership moves to the item */
        const double tr0 = now_s();
        if (nrproc > 0) {
            it->res.rt = read_from_proc(f, it->filename);
        } else {
            pthread_mutex_lock(&hdf5_lock);
            it->res.rt = read_raw(it->filename, true);                /* flappie.c:248 */
            pthread_mutex_unlock(&hdf5_lock);
        }
        t_phase[0] += now_s() - tr0;
        if (NULL != it->res.rt.raw) nsamp += it->res.rt.n;
        if (g_pack_window && NULL != it->res.rt.raw && it->res.rt.n > longest) {
            longest = it->res.rt.n;
            /* (... and a twentieth more: these are RAW samples, the rows hold trimmed ones, and rows that average a little MORE than the longest read let the planner --
             * every read into the emptiest row -- end with all rows alike; at exactly --batch x longest the trimming left the rows 2.7 % under a row that holds the
             * longest read alone, and that row is what the launch lasts: fill 0.973 -> 0.99, tools/dev/pack_log.sh) */
            size_t want = (size_t)args.batch * longest + (size_t)args.batch * longest / 20;
            if (want > PACK_WINDOW_MAX) want = PACK_WINDOW_MAX;
            if (want > sample_budget) { sample_budget = want; read_cap = PACK_WINDOW_READS * chunk_cap; }
        }
    }
    *nitem = n;
}

static void release_buffer(int buf, void *arg) { sem_post(&((reader_state *)arg)->empty[buf]); }

static void *reader_main(void *arg) {
    reader_state *rs = arg;
    size_t first = 0;
    for (int k = 0; first < rs->fl->n; k = (k + 1) % NCHUNKBUF) {
        sem_wait(&rs->empty[k]);
        /* the first chunk is one batch only, so that the GPU starts after --batch files instead of CHUNK_BATCHES times as many */
        read_chunk(rs->fl, first, first == 0 ? rs->chunk_cap / CHUNK_BATCHES : rs->chunk_cap, rs->items[k], &rs->nitem[k]);
        first += rs->nitem[k];
        sem_post(&rs->filled[k]);
    }
    return NULL;
}

/* pipe_chunk's look-ahead over the reader thread's buffers: buffer kn holds the next chunk if its `filled` can be taken now (the main loop then skips its own wait for it) */
static struct { reader_state *rs; int on, kn, more, taken; } la;
static int look_ahead_reader(item **items, int *n, int *buf) {
    if (!la.on || !la.more || la.taken) return 0;
    if (0 != sem_trywait(&la.rs->filled[la.kn])) {
        if (cli_dbg("pack_log")) fprintf(stderr, "look-ahead at %.3f s: the reader has not filled buffer %d yet\n", now_s() - t_program_start, la.kn);
        return 0;
    }
    la.taken = 1;
    if (cli_dbg("pack_log")) fprintf(stderr, "look-ahead at %.3f s: buffer %d is there (%d reads)\n", now_s() - t_program_start, la.kn, la.rs->nitem[la.kn]);
    unsigned long long raw = 0;
    for (int i = 0; i < la.rs->nitem[la.kn]; i++) if (NULL != la.rs->items[la.kn][i].res.rt.raw) raw += la.rs->items[la.kn][i].res.rt.n;
    unsigned long long least = 48ull << 20;      /* the usual chunks: their device pass is a millisecond beside the batch in flight as it is */
    const char *lm = cli_dbg("prep_ahead_min");  /* tests: FLAPPIE_DEBUG=prep_ahead_min=0 takes this path with chunks of any size */
    if (NULL != lm) least = strtoull(lm, NULL, 10);
    if (raw < least) return 0;
    *items = la.rs->items[la.kn]; *n = la.rs->nitem[la.kn]; *buf = la.kn;
    return 1;
}

/* FLAPPIE_DEBUG=segv_trace: the call stack of a crash on stderr (addresses for addr2line; development) */
static void segv_trace(int sig) {
    void *frames[48];
    const int n = backtrace(frames, 48);
    static const char msg[] = "flappie: fatal signal, call stack:\n";
    if (write(2, msg, sizeof(msg) - 1) < 0) _exit(128 + sig);
    backtrace_symbols_fd(frames, n, 2);
    _exit(128 + sig);
}

int main(int argc, char *argv[]) {
    t_program_start = now_s();
    argp_parse(&argp, argc, argv, 0, 0, NULL);
    if (cli_dbg("segv_trace")) { signal(SIGSEGV, segv_trace); signal(SIGABRT, segv_trace); signal(SIGBUS, segv_trace); }
    if (NULL == args.output) args.output = stdout;
    const double t_start = now_s();
    file_list fl = { NULL, 0, 0 };
    list_files(&fl);
    const double t_listed = now_s();
    if (!cli_dbg("no_numa_bind")) { const char *dev = getenv("FLAPPIE_HIP_DEVICE"); (void)bind_to_gpu_numa(dev ? atoi(dev) : 0); }
    if (cli_dbg("list_only")) {             /* the files this process would call, one per line -- no GPU touched (tests, tools/host_scaling.py) */
        for (size_t f = 0; f < fl.n; f++) printf("%s\n", fl.path[f]);
        return EXIT_SUCCESS;
    }
    start_reader_procs(&fl, cli_dbg("no_reader_thread") ? 0 : args.readers);      /* before the HIP runtime and libhdf5 are touched here */
    const struct ffhip_model *mdl = flappie_hip_model(args.model);
    if (NULL == mdl) { stop_reader_procs(); errx(EXIT_FAILURE, "model \"%s\" is not available (set FLAPPIE_MODEL_DIR)", flappie_model_string(args.model)); }
    struct ffhip_engine *eng = flappie_hip_engine();
    hid_t hdf5out = open_or_create_hdf5(args.trace);
    reader_state rs;
    memset(&rs, 0, sizeof(rs));
    rs.fl = &fl;
    /* reads per batch = what one layer launch takes (ffhip_rnn_split.hip): 1024 at H = 256 (the packed forms), 512 at H <= 384 (the dense
     * form), else 256 */
    if (0 == args.batch) args.batch = (int)ffhip_model_launch_reads(mdl);
    if (args.batch <= 0) args.batch = 256;              /* (the query answered 0: no model; never an empty batch) */
    rs.chunk_cap = CHUNK_BATCHES * args.batch;
    g_pack_window = pack_allowed(mdl);
    rs_chunk_cap = (g_pack_window ? PACK_WINDOW_READS : 1) * rs.chunk_cap;      /* reads a chunk may hold (a packed batch takes as many) */
    for (int k = 0; k < NCHUNKBUF; k++) {
        rs.items[k] = calloc(rs_chunk_cap, sizeof(item));
        sem_init(&rs.filled[k], 0, 0);
        sem_init(&rs.empty[k], 0, 1);
    }
    const int threaded = !cli_dbg("no_reader_thread");
    pthread_t reader;
    if (threaded && 0 != pthread_create(&reader, NULL, reader_main, &rs)) errx(EXIT_FAILURE, "could not start the reader thread");
    size_t done = 0;
    double t_wait = 0.0;
    if (threaded) { pipe_state.released = release_buffer; pipe_state.released_arg = &rs; }
    /* A LARGE chunk that follows (the window of a directory of long reads: 48 M raw samples and more, what ffhip_prep_create holds behind the engine's last layer launch) has its
     * device pass begun before this chunk's batches are submitted, if the reader has it ready by then (FLAPPIE_DEBUG=no_prep_ahead: never) */
    la.rs = &rs; la.on = threaded && !cli_dbg("no_prep_ahead"); la.taken = 0;
    for (int k = 0; done < fl.n; k = (k + 1) % NCHUNKBUF) {
        const double tw0 = now_s();
        /* buffer k was released when the chunk three back was written: at most two chunks are unfinished at a time (three with a chunk prepared ahead) */
        if (threaded) { if (!la.taken) sem_wait(&rs.filled[k]); }
        else { pipe_wait_slot_written(); read_chunk(&fl, done, done == 0 ? rs.chunk_cap / CHUNK_BATCHES : rs.chunk_cap, rs.items[k], &rs.nitem[k]); }
        t_wait += now_s() - tw0;
        const int nk = rs.nitem[k];              /* (the reader may refill the buffer as soon as the chunk is written) */
        la.taken = 0; la.kn = (k + 1) % NCHUNKBUF; la.more = done + (size_t)nk < fl.n;
        pipe_chunk(eng, mdl, rs.items[k], nk, k, hdf5out, look_ahead_reader);
        done += nk;
    }
    pipe_drain(mdl, hdf5out);
    if (threaded) pthread_join(reader, NULL);
    stop_reader_procs();
    for (int k = 0; k < NCHUNKBUF; k++) free(rs.items[k]);
    free(fl.path);
    if (hdf5out >= 0) { pthread_mutex_lock(&hdf5_lock); H5Fclose(hdf5out); pthread_mutex_unlock(&hdf5_lock); }
    if (stdout != args.output) fclose(args.output);
    for (int k = 0; k < NINFLIGHT; k++) if (batch_cache[k].b) ffhip_batch_destroy(batch_cache[k].b);
    for (int k = 0; k < NINFLIGHT; k++) if (pack_cache[k].b) ffhip_batch_destroy(pack_cache[k].b);
    if (getenv("FLAPPIE_CLI_TIMING")) {
        for (int k = 0; k < 8; k++) fprintf(stderr, "%-24s %8.3f s\n", phase_name[k], t_phase[k]);
        fprintf(stderr, "%-24s %8.3f s\n%-24s %8.3f s\n%-24s %8.3f s\n", "list files", t_listed - t_start, "waiting for the reader", t_wait,
                "files listed -> done", now_s() - t_listed);
        fprintf(stderr, "basecalled: %llu reads, %llu samples (trimmed ranges), %llu raw samples\n", n_called_reads, n_called_samples, n_raw_samples);
        if (n_batches) fprintf(stderr, "batches: %llu (%llu of them packed), %.1f reads each; padding efficiency %.3f by batch (samples / (slots x the batch's longest read, or row)), %.3f by read tile\n", n_batches, n_packed_batches,
                               (double)n_called_reads / (double)n_batches, (double)n_batch_samples / (double)(n_batch_slot_samples ? n_batch_slot_samples : 1),
                               (double)n_batch_samples / (double)(n_tile_slot_samples ? n_tile_slot_samples : 1));
    }
    /* reads with a sample so far out that the default path's operand format could not hold the convolution's output: none is returned
     * clamped, the engine ran them again on its f32 kernels (include/ffhip.h, ffhip_engine_f32_reruns) */
    if (ffhip_engine_f32_reruns(eng) > 0)
        warnx("%llu read(s) held samples beyond the range of the default kernels' operand format and were evaluated on the f32 kernels", ffhip_engine_f32_reruns(eng));
    flappie_hip_shutdown();
    if (reader_failures) { warnx("%d reader process(es) failed; see the warnings above", reader_failures); return EXIT_FAILURE; }
    return EXIT_SUCCESS;
}
