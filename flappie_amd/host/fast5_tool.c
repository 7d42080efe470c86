/*  fast5_tool -- test utility: write single-read fast5 files and dump trace files, with the HDF5 C API.
 *    fast5_tool write  out.fast5 READ_ID digitisation offset range sampling_rate samples.i16
 *    fast5_tool synth  DIR COUNT MINLEN MAXLEN SEED [FIRST [STEP]]   COUNT files DIR/read_%06d.fast5 with indices FIRST, FIRST+STEP, ...:
 *                      seeded noise around 500 +- 60 counts behind a 300-sample quiet stretch (what the trimming step removes),
 *                      lengths uniform in [MINLEN, MAXLEN); prints "files N samples S".  The signal of a file depends on SEED and
 *                      its index only, so several processes can fill one directory (bench.py's host-fed leg, tools/cli_throughput.py)
 *    fast5_tool synthln DIR COUNT MEDIAN SIGMA MINLEN MAXLEN SEED [FIRST [STEP]]   the same with LOG-NORMAL lengths: exp(ln MEDIAN + SIGMA z), z ~ N(0, 1),
 *                      clipped to [MINLEN, MAXLEN] -- a nanopore-like length mix (tools/length_mix.py)
 *    fast5_tool dump   trace.hdf5 GROUP          (prints "signal N" + values, "trace R C" + values)
 *  Layout written: /Raw/Reads/Read_1/Signal (int16) with attribute read_id (fixed string) and
 *  /UniqueGlobalKey/channel_id {digitisation, offset, range, sampling_rate} (doubles), i.e. what
 *  read_raw (fast5_interface.c:231-318) consumes.
 */
#include <hdf5.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static void dattr(hid_t g, const char *name, double v) {
    hid_t s = H5Screate(H5S_SCALAR);
    hid_t a = H5Acreate(g, name, H5T_IEEE_F64LE, s, H5P_DEFAULT, H5P_DEFAULT);
    H5Awrite(a, H5T_NATIVE_DOUBLE, &v);
    H5Aclose(a); H5Sclose(s);
}

static int write_read_x(const char *path, const char *read_id, double digitisation, double offset, double range, double rate, const short *raw, hsize_t n, unsigned flags, hsize_t chunk);
static int write_read(const char *path, const char *read_id, double digitisation, double offset, double range, double rate, const short *raw, hsize_t n) {
    return write_read_x(path, read_id, digitisation, offset, range, rate, raw, n, 0, 0);
}

static unsigned long long rng_next(unsigned long long *s) {      /* splitmix64 */
    unsigned long long z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static double rng_unit(unsigned long long *s) { return ((double)(rng_next(s) >> 11) + 0.5) * (1.0 / 9007199254740992.0); }

int main(int argc, char **argv) {
    const int lognormal = argc >= 9 && 0 == strcmp(argv[1], "synthln");
    if (lognormal || (argc >= 7 && 0 == strcmp(argv[1], "synth"))) {
        const int o = lognormal ? 2 : 0;                          /* synthln carries MEDIAN SIGMA in front of MINLEN */
        const double median = lognormal ? atof(argv[4]) : 0.0, sigma = lognormal ? atof(argv[5]) : 0.0;
        const long count = atol(argv[3]), minlen = atol(argv[4 + o]), maxlen = atol(argv[5 + o]);
        const unsigned long long seed = strtoull(argv[6 + o], NULL, 10);
        const long first = argc > 7 + o ? atol(argv[7 + o]) : 0, step = argc > 8 + o ? atol(argv[8 + o]) : 1;
        if (count < 0 || minlen < 1 || maxlen <= minlen || step < 1 || (lognormal && !(median >= 1.0 && sigma >= 0.0))) return 1;
        short *raw = malloc(((size_t)maxlen + 2) * sizeof(short));
        unsigned long long total = 0;
        for (long k = 0; k < count; k++) {
            const long idx = first + k * step;
            unsigned long long st = seed * 0x100000001B3ull + (unsigned long long)idx;
            long n = minlen + (long)(rng_unit(&st) * (double)(maxlen - minlen));
            if (lognormal) {
                const double u = rng_unit(&st), v = rng_unit(&st);
                const double len = median * exp(sigma * sqrt(-2.0 * log(u)) * cos(6.283185307179586 * v));
                n = len < (double)minlen ? minlen : (len > (double)maxlen ? maxlen : (long)len);
            }
            for (long i = 0; i < n; i += 2) {                      /* Box-Muller, two values per draw */
                const double u = rng_unit(&st), v = rng_unit(&st);
                const double r = sqrt(-2.0 * log(u)), a = 6.283185307179586 * v;
                const double g[2] = { r * cos(a), r * sin(a) };
                for (int e = 0; e < 2 && i + e < n; e++) {
                    double x = (i + e < 300) ? 520.0 + 4.0 * g[e] : 500.0 + 60.0 * g[e];
                    x = x < 0.0 ? 0.0 : (x > 8191.0 ? 8191.0 : x);
                    raw[i + e] = (short)(x + 0.5);
                }
            }
            char path[4096], id[64];
            snprintf(path, sizeof(path), "%s/read_%06ld.fast5", argv[2], idx);
            snprintf(id, sizeof(id), "uuid-%06ld", idx);
            if (write_read(path, id, 8192.0, 10.0, 1400.0, 4000.0, raw, (hsize_t)n)) return 2;
            total += (unsigned long long)n;
        }
        free(raw);
        printf("files %ld samples %llu\n", count, total);
        return 0;
    }
    if (argc >= 9 && (0 == strcmp(argv[1], "write") || 0 == strcmp(argv[1], "writex"))) {
        FILE *fh = fopen(argv[8], "rb");
        if (!fh) return 2;
        fseek(fh, 0, SEEK_END); long bytes = ftell(fh); fseek(fh, 0, SEEK_SET);
        hsize_t n = (hsize_t)(bytes / 2);
        short *raw = malloc(bytes);
        if (fread(raw, 2, n, fh) != n) return 2;
        fclose(fh);
        const unsigned flags = (0 == strcmp(argv[1], "writex") && argc > 9) ? (unsigned)strtoul(argv[9], NULL, 0) : 0;
        const hsize_t chunk = (0 == strcmp(argv[1], "writex") && argc > 10) ? (hsize_t)atol(argv[10]) : 0;
        const int rc = write_read_x(argv[2], argv[3], atof(argv[4]), atof(argv[5]), atof(argv[6]), atof(argv[7]), raw, n, flags, chunk);
        free(raw);
        return rc;
    }
    if (argc >= 4 && 0 == strcmp(argv[1], "dump")) {
        hid_t f = H5Fopen(argv[2], H5F_ACC_RDONLY, H5P_DEFAULT);
        if (f < 0) return 2;
        char path[1024];
        snprintf(path, sizeof(path), "/%s/signal", argv[3]);
        hid_t d = H5Dopen(f, path, H5P_DEFAULT);
        if (d < 0) return 3;
        hid_t sp = H5Dget_space(d); hsize_t n = 0; H5Sget_simple_extent_dims(sp, &n, NULL);
        float *x = malloc(n * sizeof(float));
        H5Dread(d, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, x);
        printf("signal %llu\n", (unsigned long long)n);
        for (hsize_t i = 0; i < n; i++) printf("%a\n", x[i]);
        free(x); H5Sclose(sp); H5Dclose(d);
        snprintf(path, sizeof(path), "/%s/trace", argv[3]);
        d = H5Dopen(f, path, H5P_DEFAULT);
        if (d < 0) return 4;
        sp = H5Dget_space(d); hsize_t dims[2] = { 0, 0 }; H5Sget_simple_extent_dims(sp, dims, NULL);
        unsigned char *t = malloc(dims[0] * dims[1]);
        H5Dread(d, H5T_NATIVE_UCHAR, H5S_ALL, H5S_ALL, H5P_DEFAULT, t);
        printf("trace %llu %llu\n", (unsigned long long)dims[0], (unsigned long long)dims[1]);
        for (hsize_t i = 0; i < dims[0] * dims[1]; i++) printf("%u\n", t[i]);
        free(t); H5Sclose(sp); H5Dclose(d); H5Fclose(f);
        return 0;
    }
    fprintf(stderr, "usage: fast5_tool write|dump ...\n");
    return 1;
}

/* FLAGS of `writex` (the layouts a single-read file can come in; tests/test_fast5_raw.py): 1 chunked Signal (CHUNK elements), 2 deflate, 4 shuffle, 8 fletcher32,
 * 16 read_id as a variable-length string, 32 the latest file format (superblock 3, version-2 object headers, link messages), 64 thirty more attributes on
 * the read group and on channel_id (continuation blocks; with 32: dense attribute storage), 128 two more read groups (Read_7, Read_12: name order),
 * 256 channel_id's numbers as float32 / int32 / int64 instead of doubles, 512 forty more groups beside /Raw/Reads/Read_1 (several symbol nodes) */
static void xattr(hid_t g, const char *name, hid_t ftype, hid_t mtype, const void *v) {
    hid_t s = H5Screate(H5S_SCALAR);
    hid_t a = H5Acreate(g, name, ftype, s, H5P_DEFAULT, H5P_DEFAULT);
    H5Awrite(a, mtype, v);
    H5Aclose(a); H5Sclose(s);
}

static int write_read_x(const char *path, const char *read_id, double digitisation, double offset, double range, double rate, const short *raw, hsize_t n, unsigned flags, hsize_t chunk) {
    hid_t fapl = H5P_DEFAULT;
    if (flags & 32) { fapl = H5Pcreate(H5P_FILE_ACCESS); H5Pset_libver_bounds(fapl, H5F_LIBVER_LATEST, H5F_LIBVER_LATEST); }
    hid_t f = H5Fcreate(path, H5F_ACC_TRUNC, H5P_DEFAULT, fapl);
    if (f < 0) return 2;
    hid_t g1 = H5Gcreate(f, "/Raw", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    hid_t g2 = H5Gcreate(f, "/Raw/Reads", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    if (flags & 128) {
        hid_t o1 = H5Gcreate(f, "/Raw/Reads/Read_7", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT); H5Gclose(o1);
        o1 = H5Gcreate(f, "/Raw/Reads/Read_12", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT); H5Gclose(o1);
    }
    hid_t g3 = H5Gcreate(f, (flags & 128) ? "/Raw/Reads/Read_100" : "/Raw/Reads/Read_1", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    if (flags & 512)
        for (int k = 0; k < 40; k++) {
            char nm[64]; snprintf(nm, sizeof(nm), "/Raw/Reads/Z_%02d", k);
            hid_t o1 = H5Gcreate(f, nm, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT); H5Gclose(o1);
        }
    hid_t ss = H5Screate(H5S_SCALAR);
    if (flags & 64)
        for (int k = 0; k < 30; k++) { char nm[64]; double v = k; snprintf(nm, sizeof(nm), "extra_%02d", k); xattr(g3, nm, H5T_IEEE_F64LE, H5T_NATIVE_DOUBLE, &v); }
    if (flags & 16) {
        hid_t st = H5Tcopy(H5T_C_S1); H5Tset_size(st, H5T_VARIABLE);
        hid_t a = H5Acreate(g3, "read_id", st, ss, H5P_DEFAULT, H5P_DEFAULT);
        H5Awrite(a, st, &read_id);
        H5Aclose(a); H5Tclose(st);
    } else {
        hid_t st = H5Tcopy(H5T_C_S1); H5Tset_size(st, strlen(read_id) + 1);
        hid_t a = H5Acreate(g3, "read_id", st, ss, H5P_DEFAULT, H5P_DEFAULT);
        H5Awrite(a, st, read_id);
        H5Aclose(a); H5Tclose(st);
    }
    H5Sclose(ss);
    hid_t sp = H5Screate_simple(1, &n, NULL);
    hid_t dcpl = H5P_DEFAULT;
    if (flags & 15) {
        dcpl = H5Pcreate(H5P_DATASET_CREATE);
        hsize_t ch = (chunk > 0 && chunk < n) ? chunk : (n > 0 ? n : 1);     /* a chunk of a fixed-size dataset is at most the dataset */
        H5Pset_chunk(dcpl, 1, &ch);
        if (flags & 4) H5Pset_shuffle(dcpl);
        if (flags & 2) H5Pset_deflate(dcpl, 1);
        if (flags & 8) H5Pset_fletcher32(dcpl);
    }
    hid_t d = H5Dcreate(g3, "Signal", H5T_STD_I16LE, sp, H5P_DEFAULT, dcpl, H5P_DEFAULT);
    if (d < 0) return 3;
    H5Dwrite(d, H5T_NATIVE_SHORT, H5S_ALL, H5S_ALL, H5P_DEFAULT, raw);
    H5Dclose(d); H5Sclose(sp);
    if (dcpl != H5P_DEFAULT) H5Pclose(dcpl);
    hid_t u1 = H5Gcreate(f, "/UniqueGlobalKey", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    hid_t u2 = H5Gcreate(f, "/UniqueGlobalKey/channel_id", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    if (flags & 64)
        for (int k = 0; k < 30; k++) { char nm[64]; double v = k; snprintf(nm, sizeof(nm), "aaa_%02d", k); xattr(u2, nm, H5T_IEEE_F64LE, H5T_NATIVE_DOUBLE, &v); }
    if (flags & 256) {
        const float dg = (float)digitisation; const int of = (int)offset; const long long rg = (long long)range;
        xattr(u2, "digitisation", H5T_IEEE_F32LE, H5T_NATIVE_FLOAT, &dg);
        xattr(u2, "offset", H5T_STD_I32LE, H5T_NATIVE_INT, &of);
        xattr(u2, "range", H5T_STD_I64LE, H5T_NATIVE_LLONG, &rg);
    } else {
        dattr(u2, "digitisation", digitisation); dattr(u2, "offset", offset); dattr(u2, "range", range);
    }
    dattr(u2, "sampling_rate", rate);
    H5Gclose(u2); H5Gclose(u1); H5Gclose(g3); H5Gclose(g2); H5Gclose(g1); H5Fclose(f);
    if (fapl != H5P_DEFAULT) H5Pclose(fapl);
    return 0;
}
