/*  fast5_interface.c -- single-read fast5 reader and trace writer (include/fast5_interface.h).
 *  Behaviour of /root/reference/src/fast5_interface.c:59-143,209-349.
 */
#include <err.h>
#include <fcntl.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>
#include "../../include/fast5_interface.h"
#include "fast5_raw.h"

static float float_attr(hid_t group, const char *name) {
    float val = NAN;
    hid_t attr = H5Aopen(group, name, H5P_DEFAULT);
    if (attr < 0) { warnx("Failed to open attribute '%s' for reading.", name); return val; }
    H5Aread(attr, H5T_NATIVE_FLOAT, &val);
    H5Aclose(attr);
    return val;
}

/* fixed- or variable-length string attribute -> malloc'd C string (fast5_interface.c:145-206) */
static char *string_attr(hid_t group, const char *name) {
    char *str = NULL;
    hid_t attr = H5Aopen(group, name, H5P_DEFAULT);
    if (attr < 0) { warnx("Failed to open attribute '%s' for reading.", name); return NULL; }
    hid_t atype = H5Aget_type(attr);
    if (atype >= 0 && H5T_STRING == H5Tget_class(atype)) {
        if (H5Tis_variable_str(atype) > 0) {
            char *tmp = NULL;
            if (H5Aread(attr, atype, &tmp) >= 0 && tmp) { str = strdup(tmp); H5free_memory(tmp); }
        } else {
            const size_t asize = H5Tget_size(atype);
            str = calloc(asize + 1, sizeof(char));
            if (str && H5Aread(attr, atype, str) < 0) { free(str); str = NULL; }
        }
    } else {
        warnx("Attribute '%s' is not a string.", name);
    }
    if (atype >= 0) H5Tclose(atype);
    H5Aclose(attr);
    return str;
}

/* How a single-read file is opened for reading (round 5: the host's whole cost per read IS this function -- 115 us a file, 25 ns a sample, and eight
 * ranks of a node ask for 23 CPUs of it, profiles/r04_host_scaling.txt).  A single-read fast5 is 15-40 KB of which libhdf5 touches a dozen places through a
 * pread each, behind a 2 MB metadata cache it sets up per file: the CORE driver reads the file once into memory (no backing store: nothing is written),
 * and the metadata cache starts at 64 KB.  Same bytes, same values.  FLAPPIE_DEBUG=plain_h5open keeps the default driver. */
static hid_t read_fapl(void) {
    static hid_t fapl = -2;
    if (-2 == fapl) {
        fapl = H5P_DEFAULT;
        const char *dbg = getenv("FLAPPIE_DEBUG");
        if (NULL == dbg || NULL == strstr(dbg, "plain_h5open")) {
            hid_t p = H5Pcreate(H5P_FILE_ACCESS);
            if (p >= 0 && H5Pset_fapl_core(p, 1 << 16, 0) >= 0) {
                H5AC_cache_config_t mdc;
                mdc.version = H5AC__CURR_CACHE_CONFIG_VERSION;
                if (H5Pget_mdc_config(p, &mdc) >= 0) {
                    mdc.set_initial_size = 1;
                    mdc.initial_size = 1 << 16;
                    mdc.min_size = 1 << 16;
                    (void)H5Pset_mdc_config(p, &mdc);
                }
                fapl = p;
            } else if (p >= 0) H5Pclose(p);
        }
    }
    return fapl;
}

raw_table read_raw(const char *filename, bool scale_to_pA) {
    if (NULL == filename) return (raw_table){ NULL, 0, 0, 0, NULL };
    {   /* round 6: the files this repo's own reader knows never reach libhdf5 (fast5_raw.c: one read(2), the structures walked in memory: ~105 -> ~10 us
         * a file); everything else -- and every failure -- goes the way it always went.  FLAPPIE_DEBUG=hdf5_read (or plain_h5open) keeps libhdf5 for all. */
        static int use_fast = -1;
        if (use_fast < 0) {
            const char *dbg = getenv("FLAPPIE_DEBUG");
            use_fast = !(NULL != dbg && (NULL != strstr(dbg, "hdf5_read") || NULL != strstr(dbg, "plain_h5open")));
        }
        fast5_raw_read fr;
        if (use_fast && fast5_read_raw_fast(filename, scale_to_pA, &fr)) return (raw_table){ fr.uuid, fr.n, 0, fr.n, fr.raw };
    }
    return read_raw_hdf5(filename, scale_to_pA);
}

/* read_raw through libhdf5 (fast5_interface.c:231-318) */
raw_table read_raw_hdf5(const char *filename, bool scale_to_pA) {
    raw_table rawtbl = { NULL, 0, 0, 0, NULL };
    if (NULL == filename) return rawtbl;
    H5Eset_auto2(H5E_DEFAULT, NULL, NULL);
    hid_t file = H5Fopen(filename, H5F_ACC_RDONLY, read_fapl());
    if (file < 0) { warnx("Failed to open %s for reading.", filename); return rawtbl; }
    /* the one read group under /Raw/Reads, opened by index (one traversal; the reference asks for its name, builds the path and walks it again for the
     * attribute and once more for the dataset: fast5_interface.c:249-283 -- the same objects) */
    char *uuid = NULL;
    hid_t rgroup = H5Oopen_by_idx(file, "/Raw/Reads", H5_INDEX_NAME, H5_ITER_INC, 0, H5P_DEFAULT);
    if (rgroup < 0) { warnx("Failed find read name under %s.", "/Raw/Reads/"); H5Fclose(file); return rawtbl; }
    uuid = string_attr(rgroup, "read_id");
    hid_t dset = H5Dopen(rgroup, "Signal", H5P_DEFAULT);
    if (dset < 0) { warnx("Failed to open dataset '%s' to read raw signal from.", "/Raw/Reads/<read>/Signal"); free(uuid); H5Oclose(rgroup); H5Fclose(file); return rawtbl; }
    static const char path[] = "/Raw/Reads/<read>/Signal";
    hid_t space = H5Dget_space(dset);
    hsize_t nsample = 0;
    if (space >= 0) H5Sget_simple_extent_dims(space, &nsample, NULL);
    float *raw = nsample ? malloc(nsample * sizeof(float)) : NULL;
    /* The reference asks libhdf5 for floats (fast5_interface.c:289) and lets its type-conversion path turn the stored 16-bit integers into
     * them, element by element through a background buffer -- a third of this function's time.  A 16-bit integer dataset (every fast5 file:
     * the Signal of a read is int16) is read as it is stored and converted here: (float)int16 is exact, the values are the same. */
    herr_t got = -1;
    hid_t ftype = (NULL != raw) ? H5Dget_type(dset) : -1;
    if (ftype >= 0 && H5T_INTEGER == H5Tget_class(ftype) && 2 == H5Tget_size(ftype) && H5T_SGN_2 == H5Tget_sign(ftype)) {
        short *tmp = malloc(nsample * sizeof(short));
        if (NULL != tmp && (got = H5Dread(dset, H5T_NATIVE_SHORT, H5S_ALL, H5S_ALL, H5P_DEFAULT, tmp)) >= 0)
            for (hsize_t i = 0; i < nsample; i++) raw[i] = (float)tmp[i];
        free(tmp);
    }
    if (ftype >= 0) H5Tclose(ftype);
    if (NULL == raw || (got < 0 && H5Dread(dset, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, raw) < 0)) {
        warnx("Failed to read raw data from dataset %s.", path);
        free(raw);
        free(uuid);
    } else {
        rawtbl = (raw_table){ uuid, nsample, 0, nsample, raw };
        if (scale_to_pA) {                                    /* fast5_interface.c:209-228,297-303 */
            hid_t g = H5Gopen(file, "/UniqueGlobalKey/channel_id", H5P_DEFAULT);
            if (g < 0) {
                warnx("Failed to group /UniqueGlobalKey/channel_id.");
            } else {
                const float digitisation = float_attr(g, "digitisation"), offset = float_attr(g, "offset"), range = float_attr(g, "range");
                H5Gclose(g);
                const float raw_unit = range / digitisation;
                for (size_t i = 0; i < nsample; i++) raw[i] = (raw[i] + offset) * raw_unit;
            }
        }
    }
    if (space >= 0) H5Sclose(space);
    H5Dclose(dset);
    H5Oclose(rgroup);
    H5Fclose(file);
    return rawtbl;
}

hid_t open_or_create_hdf5(const char *filename) {
    if (NULL == filename) return -1;
    const int fd = open(filename, O_CREAT | O_WRONLY | O_EXCL, S_IRUSR | S_IWUSR);
    if (fd < 0) return H5Fopen(filename, H5F_ACC_RDWR, H5P_DEFAULT);
    close(fd);
    unlink(filename);
    return H5Fcreate(filename, H5F_ACC_EXCL, H5P_DEFAULT, H5P_DEFAULT);
}

static hid_t compression(int rank, hsize_t *chunk, int level) {
    if (level <= 0) return H5P_DEFAULT;
    hid_t p = H5Pcreate(H5P_DATASET_CREATE);
    if (p < 0) return H5P_DEFAULT;
    H5Pset_shuffle(p);
    H5Pset_deflate(p, (unsigned)level);
    H5Pset_chunk(p, rank, chunk);
    return p;
}

void write_summary(hid_t hdf5file, const char *readname, const struct _raw_basecall_info res, hsize_t chunk_size,
                   int compression_level) {
    if (hdf5file < 0 || NULL == readname) return;
    hid_t grp = H5Gcreate(hdf5file, readname, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    if (grp < 0) { warnx("Failed to create group \"%s\" %s:%d.", readname, __FILE__, __LINE__); return; }
    {   /* signal: the trimmed, normalised samples the network saw (fast5_interface.c:332-335) */
        hsize_t n = res.rt.end - res.rt.start, ch = chunk_size < n ? chunk_size : n;
        hid_t space = H5Screate_simple(1, &n, &n);
        hid_t props = compression(1, &ch, compression_level);
        hid_t dset = H5Dcreate(grp, "signal", H5T_IEEE_F32LE, space, H5P_DEFAULT, props, H5P_DEFAULT);
        if (dset >= 0) { H5Dwrite(dset, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, res.rt.raw + res.rt.start); H5Dclose(dset); }
        if (props != H5P_DEFAULT) H5Pclose(props);
        H5Sclose(space);
    }
    int32_t *flat = array_from_flappie_imatrix(res.trace);
    if (NULL != flat) {   /* trace: u8 [nblock+1][nstate] written from int32 (fast5_interface.c:126-143) */
        hsize_t dims[2] = { res.trace->nc, res.trace->nr };
        hsize_t ch[2] = { chunk_size < dims[0] ? chunk_size : dims[0], res.trace->nr };
        hid_t space = H5Screate_simple(2, dims, dims);
        hid_t props = compression(2, ch, compression_level);
        hid_t dset = H5Dcreate(grp, "trace", H5T_STD_U8LE, space, H5P_DEFAULT, props, H5P_DEFAULT);
        if (dset >= 0) { H5Dwrite(dset, H5T_NATIVE_INT, H5S_ALL, H5S_ALL, H5P_DEFAULT, flat); H5Dclose(dset); }
        if (props != H5P_DEFAULT) H5Pclose(props);
        H5Sclose(space);
        free(flat);
    }
    H5Gclose(grp);
}

/* ---- write_summary in two steps: filters off the HDF5 lock (include/fast5_interface.h) -------------------------------- */
#include <zlib.h>

typedef struct { hsize_t dims[2], chunk[2]; int rank; size_t esize, nchunk; void **buf; size_t *len; void *plain; } packed_dset;
struct summary_pack { int level, have_trace; packed_dset sig, tr; };

/* one chunk through HDF5's shuffle (H5Zshuffle.c: byte j of every element together) and deflate (compress2) filters */
static void *filter_chunk(const unsigned char *raw, size_t nelem, size_t esize, int level, size_t *out_len) {
    const size_t nbytes = nelem * esize;
    unsigned char *sh = (unsigned char *)raw, *tmp = NULL;
    if (esize > 1) {
        tmp = malloc(nbytes);
        if (NULL == tmp) return NULL;
        for (size_t j = 0; j < esize; j++)
            for (size_t i = 0; i < nelem; i++) tmp[j * nelem + i] = raw[i * esize + j];
        sh = tmp;
    }
    uLongf dl = compressBound(nbytes);
    void *out = malloc(dl);
    if (NULL == out || Z_OK != compress2(out, &dl, sh, nbytes, level)) { free(out); free(tmp); return NULL; }
    free(tmp);
    *out_len = dl;
    return out;
}

/* `data`: the dataset row-major in its FILE type (esize bytes per element); chunks tile dimension 0 only */
static int pack_dset(packed_dset *d, const void *data, int rank, const hsize_t *dims, hsize_t chunk0, size_t esize, int level) {
    memset(d, 0, sizeof(*d));
    d->rank = rank; d->esize = esize;
    d->dims[0] = dims[0]; d->dims[1] = rank > 1 ? dims[1] : 1;
    d->chunk[0] = chunk0 < dims[0] ? chunk0 : dims[0]; d->chunk[1] = d->dims[1];
    const size_t rowb = (size_t)d->dims[1] * esize, total = (size_t)dims[0] * rowb;
    if (level <= 0 || 0 == dims[0]) {
        d->plain = malloc(total ? total : 1);
        if (NULL == d->plain) return -1;
        memcpy(d->plain, data, total);
        return 0;
    }
    d->nchunk = (size_t)((dims[0] + d->chunk[0] - 1) / d->chunk[0]);
    d->buf = calloc(d->nchunk, sizeof(void *));
    d->len = calloc(d->nchunk, sizeof(size_t));
    const size_t cb = (size_t)d->chunk[0] * rowb;
    unsigned char *full = malloc(cb);                      /* an edge chunk is stored whole, padded with the fill value (0) */
    if (NULL == d->buf || NULL == d->len || NULL == full) { free(full); return -1; }
    for (size_t k = 0; k < d->nchunk; k++) {
        const size_t r0 = k * (size_t)d->chunk[0], nr = (r0 + d->chunk[0] <= dims[0]) ? (size_t)d->chunk[0] : (size_t)(dims[0] - r0);
        const unsigned char *src = (const unsigned char *)data + r0 * rowb;
        if (nr < d->chunk[0]) { memset(full, 0, cb); memcpy(full, src, nr * rowb); src = full; }
        d->buf[k] = filter_chunk(src, (size_t)d->chunk[0] * (size_t)d->dims[1], esize, level, &d->len[k]);
        if (NULL == d->buf[k]) { free(full); return -1; }
    }
    free(full);
    return 0;
}

static void free_dset(packed_dset *d) {
    for (size_t k = 0; k < d->nchunk; k++) free(d->buf ? d->buf[k] : NULL);
    free(d->buf); free(d->len); free(d->plain);
}

summary_pack *summary_pack_create(const struct _raw_basecall_info res, hsize_t chunk_size, int compression_level) {
    summary_pack *p = calloc(1, sizeof(*p));
    if (NULL == p || NULL == res.rt.raw || res.rt.end <= res.rt.start || 0 == chunk_size) { free(p); return NULL; }
    p->level = compression_level;
    hsize_t n = res.rt.end - res.rt.start;
    int rc = pack_dset(&p->sig, res.rt.raw + res.rt.start, 1, &n, chunk_size, sizeof(float), compression_level);
    if (0 == rc && NULL != res.trace) {
        /* the trace leaves as u8 [nblock + 1][nstate]: what H5Dwrite's int32 -> u8 conversion gives (it saturates) */
        const size_t nc = res.trace->nc, nr = res.trace->nr;
        unsigned char *u8 = malloc((nc * nr > 0) ? nc * nr : 1);
        if (NULL == u8) rc = -1;
        else {
            for (size_t c = 0; c < nc; c++)
                for (size_t r = 0; r < nr; r++) {
                    const int32_t v = res.trace->data.f[c * res.trace->stride + r];
                    u8[c * nr + r] = (unsigned char)(v < 0 ? 0 : v > 255 ? 255 : v);
                }
            hsize_t dims[2] = { nc, nr };
            rc = pack_dset(&p->tr, u8, 2, dims, chunk_size, 1, compression_level);
            p->have_trace = (0 == rc);
            free(u8);
        }
    }
    if (0 != rc) { summary_pack_free(p); return NULL; }
    return p;
}

static void write_dset(hid_t grp, const char *name, hid_t ftype, hid_t mtype, const packed_dset *d, int level) {
    hsize_t ch[2] = { d->chunk[0], d->chunk[1] };
    hid_t space = H5Screate_simple(d->rank, d->dims, d->dims);
    hid_t props = (0 == d->dims[0]) ? H5P_DEFAULT : compression(d->rank, ch, level);
    hid_t dset = H5Dcreate(grp, name, ftype, space, H5P_DEFAULT, props, H5P_DEFAULT);
    if (dset >= 0) {
        if (NULL != d->plain) H5Dwrite(dset, mtype, H5S_ALL, H5S_ALL, H5P_DEFAULT, d->plain);
        else
            for (size_t k = 0; k < d->nchunk; k++) {
                const hsize_t off[2] = { (hsize_t)k * d->chunk[0], 0 };
                if (H5Dwrite_chunk(dset, H5P_DEFAULT, 0, off, d->len[k], d->buf[k]) < 0) { warnx("Failed to write chunk %zu of \"%s\" %s:%d.", k, name, __FILE__, __LINE__); break; }
            }
        H5Dclose(dset);
    }
    if (props != H5P_DEFAULT) H5Pclose(props);
    H5Sclose(space);
}

void summary_pack_write(hid_t hdf5file, const char *readname, const summary_pack *p) {
    if (hdf5file < 0 || NULL == readname || NULL == p) return;
    hid_t grp = H5Gcreate(hdf5file, readname, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    if (grp < 0) { warnx("Failed to create group \"%s\" %s:%d.", readname, __FILE__, __LINE__); return; }
    write_dset(grp, "signal", H5T_IEEE_F32LE, H5T_NATIVE_FLOAT, &p->sig, p->level);
    if (p->have_trace) write_dset(grp, "trace", H5T_STD_U8LE, H5T_NATIVE_UCHAR, &p->tr, p->level);
    H5Gclose(grp);
}

void summary_pack_free(summary_pack *p) {
    if (NULL == p) return;
    free_dset(&p->sig);
    free_dset(&p->tr);
    free(p);
}

