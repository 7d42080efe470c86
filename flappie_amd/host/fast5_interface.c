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

raw_table read_raw(const char *filename, bool scale_to_pA) {
    raw_table rawtbl = { NULL, 0, 0, 0, NULL };
    if (NULL == filename) return rawtbl;
    H5Eset_auto2(H5E_DEFAULT, NULL, NULL);
    hid_t file = H5Fopen(filename, H5F_ACC_RDONLY, H5P_DEFAULT);
    if (file < 0) { warnx("Failed to open %s for reading.", filename); return rawtbl; }
    static const char root[] = "/Raw/Reads/";
    const ssize_t size = H5Lget_name_by_idx(file, root, H5_INDEX_NAME, H5_ITER_INC, 0, NULL, 0, H5P_DEFAULT);
    if (size < 0) { warnx("Failed find read name under %s.", root); H5Fclose(file); return rawtbl; }
    char *name = calloc((size_t)size + 1, 1);
    H5Lget_name_by_idx(file, root, H5_INDEX_NAME, H5_ITER_INC, 0, name, (size_t)size + 1, H5P_DEFAULT);
    const size_t plen = sizeof(root) + (size_t)size + 8;
    char *path = calloc(plen, 1);
    snprintf(path, plen, "%s%s", root, name);
    char *uuid = NULL;
    hid_t rgroup = H5Gopen(file, path, H5P_DEFAULT);
    if (rgroup < 0) { warnx("Failed to find read_id under %s.", path); goto done; }
    uuid = string_attr(rgroup, "read_id");
    H5Gclose(rgroup);
    snprintf(path, plen, "%s%s/Signal", root, name);
    hid_t dset = H5Dopen(file, path, H5P_DEFAULT);
    if (dset < 0) { warnx("Failed to open dataset '%s' to read raw signal from.", path); free(uuid); goto done; }
    hid_t space = H5Dget_space(dset);
    hsize_t nsample = 0;
    if (space >= 0) H5Sget_simple_extent_dims(space, &nsample, NULL);
    float *raw = nsample ? calloc(nsample, sizeof(float)) : NULL;
    if (NULL == raw || H5Dread(dset, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, raw) < 0) {
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
done:
    free(path);
    free(name);
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
