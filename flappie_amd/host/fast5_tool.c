/*  fast5_tool -- test utility: write single-read fast5 files and dump trace files, with the HDF5 C API.
 *    fast5_tool write  out.fast5 READ_ID digitisation offset range sampling_rate samples.i16
 *    fast5_tool dump   trace.hdf5 GROUP          (prints "signal N" + values, "trace R C" + values)
 *  Layout written: /Raw/Reads/Read_1/Signal (int16) with attribute read_id (fixed string) and
 *  /UniqueGlobalKey/channel_id {digitisation, offset, range, sampling_rate} (doubles), i.e. what
 *  read_raw (fast5_interface.c:231-318) consumes.
 */
#include <hdf5.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static void dattr(hid_t g, const char *name, double v) {
    hid_t s = H5Screate(H5S_SCALAR);
    hid_t a = H5Acreate(g, name, H5T_IEEE_F64LE, s, H5P_DEFAULT, H5P_DEFAULT);
    H5Awrite(a, H5T_NATIVE_DOUBLE, &v);
    H5Aclose(a); H5Sclose(s);
}

int main(int argc, char **argv) {
    if (argc >= 9 && 0 == strcmp(argv[1], "write")) {
        FILE *fh = fopen(argv[8], "rb");
        if (!fh) return 2;
        fseek(fh, 0, SEEK_END); long bytes = ftell(fh); fseek(fh, 0, SEEK_SET);
        hsize_t n = (hsize_t)(bytes / 2);
        short *raw = malloc(bytes);
        if (fread(raw, 2, n, fh) != n) return 2;
        fclose(fh);
        hid_t f = H5Fcreate(argv[2], H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
        hid_t g1 = H5Gcreate(f, "/Raw", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
        hid_t g2 = H5Gcreate(f, "/Raw/Reads", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
        hid_t g3 = H5Gcreate(f, "/Raw/Reads/Read_1", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
        hid_t st = H5Tcopy(H5T_C_S1); H5Tset_size(st, strlen(argv[3]) + 1);
        hid_t ss = H5Screate(H5S_SCALAR);
        hid_t a = H5Acreate(g3, "read_id", st, ss, H5P_DEFAULT, H5P_DEFAULT);
        H5Awrite(a, st, argv[3]);
        H5Aclose(a); H5Sclose(ss); H5Tclose(st);
        hid_t sp = H5Screate_simple(1, &n, NULL);
        hid_t d = H5Dcreate(g3, "Signal", H5T_STD_I16LE, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
        H5Dwrite(d, H5T_NATIVE_SHORT, H5S_ALL, H5S_ALL, H5P_DEFAULT, raw);
        H5Dclose(d); H5Sclose(sp);
        hid_t u1 = H5Gcreate(f, "/UniqueGlobalKey", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
        hid_t u2 = H5Gcreate(f, "/UniqueGlobalKey/channel_id", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
        dattr(u2, "digitisation", atof(argv[4])); dattr(u2, "offset", atof(argv[5]));
        dattr(u2, "range", atof(argv[6])); dattr(u2, "sampling_rate", atof(argv[7]));
        H5Gclose(u2); H5Gclose(u1); H5Gclose(g3); H5Gclose(g2); H5Gclose(g1); H5Fclose(f);
        free(raw);
        return 0;
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
