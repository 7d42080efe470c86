/*  fast5_raw.h -- a single-read fast5 file read without libhdf5 (fast5_raw.c): the fast path of read_raw (fast5_interface.c).
 *  Replaces, for the files it knows, the libhdf5 calls of /root/reference/src/fast5_interface.c:231-318. */
#ifndef FFHIP_FAST5_RAW_H
#define FFHIP_FAST5_RAW_H
#include <stddef.h>

typedef struct { char *uuid; float *raw; size_t n; } fast5_raw_read;

/* 1: `out` holds malloc'd `uuid` (the read_id attribute) and `raw` (n samples, scaled to pA if asked) -- the values read_raw's
 * libhdf5 path gives; 0: a file (or a part of it) this reader does not know: nothing allocated, ask libhdf5. */
int fast5_read_raw_fast(const char *filename, int scale_to_pA, fast5_raw_read *out);
#endif
