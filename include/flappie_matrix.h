/*  flappie_matrix.h -- matrix type of the drop-in boundary.
 *
 *  Replaces /root/reference/src/flappie_matrix.h:18-83.  The first five members of `_Mat`
 *  (nr, nrq, nc, stride, data) keep the reference's order, meaning and the `.data.f` spelling, so
 *  that `.mdl` model headers -- which are C designated initialisers
 *  `_Mat _NAME = { .nr=, .nrq=, .nc=, .stride=, .data.f = __NAME };`
 *  (misc/taiyaki_flipflop5_guppy.py:52-54) -- compile unchanged against this header.
 *  The reference declares `data.v` as `__m128 *`; here it is `void *` (no x86 intrinsics on this
 *  side of the boundary).  Two members are appended: a device mirror and its state.
 *
 *  Layout: column-major fp32, each column padded with zeros to a multiple of 4 rows
 *  (stride = 4 * nrq), 16-byte aligned, zero-filled on creation (flappie_matrix.c:20-51).
 */
#ifndef FFHIP_FLAPPIE_MATRIX_H
#define FFHIP_FLAPPIE_MATRIX_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    size_t nr, nrq, nc, stride;
    union {
        void *v;
        float *f;
    } data;
    /* -- extension (zero in static .mdl initialisers) -- */
    void *dev;            /* HIP device buffer holding the same [nc][stride] image, or NULL */
    int dev_state;        /* 0 = host only, 1 = host and device in sync, 2 = DEVICE NEWER: data.f is stale until flappie_matrix_sync() */
} _Mat;

typedef struct {
    size_t nr, nrq, nc, stride;
    union {
        void *v;
        int32_t *f;
    } data;
} _iMat;

typedef _Mat *flappie_matrix;
typedef _iMat *flappie_imatrix;
typedef _Mat const *const_flappie_matrix;
typedef _iMat const *const_flappie_imatrix;

/* flappie_matrix.h:39-43 / flappie_matrix.c:20-148.  NULL on allocation failure;
 * free_* returns NULL for the `x = free_flappie_matrix(x)` idiom. */
flappie_matrix make_flappie_matrix(size_t nr, size_t nc);
flappie_matrix remake_flappie_matrix(flappie_matrix M, size_t nr, size_t nc);
flappie_matrix copy_flappie_matrix(const_flappie_matrix mat);
flappie_matrix free_flappie_matrix(flappie_matrix mat);
void zero_flappie_matrix(flappie_matrix M);
flappie_matrix mat_from_array(const float *x, size_t nr, size_t nc);
float *array_from_flappie_matrix(const_flappie_matrix mat);
bool equality_flappie_matrix(const_flappie_matrix mat1, const_flappie_matrix mat2, const float tol);

/* ---- device images (this boundary's addition) ------------------------------------------------------
 * The matrices of the hot path live in HBM: calculate_transitions() returns its scores as a device image (dev_state 2), and every
 * function of decode.h / layers.h / this header that is given a matrix with a device image reads THAT (no upload) and leaves its
 * result on the device (no download).  The library's own host-side readers (array_from_flappie_matrix, fprint_*, equality_*,
 * min / max / validate_*, copy_*) synchronise first; code that reads `->data.f` itself calls flappie_matrix_sync() before it does.
 * FLAPPIE_HOST_MATRICES=1 in the environment restores the reference's behaviour (host images always current, a copy each way per call). */
void flappie_matrix_sync(const_flappie_matrix mat);          /* host image := device image, if that is the newer one (dev_state 2 -> 1) */
bool flappie_matrix_to_device(flappie_matrix mat);           /* device image := host image (dev_state 1): operators on it now stay on the device */
void flappie_matrix_host_changed(flappie_matrix mat);        /* the caller wrote data.f: the device image is dropped (dev_state 0) */

/* flappie_matrix.h:56-61 */
flappie_imatrix make_flappie_imatrix(size_t nr, size_t nc);
flappie_imatrix remake_flappie_imatrix(flappie_imatrix M, size_t nr, size_t nc);
flappie_imatrix free_flappie_imatrix(flappie_imatrix mat);
int32_t *array_from_flappie_imatrix(const_flappie_imatrix mat);
flappie_imatrix copy_flappie_imatrix(const_flappie_imatrix mat);
void zero_flappie_imatrix(flappie_imatrix M);

/* flappie_matrix.h:48-55,75-83: host-side inspection helpers.  validate_* always check (the reference's
 * Release build compiles them to `return true`); min_flappie_matrix returns the minimum (the reference's
 * loop, flappie_matrix.c:487-502, compares the wrong way round and returns the maximum). */
void fprint_flappie_matrix(FILE *fh, const char *header, const_flappie_matrix mat, size_t nr, size_t nc, bool include_padding);
bool validate_flappie_matrix(flappie_matrix mat, float lower, const float upper, const float maskval, const bool only_finite,
                             const char *file, const int line);
float min_flappie_matrix(const_flappie_matrix mat);
float max_flappie_matrix(const_flappie_matrix mat);
bool validate_ivector(int *vec, const size_t n, const int lower, const int upper, const char *file, const int line);
bool validate_vector(float *vec, const size_t n, const float lower, const float upper, const char *file, const int line);
void clip_matrix_inplace(flappie_matrix C, float thresh);
void filter_matrix_inplace(flappie_matrix C, float fill_val, float thresh);
void difference_matrix_inplace(flappie_matrix C, float val);

/* flappie_matrix.h:67-73,79: computed on the GPU (host/layers.c -> ffhip_op_affine / ffhip_op_row_normalise /
 * ffhip_op_activation) */
flappie_matrix affine_map(const_flappie_matrix X, const_flappie_matrix W, const_flappie_matrix b, flappie_matrix C);
flappie_matrix affine_map2(const_flappie_matrix Xf, const_flappie_matrix Xb, const_flappie_matrix Wf, const_flappie_matrix Wb,
                           const_flappie_matrix b, flappie_matrix C);
void row_normalise_inplace(flappie_matrix C);
void log_row_normalise_inplace(flappie_matrix C);
void shift_scale_matrix_inplace(flappie_matrix sigmat, float shift, float scale);

#ifdef __cplusplus
}
#endif
#endif
