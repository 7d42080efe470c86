/*  flappie_matrix.c -- host side of the matrix type (include/flappie_matrix.h).
 *  Behaviour follows /root/reference/src/flappie_matrix.c:20-148,246-359: zero-filled 16-byte aligned
 *  storage, rows padded to a multiple of 4, NULL on failure, free_* returns NULL.
 */
#include <err.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/ffhip.h"
#include "../../include/flappie_matrix.h"
#include "../../include/flappie_structures.h"

static void *forget_device_image(const void *mat);

/* ---- device images (include/flappie_matrix.h) ---- */
void flappie_matrix_sync(const_flappie_matrix mat) {
    if (NULL == mat || 2 != mat->dev_state || NULL == mat->dev) return;
    flappie_matrix m = (flappie_matrix)mat;               /* the mirror state is a cache: logically const */
    if (0 == ffhip_dev_download(m->dev, m->data.f, m->nc * m->stride)) m->dev_state = 1;
    else warnx("%s: %s", __func__, ffhip_last_error());
}

bool flappie_matrix_to_device(flappie_matrix mat) {
    if (NULL == mat) return false;
    if (mat->dev_state >= 1 && NULL != mat->dev) return true;
    if (NULL != mat->dev) { ffhip_dev_release(mat->dev); mat->dev = NULL; }
    mat->dev = ffhip_dev_upload(mat->data.f, mat->nc * mat->stride);
    mat->dev_state = (NULL != mat->dev) ? 1 : 0;
    return NULL != mat->dev;
}

void flappie_matrix_host_changed(flappie_matrix mat) {
    if (NULL == mat) return;
    (void)forget_device_image(mat);
    if (NULL != mat->dev) ffhip_dev_release(mat->dev);
    mat->dev = NULL;
    mat->dev_state = 0;
}

/* The reference's own calculate_post releases the transition matrix with a plain free() (flappie.c:281): the struct goes, its data
 * -- and here its device image -- would leak, one per read.  Matrices handed out with a device image by calculate_transitions are
 * therefore remembered by ADDRESS in the engine library (ffhip_dev_remember: a mutex-guarded map, cleared by whoever releases the image):
 * when malloc returns the address of a remembered struct for a new matrix, the old one was freed behind this library's back, and its
 * device buffer goes back to the pool.  (The host image leaks as it does in the reference; an address malloc never hands out again
 * keeps its buffer: the library warns once should such records pile up.) */
void flappie_matrix_remember_device_image(const_flappie_matrix mat) {
    if (NULL == mat || NULL == mat->dev) return;
    ffhip_dev_remember(mat, mat->dev);
}

static void *forget_device_image(const void *mat) { return ffhip_dev_forget(mat); }

flappie_matrix make_flappie_matrix(size_t nr, size_t nc) {
    if (nr == 0 || nc == 0) return NULL;
    flappie_matrix mat = malloc(sizeof(*mat));
    if (NULL == mat) return NULL;
    {
        void *orphan = forget_device_image(mat);      /* this address was a remembered matrix: it was free()d, not free_flappie_matrix()ed */
        if (NULL != orphan) ffhip_dev_release(orphan);
    }
    mat->nr = nr;
    mat->nrq = (nr + 3) / 4;
    mat->nc = nc;
    mat->stride = mat->nrq * 4;
    mat->dev = NULL;
    mat->dev_state = 0;
    const size_t colbytes = mat->stride * sizeof(float);
    if (colbytes != 0 && (colbytes * nc) / colbytes != nc) { free(mat); return NULL; }    /* overflow */
    void *p = NULL;
    if (0 != posix_memalign(&p, 16, colbytes * nc)) {
        warnx("Error allocating memory in %s.\n", __func__);
        free(mat);
        return NULL;
    }
    memset(p, 0, colbytes * nc);
    mat->data.v = p;
    return mat;
}

flappie_matrix free_flappie_matrix(flappie_matrix mat) {
    if (NULL != mat) {
        (void)forget_device_image(mat);
        if (NULL != mat->dev) ffhip_dev_release(mat->dev);          /* back to the engine's pool */
        free(mat->data.v);
        free(mat);
    }
    return NULL;
}

flappie_matrix remake_flappie_matrix(flappie_matrix M, size_t nr, size_t nc) {
    if ((NULL == M) || (M->nr != nr) || (M->nc != nc)) {
        M = free_flappie_matrix(M);
        M = make_flappie_matrix(nr, nc);
    }
    return M;
}

flappie_matrix copy_flappie_matrix(const_flappie_matrix M) {
    if (NULL == M) return NULL;
    flappie_matrix C = make_flappie_matrix(M->nr, M->nc);
    if (NULL == C) return NULL;
    flappie_matrix_sync(M);
    memcpy(C->data.f, M->data.f, sizeof(float) * C->stride * C->nc);
    return C;
}

void zero_flappie_matrix(flappie_matrix M) {
    if (NULL == M) return;
    flappie_matrix_host_changed(M);
    memset(M->data.f, 0, M->stride * M->nc * sizeof(float));
}

flappie_matrix mat_from_array(const float *x, size_t nr, size_t nc) {
    flappie_matrix res = make_flappie_matrix(nr, nc);
    if (NULL == res || NULL == x) return free_flappie_matrix(res);
    for (size_t c = 0; c < nc; c++) memcpy(res->data.f + c * res->stride, x + c * nr, nr * sizeof(float));
    return res;
}

float *array_from_flappie_matrix(const_flappie_matrix mat) {
    if (NULL == mat) return NULL;
    flappie_matrix_sync(mat);
    float *res = calloc(mat->nr * mat->nc, sizeof(float));
    if (NULL == res) return NULL;
    for (size_t c = 0; c < mat->nc; c++) memcpy(res + c * mat->nr, mat->data.f + c * mat->stride, mat->nr * sizeof(float));
    return res;
}

bool equality_flappie_matrix(const_flappie_matrix mat1, const_flappie_matrix mat2, const float tol) {
    if (NULL == mat1 || NULL == mat2) return NULL == mat1 && NULL == mat2;
    if (mat1->nc != mat2->nc || mat1->nr != mat2->nr) return false;
    flappie_matrix_sync(mat1);
    flappie_matrix_sync(mat2);
    for (size_t c = 0; c < mat1->nc; ++c)
        for (size_t r = 0; r < mat1->nr; ++r)
            if (fabsf(mat1->data.f[c * mat1->stride + r] - mat2->data.f[c * mat2->stride + r]) > tol) return false;
    return true;
}

flappie_imatrix make_flappie_imatrix(size_t nr, size_t nc) {
    if (nr == 0 || nc == 0) return NULL;
    flappie_imatrix mat = malloc(sizeof(*mat));
    if (NULL == mat) return NULL;
    mat->nr = nr;
    mat->nrq = (nr + 3) / 4;
    mat->nc = nc;
    mat->stride = mat->nrq * 4;
    void *p = NULL;
    if (0 != posix_memalign(&p, 16, mat->stride * nc * sizeof(int32_t))) {
        warnx("Error allocating memory in %s.\n", __func__);
        free(mat);
        return NULL;
    }
    memset(p, 0, mat->stride * nc * sizeof(int32_t));
    mat->data.v = p;
    return mat;
}

flappie_imatrix free_flappie_imatrix(flappie_imatrix mat) {
    if (NULL != mat) {
        free(mat->data.v);
        free(mat);
    }
    return NULL;
}

flappie_imatrix remake_flappie_imatrix(flappie_imatrix M, size_t nr, size_t nc) {
    if ((NULL == M) || (M->nr != nr) || (M->nc != nc)) {
        M = free_flappie_imatrix(M);
        M = make_flappie_imatrix(nr, nc);
    }
    return M;
}

int32_t *array_from_flappie_imatrix(const_flappie_imatrix mat) {
    if (NULL == mat) return NULL;
    int32_t *res = calloc(mat->nr * mat->nc, sizeof(int32_t));
    if (NULL == res) return NULL;
    for (size_t c = 0; c < mat->nc; c++)
        for (size_t r = 0; r < mat->nr; r++) res[c * mat->nr + r] = mat->data.f[c * mat->stride + r];
    return res;
}

flappie_imatrix copy_flappie_imatrix(const_flappie_imatrix M) {
    if (NULL == M) return NULL;
    flappie_imatrix C = make_flappie_imatrix(M->nr, M->nc);
    if (NULL == C) return NULL;
    memcpy(C->data.f, M->data.f, sizeof(int32_t) * C->stride * C->nc);
    return C;
}

void zero_flappie_imatrix(flappie_imatrix M) {
    if (NULL == M) return;
    memset(M->data.f, 0, M->stride * M->nc * sizeof(int32_t));
}

/* ---- host-side inspection helpers (flappie_matrix.c:109-232,468-618): bookkeeping on host images, no
 *      arithmetic of the network.  The compute entries of flappie_matrix.h (affine_map, affine_map2,
 *      row_normalise_inplace, log_row_normalise_inplace, shift_scale_matrix_inplace) live in layers.c and run
 *      on the GPU. ---- */
void fprint_flappie_matrix(FILE *fh, const char *header, const_flappie_matrix mat, size_t nr, size_t nc, bool include_padding) {
    if (NULL == fh || NULL == mat) return;
    flappie_matrix_sync(mat);
    const size_t rlim = include_padding ? mat->stride : mat->nr;
    if (nr <= 0 || nr > rlim) nr = rlim;
    if (nc <= 0 || nc > mat->nc) nc = mat->nc;
    if (NULL != header) {
        if (fputs(header, fh) < 0) return;
        fputc('\n', fh);
    }
    for (size_t c = 0; c < nc; c++) {
        const size_t offset = c * mat->stride;
        fprintf(fh, "%4zu : % 12e", c, mat->data.f[offset]);
        for (size_t r = 1; r < nr; r++) fprintf(fh, "  % 12e", mat->data.f[offset + r]);
        fputc('\n', fh);
    }
}

/* flappie_matrix.c:150-232.  The reference compiles this to `return true` under NDEBUG (its Release build);
 * here the checks always run. */
bool validate_flappie_matrix(flappie_matrix mat, float lower, const float upper, const float maskval, const bool only_finite,
                             const char *file, const int line) {
    if (NULL == mat || NULL == mat->data.f || 0 == mat->nc || 0 == mat->nr || mat->stride < mat->nr || mat->nrq * 4 != mat->stride) return false;
    flappie_matrix_sync(mat);
    const size_t nc = mat->nc, nr = mat->nr, ld = mat->stride;
    for (size_t c = 0; c < nc; ++c) {
        const float *col = mat->data.f + c * ld;
        if (!isnan(maskval))
            for (size_t r = nr; r < ld; ++r)
                if (maskval != col[r]) { warnx("%s:%d  Matrix entry [%zu,%zu] = %f violates masking rules\n", file, line, r, c, col[r]); return false; }
        for (size_t r = 0; r < nr; ++r) {
            if (only_finite && !isfinite(col[r])) { warnx("%s:%d  Matrix entry [%zu,%zu] = %f contains a non-finite value\n", file, line, r, c, col[r]); return false; }
            if (!isnan(lower) && col[r] + 1.1920929e-07f < lower) { warnx("%s:%d  Matrix entry [%zu,%zu] = %f (%e) violates lower bound\n", file, line, r, c, col[r], col[r] - lower); return false; }
            if (!isnan(upper) && col[r] > upper + 1.1920929e-07f) { warnx("%s:%d  Matrix entry [%zu,%zu] = %f (%e) violates upper bound\n", file, line, r, c, col[r], col[r] - upper); return false; }
        }
    }
    return true;
}

float max_flappie_matrix(const_flappie_matrix x) {
    if (NULL == x) return NAN;
    flappie_matrix_sync(x);
    float amax = x->data.f[0];
    for (size_t col = 0; col < x->nc; col++)
        for (size_t r = 0; r < x->nr; r++)
            if (amax < x->data.f[col * x->stride + r]) amax = x->data.f[col * x->stride + r];
    return amax;
}

/* flappie_matrix.c:487-502 */
float min_flappie_matrix(const_flappie_matrix x) {
    if (NULL == x) return NAN;
    flappie_matrix_sync(x);
    float amin = x->data.f[0];
    for (size_t col = 0; col < x->nc; col++)
        for (size_t r = 0; r < x->nr; r++)
            if (amin > x->data.f[col * x->stride + r]) amin = x->data.f[col * x->stride + r];
    return amin;
}

bool validate_vector(float *vec, const size_t n, const float lower, const float upper, const char *file, const int line) {
    if (NULL == vec) return false;
    for (size_t i = 0; i < n; ++i) {
        if (!isnan(lower) && lower > vec[i]) { warnx("%s:%d  Vector entry %zu = %f violates lower bound\n", file, line, i, vec[i]); return false; }
        if (!isnan(upper) && upper < vec[i]) { warnx("%s:%d  Vector entry %zu = %f violates upper bound\n", file, line, i, vec[i]); return false; }
    }
    return true;
}

bool validate_ivector(int *vec, const size_t n, const int lower, const int upper, const char *file, const int line) {
    if (NULL == vec) return false;
    for (size_t i = 0; i < n; ++i) {
        if (lower > vec[i]) { warnx("%s:%d  Vector entry %zu = %d violates lower bound\n", file, line, i, vec[i]); return false; }
        if (upper < vec[i]) { warnx("%s:%d  Vector entry %zu = %d violates upper bound\n", file, line, i, vec[i]); return false; }
    }
    return true;
}

/* flappie_matrix.c:647-720: signal-conditioning helpers of the delta-sample path, element selection only */
void clip_matrix_inplace(flappie_matrix C, float thresh) {
    if (NULL == C) return;
    flappie_matrix_sync(C);
    flappie_matrix_host_changed(C);          /* host-side element selection: the host image is the current one afterwards */
    for (size_t c = 0; c < C->nc; c++)
        for (size_t r = 0; r < C->nr; r++) {
            const float obs = C->data.f[c * C->stride + r];
            C->data.f[c * C->stride + r] = copysignf(fminf(thresh, fabsf(obs)), obs);
        }
}

void filter_matrix_inplace(flappie_matrix C, float fill_val, float thresh) {
    if (NULL == C) return;
    flappie_matrix_sync(C);
    flappie_matrix_host_changed(C);          /* host-side element selection: the host image is the current one afterwards */
    for (size_t c = 0; c < C->nc; c++)
        for (size_t r = 0; r < C->nr; r++)
            if (fabsf(C->data.f[c * C->stride + r]) > thresh) C->data.f[c * C->stride + r] = fill_val;
}

void difference_matrix_inplace(flappie_matrix C, float val) {
    if (NULL == C) return;
    flappie_matrix_sync(C);
    flappie_matrix_host_changed(C);          /* host-side element selection: the host image is the current one afterwards */
    for (size_t c = 1; c < C->nc; c++)
        for (size_t r = 0; r < C->nr; r++)
            C->data.f[(c - 1) * C->stride + r] = C->data.f[c * C->stride + r] - C->data.f[(c - 1) * C->stride + r];
    for (size_t r = 0; r < C->nr; r++) C->data.f[(C->nc - 1) * C->stride + r] = val;
}

/* flappie_structures.c:13-24 */
void free_raw_table(raw_table *tbl) {
    if (NULL == tbl) return;
    free(tbl->uuid);
    free(tbl->raw);
    tbl->uuid = NULL;
    tbl->raw = NULL;
}

void free_raw_basecall_info(struct _raw_basecall_info *ptr) {
    if (NULL == ptr) return;
    free_raw_table(&ptr->rt);
    free(ptr->basecall);
    free(ptr->quality);
    free(ptr->pos);
    ptr->trace = free_flappie_imatrix(ptr->trace);
    ptr->basecall = NULL;
    ptr->quality = NULL;
    ptr->pos = NULL;
}
