/*  flappie_matrix.c -- host side of the matrix type (include/flappie_matrix.h).
 *  Behaviour follows /root/reference/src/flappie_matrix.c:20-148,246-359: zero-filled 16-byte aligned
 *  storage, rows padded to a multiple of 4, NULL on failure, free_* returns NULL.
 */
#include <err.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/flappie_matrix.h"
#include "../../include/flappie_structures.h"

flappie_matrix make_flappie_matrix(size_t nr, size_t nc) {
    if (nr == 0 || nc == 0) return NULL;
    flappie_matrix mat = malloc(sizeof(*mat));
    if (NULL == mat) return NULL;
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
    memcpy(C->data.f, M->data.f, sizeof(float) * C->stride * C->nc);
    return C;
}

void zero_flappie_matrix(flappie_matrix M) {
    if (NULL == M) return;
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
    float *res = calloc(mat->nr * mat->nc, sizeof(float));
    if (NULL == res) return NULL;
    for (size_t c = 0; c < mat->nc; c++) memcpy(res + c * mat->nr, mat->data.f + c * mat->stride, mat->nr * sizeof(float));
    return res;
}

bool equality_flappie_matrix(const_flappie_matrix mat1, const_flappie_matrix mat2, const float tol) {
    if (NULL == mat1 || NULL == mat2) return NULL == mat1 && NULL == mat2;
    if (mat1->nc != mat2->nc || mat1->nr != mat2->nr) return false;
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
