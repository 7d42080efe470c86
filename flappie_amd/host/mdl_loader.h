/* mdl_loader.h -- see mdl_loader.c */
#ifndef FFHIP_MDL_LOADER_H
#define FFHIP_MDL_LOADER_H
#include <stddef.h>
#include "../../include/flappie_matrix.h"

#define MDL_NAME_MAX 128

typedef struct {
    char name[MDL_NAME_MAX];
    float *values;
    size_t nvalue;
    _Mat mat;
    int have_mat;
} mdl_tensor;

typedef struct {
    char name[MDL_NAME_MAX];
    int value;
} mdl_define_t;

typedef struct {
    mdl_tensor *tensor;
    size_t ntensor;
    mdl_define_t *define;
    size_t ndefine;
} mdl_file;

mdl_file *mdl_load(const char *path);
void mdl_free(mdl_file *m);
const_flappie_matrix mdl_matrix(const mdl_file *m, const char *name);
int mdl_define(const mdl_file *m, const char *name, int fallback);
#endif
