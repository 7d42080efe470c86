/*  mdl_loader.c -- run-time reader of flappie's `.mdl` model headers.
 *
 *  A `.mdl` is C source written by the reference's dump scripts (misc/taiyaki_flipflop5_guppy.py:38-99,
 *  misc/taiyaki_flipflop_guppy.py:92-133):
 *      float __NAME[] = { <C99 hex floats, one text line per matrix column> };
 *      _Mat _NAME = { .nr = R, .nrq = Q, .nc = C, .stride = S, .data.f = __NAME };
 *      const flappie_matrix NAME = &_NAME;
 *      #define <conv prefix>stride N
 *  The reference #includes it (networks.c:10-14); this reader consumes the same bytes at run time so a
 *  shipped model drops in without recompiling.  (Compile-time inclusion also still works: the
 *  initialisers above are valid against include/flappie_matrix.h.)
 */
#include <ctype.h>
#include <err.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "mdl_loader.h"

static char *read_file(const char *path, size_t *len) {
    FILE *fh = fopen(path, "rb");
    if (NULL == fh) return NULL;
    fseek(fh, 0, SEEK_END);
    long n = ftell(fh);
    fseek(fh, 0, SEEK_SET);
    if (n < 0) { fclose(fh); return NULL; }
    char *buf = malloc((size_t)n + 1);
    if (NULL == buf) { fclose(fh); return NULL; }
    if (fread(buf, 1, (size_t)n, fh) != (size_t)n) { free(buf); fclose(fh); return NULL; }
    buf[n] = 0;
    fclose(fh);
    *len = (size_t)n;
    return buf;
}

static mdl_tensor *find_tensor(mdl_file *m, const char *name) {
    for (size_t i = 0; i < m->ntensor; i++)
        if (0 == strcmp(m->tensor[i].name, name)) return &m->tensor[i];
    return NULL;
}

const_flappie_matrix mdl_matrix(const mdl_file *m, const char *name) {
    for (size_t i = 0; i < m->ntensor; i++)
        if (0 == strcmp(m->tensor[i].name, name) && m->tensor[i].have_mat) return &m->tensor[i].mat;
    return NULL;
}

int mdl_define(const mdl_file *m, const char *name, int fallback) {
    for (size_t i = 0; i < m->ndefine; i++)
        if (0 == strcmp(m->define[i].name, name)) return m->define[i].value;
    return fallback;
}

void mdl_free(mdl_file *m) {
    if (NULL == m) return;
    for (size_t i = 0; i < m->ntensor; i++) free(m->tensor[i].values);
    free(m->tensor);
    free(m->define);
    free(m);
}

static void copy_ident(char *dst, size_t cap, const char *p, const char **end) {
    size_t n = 0;
    while ((isalnum((unsigned char)*p) || *p == '_') && n + 1 < cap) dst[n++] = *p++;
    dst[n] = 0;
    *end = p;
}

mdl_file *mdl_load(const char *path) {
    size_t len = 0;
    char *text = read_file(path, &len);
    if (NULL == text) { warnx("Cannot read model file %s", path); return NULL; }
    if (0 == strncmp(text, "version https://git-lfs", 23)) {
        warnx("%s is a git-LFS pointer stub, not a model (fetch the LFS object)", path);
        free(text);
        return NULL;
    }
    mdl_file *m = calloc(1, sizeof(*m));
    size_t cap_t = 64, cap_d = 32;
    if (NULL == m) { warnx("%s: out of memory", path); free(text); return NULL; }
    m->tensor = calloc(cap_t, sizeof(mdl_tensor));
    m->define = calloc(cap_d, sizeof(mdl_define_t));
    if (NULL == m->tensor || NULL == m->define) goto oom;
    const char *p = text;
    while (*p) {
        if (0 == strncmp(p, "float __", 8)) {                       /* value array */
            if (m->ntensor == cap_t) {
                mdl_tensor *bigger = realloc(m->tensor, 2 * cap_t * sizeof(mdl_tensor));
                if (NULL == bigger) goto oom;
                m->tensor = bigger;
                cap_t *= 2;
            }
            mdl_tensor *t = &m->tensor[m->ntensor];
            memset(t, 0, sizeof(*t));
            const char *q;
            copy_ident(t->name, sizeof(t->name), p + 8, &q);
            q = strchr(q, '{');
            if (NULL == q) break;
            q++;
            size_t cap = 1024;
            t->values = malloc(cap * sizeof(float));
            m->ntensor++;                                           /* (counted now so that mdl_free releases its values on every exit) */
            if (NULL == t->values) goto oom;
            while (*q && *q != '}') {
                while (*q && (isspace((unsigned char)*q) || *q == ',')) q++;
                if (*q == '}' || !*q) break;
                char *e;
                const float v = strtof(q, &e);
                if (e == q) { q++; continue; }
                if (t->nvalue == cap) {
                    float *bigger = realloc(t->values, 2 * cap * sizeof(float));
                    if (NULL == bigger) goto oom;
                    t->values = bigger;
                    cap *= 2;
                }
                t->values[t->nvalue++] = v;
                q = e;
            }
            p = q;
        } else if (0 == strncmp(p, "_Mat _", 6)) {                  /* matrix header */
            char name[MDL_NAME_MAX];
            const char *q;
            copy_ident(name, sizeof(name), p + 6, &q);
            const char *end = strchr(q, ';');
            mdl_tensor *t = find_tensor(m, name);
            if (t && end) {
                unsigned long nr = 0, nrq = 0, nc = 0, stride = 0;
                const char *f;
                if ((f = strstr(q, ".nr =")) && f < end) sscanf(f, ".nr = %lu", &nr);
                if ((f = strstr(q, ".nrq =")) && f < end) sscanf(f, ".nrq = %lu", &nrq);
                if ((f = strstr(q, ".nc =")) && f < end) sscanf(f, ".nc = %lu", &nc);
                if ((f = strstr(q, ".stride =")) && f < end) sscanf(f, ".stride = %lu", &stride);
                if (nr && nc && stride && (size_t)nc * stride == t->nvalue) {
                    t->mat.nr = nr; t->mat.nrq = nrq; t->mat.nc = nc; t->mat.stride = stride;
                    t->mat.data.f = t->values; t->mat.dev = NULL; t->mat.dev_state = 0;
                    t->have_mat = 1;
                } else {
                    warnx("%s: tensor %s has %zu values but header says %lu x %lu (stride %lu)", path, name, t->nvalue, nr, nc, stride);
                }
            }
            p = end ? end : q;
        } else if (0 == strncmp(p, "#define", 7)) {
            const char *q = p + 7;
            while (*q == ' ' || *q == '\t') q++;
            if (m->ndefine == cap_d) {
                mdl_define_t *bigger = realloc(m->define, 2 * cap_d * sizeof(mdl_define_t));
                if (NULL == bigger) goto oom;
                m->define = bigger;
                cap_d *= 2;
            }
            mdl_define_t *d = &m->define[m->ndefine];
            copy_ident(d->name, sizeof(d->name), q, &q);
            while (*q == ' ' || *q == '\t') q++;
            char *e;
            const long v = strtol(q, &e, 10);
            if (e != q && d->name[0]) { d->value = (int)v; m->ndefine++; }
            p = q;
        } else {
            p++;
            continue;
        }
    }
    free(text);
    /* a value array without a consistent `_Mat` header means a truncated or damaged file: fail the load here, with the
     * tensor's name, rather than later as "model file lacks tensor" */
    for (size_t i = 0; i < m->ntensor; i++) {
        if (!m->tensor[i].have_mat) {
            warnx("%s: tensor %s has %zu values but no matching _Mat header (truncated file?)", path, m->tensor[i].name, m->tensor[i].nvalue);
            mdl_free(m);
            return NULL;
        }
    }
    return m;
oom:
    warnx("%s: out of memory while parsing", path);
    free(text);
    mdl_free(m);
    return NULL;
}
