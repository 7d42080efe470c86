/*  flappie_structures.h -- types crossing the drop-in boundary.
 *  Replaces /root/reference/src/flappie_structures.h:16-38 (same member order and meaning).
 */
#ifndef FFHIP_FLAPPIE_STRUCTURES_H
#define FFHIP_FLAPPIE_STRUCTURES_H

#include <stddef.h>
#include "flappie_matrix.h"

#ifdef __cplusplus
extern "C" {
#endif

/* flappie_structures.h:16-22.  Passed BY VALUE; the callee reads raw[start..end) and never takes
 * ownership of `raw` or `uuid`. */
typedef struct {
    char *uuid;
    size_t n;
    size_t start;
    size_t end;
    float *raw;
} raw_table;

/* flappie_structures.h:24-35 */
struct _raw_basecall_info {
    float score;
    raw_table rt;

    char *basecall;
    char *quality;
    size_t basecall_length;
    flappie_imatrix trace;

    int *pos;
    size_t nblock;
};

/* flappie_structures.c:13-24 */
void free_raw_table(raw_table *tbl);
void free_raw_basecall_info(struct _raw_basecall_info *ptr);

#ifdef __cplusplus
}
#endif
#endif
