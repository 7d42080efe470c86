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

/* One read's raw signal (flappie_structures.h:16-22; member order is ABI).  Passed BY VALUE; the callee reads
 * raw[start..end) and never takes ownership of `raw` or `uuid`. */
typedef struct {
    char *uuid;                 /* read identifier from the fast5 file, or NULL         */
    size_t n, start, end;       /* samples in raw[]; the range kept after trimming      */
    float *raw;                 /* n samples                                             */
} raw_table;

/* Everything the CLI keeps per called read (flappie_structures.h:24-35; member order is ABI). */
struct _raw_basecall_info {
    float score;                /* path score of the decode                              */
    raw_table rt;
    char *basecall, *quality;   /* NUL-terminated, basecall_length characters each       */
    size_t basecall_length;
    flappie_imatrix trace;      /* [nstate x nblock+1] or NULL                           */
    int *pos;                   /* change positions (unused by the batched driver)       */
    size_t nblock;
};

/* Small accessors used by the batched driver and the engine's argument checks (not in the reference). */
static inline size_t raw_table_kept(const raw_table *rt) { return (rt && rt->raw && rt->end > rt->start) ? rt->end - rt->start : 0; }
static inline const float *raw_table_first(const raw_table *rt) { return (rt && rt->raw) ? rt->raw + rt->start : NULL; }
static inline int raw_table_valid(const raw_table *rt) { return rt && rt->raw && rt->n > 0 && rt->start <= rt->end && rt->end <= rt->n; }

/* release the owned members and NULL them (flappie_structures.c:13-24) */
void free_raw_table(raw_table *tbl);
void free_raw_basecall_info(struct _raw_basecall_info *ptr);

#ifdef __cplusplus
}
#endif
#endif
