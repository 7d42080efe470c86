/*  flappie_common.h / util.h subset -- host-side signal preparation of the drop-in boundary.
 *  Same signatures as /root/reference/src/flappie_common.h:15-16 and util.h (quantilef, medianf, madf,
 *  medmad_normalise_array: util.c:100-212).  These run on the host (O(n log n) per read, SURVEY.md
 *  section 8f row N1) and feed ffhip_batch_set_reads().
 */
#ifndef FFHIP_FLAPPIE_COMMON_H
#define FFHIP_FLAPPIE_COMMON_H
#include <stddef.h>
#include "flappie_structures.h"

#ifdef __cplusplus
extern "C" {
#endif

/* util.c:100-212 */
void quantilef(const float *x, size_t nx, float *p, size_t np);
float medianf(const float *x, size_t n);
float madf(const float *x, size_t n, const float *med);
void medmad_normalise_array(float *x, size_t n);
/* util.c:215-223,278-287 (--delta mode) */
void shift_scale_array(float *x, size_t n, float shift, float scale);
void difference_array(float *x, size_t n);
/* util.c:416-438 */
void reverse_char_array(char *x, size_t n);

/* flappie_common.c:13-81.  On failure `raw` is freed and a zeroed table returned, as the reference. */
raw_table trim_and_segment_raw(raw_table rt, size_t trim_start, size_t trim_end, size_t varseg_chunk, float varseg_thresh);
raw_table trim_raw_by_mad(raw_table rt, size_t chunk_size, float proportion);

#ifdef __cplusplus
}
#endif
#endif
