/*  fast5_interface.h -- single-read fast5 input and `--trace` HDF5 output.
 *  Same signatures as /root/reference/src/fast5_interface.h:17-23.  Built only when libhdf5 is available
 *  (FLAPPIE_HAVE_HDF5); the HIP engine does not depend on it.
 */
#ifndef FFHIP_FAST5_INTERFACE_H
#define FFHIP_FAST5_INTERFACE_H
#include <hdf5.h>
#include <stdbool.h>
#include "flappie_structures.h"

#ifdef __cplusplus
extern "C" {
#endif

/* fast5_interface.c:231-318: /Raw/Reads/<first entry>/Signal + its `read_id`, optionally scaled to pA with
 * /UniqueGlobalKey/channel_id {digitisation, offset, range}.  raw == NULL on failure. */
raw_table read_raw(const char *filename, bool scale_to_pA);
/* the same through libhdf5 only: read_raw tries host/fast5_raw.c first (a single-read file walked in memory, no libhdf5 call) and comes here for every file
 * that reader does not know (not in the reference's header) */
raw_table read_raw_hdf5(const char *filename, bool scale_to_pA);
/* fast5_interface.c:59-74: -1 if filename is NULL; opens an existing file read-write, else creates it */
hid_t open_or_create_hdf5(const char *filename);
/* fast5_interface.c:321-349: group `readname` with `signal` (f32, trimmed normalised signal) and `trace`
 * (u8 [nblock+1 x nstate]), shuffle + deflate when compression_level > 0 */
void write_summary(hid_t hdf5file, const char *readname, const struct _raw_basecall_info res, hsize_t chunk_size,
                   int compression_level);

/* The same group in two steps, for writers that want the compression off the HDF5 lock: summary_pack_create does the
 * filters' work (shuffle + deflate per chunk, the u8 conversion of the trace) WITHOUT calling libhdf5 -- any number of threads may
 * run it --, summary_pack_write creates the datasets with the properties write_summary uses and hands the finished chunks over
 * (H5Dwrite_chunk).  A file written this way reads back exactly like one written by write_summary. */
typedef struct summary_pack summary_pack;
summary_pack *summary_pack_create(const struct _raw_basecall_info res, hsize_t chunk_size, int compression_level);
void summary_pack_write(hid_t hdf5file, const char *readname, const summary_pack *pack);
void summary_pack_free(summary_pack *pack);

#ifdef __cplusplus
}
#endif
#endif
