/*  flappie_output.h -- FASTA / FASTQ / SAM records.
 *  Same enum, names and byte-for-byte record layout as /root/reference/src/flappie_output.h:18-47 and
 *  flappie_output.c:16-132.
 */
#ifndef FFHIP_FLAPPIE_OUTPUT_H
#define FFHIP_FLAPPIE_OUTPUT_H
#include <stdbool.h>
#include <stdio.h>
#include "flappie_structures.h"

#ifdef __cplusplus
extern "C" {
#endif

enum flappie_outformat_type { FLAPPIE_OUTFORMAT_FASTA, FLAPPIE_OUTFORMAT_FASTQ, FLAPPIE_OUTFORMAT_SAM, FLAPPIE_OUTFORMAT_INVALID };

enum flappie_outformat_type get_outformat(const char *formatstr);
const char *flappie_outformat_string(enum flappie_outformat_type format);
void printf_format(enum flappie_outformat_type outformat, const char *uuid, const char *readname, bool uuid_primary,
                   const char *prefix, const struct _raw_basecall_info res);
void fprintf_format(enum flappie_outformat_type outformat, FILE *fp, const char *uuid, const char *readname,
                    bool uuid_primary, const char *prefix, const struct _raw_basecall_info res);
void fprintf_fasta(FILE *fp, const char *uuid, const char *readname, bool uuid_primary, const char *prefix,
                   const struct _raw_basecall_info res);
void fprintf_fastq(FILE *fp, const char *uuid, const char *readname, bool uuid_primary, const char *prefix,
                   const struct _raw_basecall_info res);
void fprintf_sam(FILE *fp, const char *uuid, const char *readname, bool uuid_primary, const char *prefix,
                 const struct _raw_basecall_info res);

#ifdef __cplusplus
}
#endif
#endif
