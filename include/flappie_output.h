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

/* output flavours of the command line (`--format`); INVALID is what get_outformat returns for an unknown name */
enum flappie_outformat_type {
    FLAPPIE_OUTFORMAT_FASTA,
    FLAPPIE_OUTFORMAT_FASTQ,
    FLAPPIE_OUTFORMAT_SAM,
    FLAPPIE_OUTFORMAT_INVALID
};

/* name <-> enum: exactly "fasta", "fastq", "sam" (flappie_output.c:16-41) */
enum flappie_outformat_type get_outformat(const char *name);
const char *flappie_outformat_string(enum flappie_outformat_type fmt);

/* One record per read.  `call` is passed by value as in the reference; the header line carries `prefix`, then the uuid or the
 * file name (whichever `uuid_first` selects) -- byte for byte the reference's layout (flappie_output.c:43-132). */
typedef struct _raw_basecall_info flappie_call_t;
void fprintf_fasta(FILE *out, const char *uuid, const char *filename, bool uuid_first, const char *prefix, const flappie_call_t call);
void fprintf_fastq(FILE *out, const char *uuid, const char *filename, bool uuid_first, const char *prefix, const flappie_call_t call);
void fprintf_sam(FILE *out, const char *uuid, const char *filename, bool uuid_first, const char *prefix, const flappie_call_t call);
/* dispatch on the format; printf_format writes to stdout */
void fprintf_format(enum flappie_outformat_type fmt, FILE *out, const char *uuid, const char *filename, bool uuid_first,
                    const char *prefix, const flappie_call_t call);
void printf_format(enum flappie_outformat_type fmt, const char *uuid, const char *filename, bool uuid_first, const char *prefix,
                   const flappie_call_t call);

#ifdef __cplusplus
}
#endif
#endif
