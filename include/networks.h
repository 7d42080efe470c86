/*  networks.h -- model registry of the drop-in boundary.
 *
 *  Same names, enum values, argument meaning and error behaviour as
 *  /root/reference/src/networks.h:16-42 and networks.c:21-111.  The reference compiles the model
 *  weights into the binary (`#include "models/flipflop5_r941native.h"`, networks.c:10-14); here a
 *  model is the SAME `.mdl` text (misc/taiyaki_flipflop5_guppy.py:38-99) parsed at first use from
 *  `$FLAPPIE_MODEL_DIR/<header name the reference includes>` and kept resident in HBM.
 *
 *  One read per call cannot fill a GPU: these single-read entry points exist so that existing host
 *  code keeps compiling; throughput callers use the batch API in ffhip.h (same engine underneath).
 */
#ifndef FFHIP_NETWORKS_H
#define FFHIP_NETWORKS_H
#include <stdbool.h>
#include "flappie_matrix.h"
#include "flappie_structures.h"

#ifdef __cplusplus
extern "C" {
#endif

/* a model's single-read network: signal and temperature in, transition score matrix out (networks.h:16) */
typedef flappie_matrix (*transition_function_ptr)(const raw_table, float);

/* Registry slots; numeric values are part of the ABI (networks.h:18-26).  The two *_INVALID entries double as the
 * counts of flappie and runnie models (networks.h:28-29). */
enum model_type {
    FLAPPIE_MODEL_R941_NATIVE = 0, FLAPPIE_MODEL_R941_RNA002 = 1, FLAPPIE_MODEL_R941_5mC = 2, FLAPPIE_MODEL_R103_NATIVE = 3,
    FLAPPIE_MODEL_INVALID = 4,
    RUNNIE_MODEL_R941_NATIVE = 5,
    RUNNIE_MODEL_INVALID = 6
};
static const enum model_type flappie_nmodel = FLAPPIE_MODEL_INVALID;
static const enum model_type runnie_nmodel = (enum model_type)(RUNNIE_MODEL_INVALID - FLAPPIE_MODEL_INVALID);

/* name <-> slot (networks.c:21-83); an invalid slot ends the process through errx(EXIT_FAILURE), as the reference */
enum model_type get_flappie_model_type(const char *name);
const char *flappie_model_string(enum model_type slot);
const char *flappie_model_description(enum model_type slot);

/* networks.c:86-111.  calculate_transitions returns a host matrix [nstate*(nbase+1) x nblock] owned by the caller
 * (free_flappie_matrix), or NULL if read.n == 0, read.raw == NULL (networks.c:540-541), the model file is missing,
 * or the GPU path fails. */
transition_function_ptr get_transition_function(enum model_type slot);
flappie_matrix calculate_transitions(raw_table read, float temperature, enum model_type slot);

/* the per-model entry points behind get_transition_function (networks.c:725-743) */
flappie_matrix flipflop5_transitions_r941native(raw_table read, float temperature);       /* LSTM x5, flip-flop head   */
flappie_matrix flipflop5_transitions_r941rna002(raw_table read, float temperature);
flappie_matrix flipflop5_transitions_r103native(raw_table read, float temperature);
flappie_matrix flipflop_transitions_r941native5mC(raw_table read, float temperature);     /* GRUmod x5, 5-base alphabet */
flappie_matrix runlength5_transitions_r941native(raw_table read, float temperature);      /* LSTM x5, run-length head   */

/* ---- additions ---------------------------------------------------------------------------- */
struct ffhip_engine;
struct ffhip_model;
/* process-wide engine (device $FLAPPIE_HIP_DEVICE, default 0) and resident model of the registry */
struct ffhip_engine *flappie_hip_engine(void);
const struct ffhip_model *flappie_hip_model(enum model_type model);
/* load a `.mdl` file for a registry slot explicitly (overrides $FLAPPIE_MODEL_DIR); 0 on success */
int flappie_hip_load_model(enum model_type model, const char *mdl_path);
void flappie_hip_shutdown(void);

#ifdef __cplusplus
}
#endif
#endif
