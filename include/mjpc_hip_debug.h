/* mjpc_hip_debug.h — diagnostics knobs of libmjpc_hip.so.  NOT part of the drop-in ABI (include/mjpc_hip.h): used by this
 * repository's tests and measurement scripts to force code paths that the engine otherwise picks by itself.  Nothing here is read
 * from the environment; a knob only exists after an explicit call, is process-wide, and is looked at while an engine is created
 * (mjpc_hip_create / mjpc_hip_multi_create).
 *
 *   name              value            effect
 *   "tier"            "A" | "B"        never / always launch the dense capacity tier (two candidates per CU) first
 *   "dense_tier_cap"  "rows,contacts"  capacity of the dense tier (a tiny one makes most candidates take the checkpointed retry)
 *   "no_model_cache"  any              read the model tables from HBM / L2 instead of the per-workgroup LDS copy
 *   "dense_factor"    any              dense elimination order for every factorisation (ignore the elimination tree)
 *   "fault_inject"    "sync"           one helper wave of candidate 1 stays silent in step 2 (exercises the hand-shake timeout)
 *
 * value == NULL removes the knob. */
#ifndef MJPC_HIP_DEBUG_H_
#define MJPC_HIP_DEBUG_H_
#ifdef __cplusplus
extern "C" {
#endif
void mjpc_hip_debug_set(const char *name, const char *value);
/* Capacity of the engine's dense tier (rows, contacts; 0, 0 without one); *hot = 1 when it is the variant with the hot tables in
   LDS (rollout_dense2h.hip).  Diagnostics / tests only. */
void mjpc_hip_debug_dense_capacity(struct MjpcHipEngine *e, int *nefc, int *ncon, int *hot);
#ifdef __cplusplus
}
#endif
#endif /* MJPC_HIP_DEBUG_H_ */
