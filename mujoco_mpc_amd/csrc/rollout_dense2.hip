// rollout_dense2.hip — the two-candidates-per-CU flavour: model tables from HBM / L2 (no LDS copy) and at most 256 registers per
// wave, so that two workgroups (2 x <= 80 KiB of LDS) are resident on every CU and hide each other's stalls.  The engine
// uses it when a shard has more candidates than the GPU has CUs, with a reduced contact / constraint-row capacity; the rare
// candidate that overflows it is re-run by the full-capacity flavour (engine.hip: capacity tiers).
#define MJPC_TU dense2
#define MJPC_NO_MODEL_CACHE 1
#define MJPC_MIN_BLOCKS 2
#define MJPC_LEAN_LDS 1
#define MJPC_TU_NVT_LIST(X) X(18) X(27) X(33)
#include "rollout_tu.inc"
