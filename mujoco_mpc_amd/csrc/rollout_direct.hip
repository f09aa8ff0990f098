// rollout_direct.hip — rollout kernels that read the model tables straight from HBM / L2 (no LDS copy): for models whose
// per-candidate state would not fit 160 KiB of LDS next to a table copy (the 33-dof hand).
#define MJPC_TU direct
#define MJPC_NO_MODEL_CACHE 1
#define MJPC_HOT_CACHE 1        // ... except the hot prefix (kinematic / tree tables, ~10 KB), which fits next to the hand's state
#define MJPC_TU_NVT_LIST(X) X(33)
#include "rollout_tu.inc"
