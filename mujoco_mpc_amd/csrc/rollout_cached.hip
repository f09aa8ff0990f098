// rollout_cached.hip — rollout kernels that read the model tables from the workgroup's LDS copy (one candidate per CU).
#define MJPC_TU cached
#define MJPC_TU_NVT_LIST(X) X(2) X(18) X(27)
#include "rollout_tu.inc"
