// rollout_dense2h.hip — the two-candidates-per-CU flavour WITH the hot prefix of the model tables (kinematic / tree tables) in
// LDS: for models whose lean layout leaves room for it without giving up too much contact / constraint-row capacity (the
// 18-dof A1: 112 -> 104 rows at 80 KiB, its workload peaks at 66).  Measured on the A1: 1.9 % faster than rollout_dense2 at 512
// candidates per GPU (the tables' L2 round trips sit on the owner wave's critical path in kinematics and the sweeps).
#define MJPC_TU dense2h
#define MJPC_NO_MODEL_CACHE 1
#define MJPC_HOT_CACHE 1
#define MJPC_MIN_BLOCKS 2
#define MJPC_LEAN_LDS 1
#define MJPC_TU_NVT_LIST(X) X(18)
#include "rollout_tu.inc"
