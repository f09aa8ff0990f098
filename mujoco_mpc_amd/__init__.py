"""mujoco_mpc_amd — MI355X-native Predictive-Sampling rollout engine for MJPC (hot path only).

Product code: `csrc/` (HIP kernels + C ABI, include/mjpc_hip.h), `capi.py` (ctypes binding),
`planner.py` (ctypes wrappers of the engine ABI), `cplanner.py` (ctypes view of the C++ host planner), `modelgen/` (model authoring without MuJoCo).
The CPU oracle lives in /oracle and is test infrastructure only.
"""
__version__ = "0.1.0"
