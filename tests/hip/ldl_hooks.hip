// tests/hip/ldl_hooks.hip — TEST / MEASUREMENT HOOKS, not part of the product library.
// Built into tests/hip/libmjpc_hip_testhooks.so by __graft_entry__.build_test_hooks(); exercises the register L^T D L
// factorisations of mujoco_mpc_amd/csrc/linalg.h in isolation (tests/test_gpu_parity.py::test_register_ldl_*,
// tools/ldl_bench.py).
#include <hip/hip_runtime.h>
#include <string>
#include "../../mujoco_mpc_amd/csrc/core.h"

static thread_local std::string g_error;
static void set_error(const std::string &s) { g_error = s; }
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_error(std::string(#x) + ": " + hipGetErrorString(e_)); return -2; } } while (0)

// test hook for the register factorisations (tests/test_gpu_parity.py): one wave, A (n x n, symmetric) staged in LDS with the
// odd row stride; out[0..n) = fused factor+solve, out[n..2n) = split factor / solve through LDS
template <int N>
__global__ void __launch_bounds__(64) ldl_test_kernel(const double *A, const double *b, double *out, int tree) {
  constexpr int nvp = NVP_OF(N);
  __shared__ double sA[N * nvp], sB[N * nvp], sx[N], sy[N], sD[N];
  for (int e = LANE; e < N * N; e += 64) { int i = e / N, j = e % N; sA[i * nvp + j] = A[e]; sB[i * nvp + j] = A[e]; }
  if (LANE < N) { sx[LANE] = b[LANE]; sy[LANE] = b[LANE]; }
  __syncthreads();
  chol_factor_solve_reg<N>(sA, sx, nvp, tree);
  chol_factor_reg<N>(sB, sD, nvp, tree);
  chol_solve_reg<N>(sB, sD, sy, nvp, tree);
  __syncthreads();
  if (LANE < N) { out[LANE] = sx[LANE]; out[N + LANE] = sy[LANE]; }
}

// micro-benchmark hook (scratch measurements only): wave 0 repeats the fused factor+solve `reps` times, the other three waves
// optionally busy-poll an LDS flag like the solver helpers do; out[0] = s_memtime ticks per repetition
template <int N>
__global__ void __launch_bounds__(256) ldl_bench_kernel(const double *A, const double *b, double *out, int tree, int reps, int spin) {
  constexpr int nvp = NVP_OF(N);
  __shared__ double sA[N * nvp], sx[N];
  __shared__ int flag;
  const int wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) flag = 0;
  for (int e = threadIdx.x; e < N * N; e += 256) { int i = e / N, j = e % N; sA[i * nvp + j] = A[e]; }
  __syncthreads();
  if (wave == 0) {
    long long t0 = (long long)__builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; r++) {
      if (LANE < N) sx[LANE] = b[LANE] + r;
      chol_factor_solve_reg<N>(sA, sx, nvp, tree);
    }
    long long t1 = (long long)__builtin_amdgcn_s_memtime();
    if (LANE == 0) { out[0] = (double)(t1 - t0) / reps; out[1] = sx[0]; }
    flag_set(&flag, 1);
  } else if (spin) {
    flag_wait(&flag, 1);
  }
}


extern "C" {
const char *mjpc_hip_testhooks_last_error(void) { return g_error.c_str(); }

// test hook: x = A^-1 b with the register L^T D L of the rollout kernel (n = 18 or 27; tree = 1: level-ordered sparse form)
int mjpc_hip_debug_ldl(int n, int tree, const double *A, const double *b, double *out, int device) {
  if (n != 18 && n != 27 && n != 33) { set_error("mjpc_hip_debug_ldl: n must be 18, 27 or 33"); return -1; }
  HIPCHK(hipSetDevice(device));
  double *dA = nullptr, *db = nullptr, *dout = nullptr;
  HIPCHK(hipMalloc(&dA, sizeof(double) * n * n)); HIPCHK(hipMalloc(&db, sizeof(double) * n)); HIPCHK(hipMalloc(&dout, sizeof(double) * 2 * n));
  HIPCHK(hipMemcpy(dA, A, sizeof(double) * n * n, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(db, b, sizeof(double) * n, hipMemcpyHostToDevice));
  if (n == 18) hipLaunchKernelGGL(ldl_test_kernel<18>, dim3(1), dim3(64), 0, 0, dA, db, dout, tree);
  else if (n == 27) hipLaunchKernelGGL(ldl_test_kernel<27>, dim3(1), dim3(64), 0, 0, dA, db, dout, tree);
  else hipLaunchKernelGGL(ldl_test_kernel<33>, dim3(1), dim3(64), 0, 0, dA, db, dout, tree);
  HIPCHK(hipGetLastError());
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(out, dout, sizeof(double) * 2 * n, hipMemcpyDeviceToHost));
  hipFree(dA); hipFree(db); hipFree(dout);
  return 0;
}


int mjpc_hip_debug_ldl_bench(int n, int tree, int reps, int spin, const double *A, const double *b, double *out, int device) {
  if (n != 18 && n != 27) return -1;
  HIPCHK(hipSetDevice(device));
  double *dA = nullptr, *db = nullptr, *dout = nullptr;
  HIPCHK(hipMalloc(&dA, sizeof(double) * n * n)); HIPCHK(hipMalloc(&db, sizeof(double) * n)); HIPCHK(hipMalloc(&dout, sizeof(double) * 2));
  HIPCHK(hipMemcpy(dA, A, sizeof(double) * n * n, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(db, b, sizeof(double) * n, hipMemcpyHostToDevice));
  for (int w = 0; w < 2; w++) {
    if (n == 18) hipLaunchKernelGGL(ldl_bench_kernel<18>, dim3(256), dim3(256), 0, 0, dA, db, dout, tree, reps, spin);
    else hipLaunchKernelGGL(ldl_bench_kernel<27>, dim3(256), dim3(256), 0, 0, dA, db, dout, tree, reps, spin);
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(out, dout, sizeof(double) * 2, hipMemcpyDeviceToHost));
  hipFree(dA); hipFree(db); hipFree(dout);
  return 0;
}


}  // extern "C"
