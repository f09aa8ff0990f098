// tools/mfma_hessian.hip — micro-benchmark that settles "MFMA or VALU for the Newton Hessian build" with numbers.
//
// The solver builds  H = M + JH^T JH  (solver.h: newton_entries) from the scaled constraint rows JH[R][nvp] that sit in LDS
// (fp64, odd row stride).  Three implementations over the SAME LDS layout, one wavefront per CU like the owner wave of the
// rollout kernel, every CU busy:
//   valu_pattern  the shipped form: one lane per non-zero of M's sparsity pattern (humanoid, nv = 27: 159 of 378 lower-
//                 triangle entries + 27 gradient entries), 3 entries per lane, 8 rows per trip (16 LDS reads in flight)
//   valu_dense    the same loop over the whole lower triangle (what cross-branch contacts need)
//   mfma          v_mfma_f64_16x16x4_f64 on the three 16x16 blocks of the lower triangle of the 32x32 padded product
// Output: cycles (s_memtime ticks, 100 MHz -> converted with the measured shader clock ratio) per build, per row count.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_hessian tools/mfma_hessian.hip && /tmp/mfma_hessian
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

#define NV 27
#define NVP 29
#define LANE ((int)(threadIdx.x & 63))
typedef double double4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ long long ticks() { return (long long)__builtin_amdgcn_s_memtime(); }

// entries e -> (i, j) packed i | j << 8 (j == NV: gradient column)
template <int G>
__device__ __forceinline__ void valu_entries(const double *JH, const int *pairs, int nent, int rows, double *H) {
  for (int e0 = LANE; e0 < nent; e0 += G * 64) {
    int ii[G], jj[G];
#pragma unroll
    for (int g = 0; g < G; g++) { int e = e0 + g * 64; int pk = e < nent ? pairs[e] : 0; ii[g] = pk & 255; jj[g] = pk >> 8; }
    double hp[G], hq[G];
#pragma unroll
    for (int g = 0; g < G; g++) { hp[g] = 0; hq[g] = 0; }
    for (int a = 0; a < rows; a += 8) {
      double x[G][8], y[G][8];
#pragma unroll
      for (int g = 0; g < G; g++)
#pragma unroll
        for (int k = 0; k < 8; k++) { x[g][k] = JH[(a + k) * NVP + ii[g]]; y[g][k] = JH[(a + k) * NVP + jj[g]]; }
#pragma unroll
      for (int g = 0; g < G; g++)
#pragma unroll
        for (int k = 0; k < 8; k += 2) { hp[g] += x[g][k] * y[g][k]; hq[g] += x[g][k + 1] * y[g][k + 1]; }
    }
#pragma unroll
    for (int g = 0; g < G; g++) { int e = e0 + g * 64; if (e < nent) H[ii[g] * 32 + jj[g]] = hp[g] + hq[g]; }
  }
}

// blocks (0,0), (1,0), (1,1) of the 32 x 32 padded product; D layout of v_mfma_f64_16x16x4_f64 (probed on gfx950): lane l, v -> D[4*v + l/16][l%16]
__device__ __forceinline__ void mfma_build(const double *JH, int rows, double *H) {
  const int li = LANE & 15, lk = LANE >> 4;
  double4_t c00 = {0, 0, 0, 0}, c10 = {0, 0, 0, 0}, c11 = {0, 0, 0, 0};
  for (int r = 0; r < rows; r += 4) {
    double a0 = JH[(r + lk) * NVP + li];            // columns 0..15 of row r + lk   (A[i][k] and B[k][j] coincide for JH^T JH)
    double a1 = JH[(r + lk) * NVP + 16 + li];       // columns 16..31 (27.. are padding / the next row: only padded entries see it)
    c00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a0, c00, 0, 0, 0);
    c10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, a0, c10, 0, 0, 0);
    c11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, a1, c11, 0, 0, 0);
  }
#pragma unroll
  for (int v = 0; v < 4; v++) {
    int i = 4 * v + lk;
    H[i * 32 + li] = c00[v];
    H[(16 + i) * 32 + li] = c10[v];
    H[(16 + i) * 32 + 16 + li] = c11[v];
  }
}

// the same with two accumulators per block and the loads of both row groups issued before the six MFMAs (breaks the dependent
// accumulate chains, keeps four LDS reads in flight)
__device__ __forceinline__ void mfma_build2(const double *JH, int rows, double *H) {
  const int li = LANE & 15, lk = LANE >> 4;
  double4_t c00 = {0, 0, 0, 0}, c10 = {0, 0, 0, 0}, c11 = {0, 0, 0, 0}, d00 = {0, 0, 0, 0}, d10 = {0, 0, 0, 0}, d11 = {0, 0, 0, 0};
  for (int r = 0; r < rows; r += 8) {
    double a0 = JH[(r + lk) * NVP + li], a1 = JH[(r + lk) * NVP + 16 + li];
    double b0 = JH[(r + 4 + lk) * NVP + li], b1 = JH[(r + 4 + lk) * NVP + 16 + li];
    c00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a0, c00, 0, 0, 0);
    d00 = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, b0, d00, 0, 0, 0);
    c10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, a0, c10, 0, 0, 0);
    d10 = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, b0, d10, 0, 0, 0);
    c11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, a1, c11, 0, 0, 0);
    d11 = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, b1, d11, 0, 0, 0);
  }
#pragma unroll
  for (int v = 0; v < 4; v++) {
    int i = 4 * v + lk;
    H[i * 32 + li] = c00[v] + d00[v];
    H[(16 + i) * 32 + li] = c10[v] + d10[v];
    H[(16 + i) * 32 + 16 + li] = c11[v] + d11[v];
  }
}
// one block only (what each of the three waves of a candidate would do when the blocks are shared out)
__device__ __forceinline__ void mfma_block(const double *JH, int rows, double *H) {
  const int li = LANE & 15, lk = LANE >> 4;
  double4_t c = {0, 0, 0, 0}, d = {0, 0, 0, 0};
  for (int r = 0; r < rows; r += 8) {
    double a0 = JH[(r + lk) * NVP + li], a1 = JH[(r + lk) * NVP + 16 + li];
    double b0 = JH[(r + 4 + lk) * NVP + li], b1 = JH[(r + 4 + lk) * NVP + 16 + li];
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, a0, c, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, b0, d, 0, 0, 0);
  }
#pragma unroll
  for (int v = 0; v < 4; v++) H[(16 + 4 * v + lk) * 32 + li] = c[v] + d[v];
}

__global__ void __launch_bounds__(64) bench(const double *JHg, const int *pat, int npat, const int *dense, int ndense, int rows, int reps,
                                            double *Hout, long long *cyc) {
  __shared__ double JH[(128 + 4) * NVP + 32];
  __shared__ double H[3][32 * 32];
  __shared__ int spat[512], sdense[512];
  for (int e = LANE; e < (128 + 4) * NVP + 32; e += 64) JH[e] = e < rows * NVP ? JHg[e] : 0.0;
  for (int e = LANE; e < npat; e += 64) spat[e] = pat[e];
  for (int e = LANE; e < ndense; e += 64) sdense[e] = dense[e];
  for (int e = LANE; e < 3 * 1024; e += 64) (&H[0][0])[e] = 0;
  __syncthreads();
  long long t[7];
  t[0] = ticks();
  for (int r = 0; r < reps; r++) { valu_entries<3>(JH, spat, npat, rows, H[0]); __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }
  t[1] = ticks();
  for (int r = 0; r < reps; r++) { valu_entries<3>(JH, sdense, ndense, rows, H[1]); __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }
  t[2] = ticks();
  for (int r = 0; r < reps; r++) { mfma_build(JH, rows, H[2]); __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }
  t[3] = ticks();
  for (int r = 0; r < reps; r++) { mfma_build2(JH, rows, H[2]); __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }
  t[4] = ticks();
  for (int r = 0; r < reps; r++) { mfma_block(JH, rows, H[0]); __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }
  t[5] = ticks();
  for (int r = 0; r < reps; r++) { valu_entries<2>(JH, spat, (npat + 2) / 3, rows, H[0]); __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }   // a third of the pattern (one of three waves)
  t[6] = ticks();
  __syncthreads();
  if (blockIdx.x == 0) {
    for (int e = LANE; e < 3 * 1024; e += 64) Hout[e] = (&H[0][0])[e];
    if (LANE == 0) for (int k = 0; k < 6; k++) cyc[k] = (t[k + 1] - t[k]) / reps;
  }
}

// shader-clock ticks per s_memtime tick, measured with a dependent v_fma_f64 chain (5.6 cycles per link, DESIGN.md calibration
// is not assumed: the ratio comes from clock64 vs memtime below)
__global__ void clocks(long long *out) {
  long long m0 = ticks(), c0 = clock64();
  double x = 1.0;
  for (int i = 0; i < 100000; i++) x = __builtin_fma(x, 1.0000001, 1e-9);
  long long m1 = ticks(), c1 = clock64();
  if (threadIdx.x == 0) { out[0] = m1 - m0; out[1] = c1 - c0; out[2] = (long long)x; }
}

int main() {
  const int parent[27] = {-1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 8, 15, 16, 17, 18, 19, 5, 21, 22, 5, 24, 25};     // DofTree<27>
  std::vector<int> pat, dense;
  for (int i = 0; i < NV; i++) for (int j = i; j >= 0; j = parent[j]) pat.push_back(i | (j << 8));
  for (int i = 0; i < NV; i++) pat.push_back(i | (NV << 8));
  for (int i = 0; i < NV; i++) for (int j = 0; j <= i; j++) dense.push_back(i | (j << 8));
  for (int i = 0; i < NV; i++) dense.push_back(i | (NV << 8));
  std::vector<double> JH((128 + 4) * NVP + 32);
  srand(1);
  for (auto &v : JH) v = rand() / (double)RAND_MAX - 0.5;
  double *dJ, *dH; int *dp, *dd; long long *dc;
  hipMalloc(&dJ, JH.size() * 8); hipMalloc(&dH, 3 * 1024 * 8); hipMalloc(&dp, pat.size() * 4); hipMalloc(&dd, dense.size() * 4); hipMalloc(&dc, 128);
  hipMemcpy(dJ, JH.data(), JH.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(dp, pat.data(), pat.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dd, dense.data(), dense.size() * 4, hipMemcpyHostToDevice);
  long long hc[8];
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(clocks, dim3(1), dim3(64), 0, 0, dc);       // warm-up
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(clocks, dim3(1), dim3(64), 0, 0, dc);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(hc, dc, 24, hipMemcpyDeviceToHost);
  double ns_per_tick = 1e6 * ms / (double)hc[0];                   // s_memtime tick in nanoseconds (kernel = 100000 dependent v_fma_f64)
  double ratio = 1.0;
  printf("{\"nv\": %d, \"pattern_entries\": %zu, \"dense_entries\": %zu, \"memtime_ticks_per_100k_dependent_fma\": %lld, \"ns_per_memtime_tick\": %.3f, \"unit\": \"s_memtime ticks\", \"results\": [\n",
         NV, pat.size(), dense.size(), hc[0], ns_per_tick);
  const int rowsv[4] = {32, 64, 96, 128};
  for (int q = 0; q < 4; q++) {
    int rows = rowsv[q];
    hipLaunchKernelGGL(bench, dim3(256), dim3(64), 0, 0, dJ, dp, (int)pat.size(), dd, (int)dense.size(), rows, 200, dH, dc);
    hipDeviceSynchronize();
    std::vector<double> H(3 * 1024);
    hipMemcpy(H.data(), dH, 3 * 1024 * 8, hipMemcpyDeviceToHost); hipMemcpy(hc, dc, 48, hipMemcpyDeviceToHost);
    // reference on the host
    double errp = 0, errd = 0, errm = 0;
    for (int i = 0; i < NV; i++) for (int j = 0; j <= i; j++) {
      double s = 0;
      for (int r = 0; r < rows; r++) s += JH[r * NVP + i] * JH[r * NVP + j];
      bool inpat = false;
      for (int a = i; a >= 0; a = parent[a]) if (a == j) inpat = true;
      if (inpat) errp = fmax(errp, fabs(H[i * 32 + j] - s));
      errd = fmax(errd, fabs(H[1024 + i * 32 + j] - s));
      errm = fmax(errm, fabs(H[2048 + i * 32 + j] - s));
    }
    printf("  {\"rows\": %d, \"valu_pattern_cycles\": %.0f, \"valu_dense_cycles\": %.0f, \"mfma_cycles\": %.0f, \"mfma_2acc_cycles\": %.0f, \"mfma_one_block_of_three_cycles\": %.0f, \"valu_pattern_third_cycles\": %.0f, \"max_abs_err\": [%.2e, %.2e, %.2e]}%s\n",
           rows, hc[0] * ratio, hc[1] * ratio, hc[2] * ratio, hc[3] * ratio, hc[4] * ratio, hc[5] * ratio, errp, errd, errm, q < 3 ? "," : "");
  }
  printf("]}\n");
  return 0;
}
