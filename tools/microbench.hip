// microbench.hip -- gfx950 fp64 issue-rate probes used to size the kernels (not product code).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2);} } while (0)

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(double* out, long long* cyc, int iters) {
  v4d acc[NACC];
#pragma unroll
  for (int u = 0; u < NACC; ++u) acc[u] = v4d{0, 0, 0, 0};
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < NACC; ++u) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[u], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int u = 0; u < NACC; ++u) s += acc[u][0] + acc[u][1] + acc[u][2] + acc[u][3];
  long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (s == 12345.678) out[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = r1 - r0; }
}

template <int NACC>
__global__ __launch_bounds__(256) void k_dfma(double* out, long long* cyc, int iters) {
  double acc[NACC];
#pragma unroll
  for (int u = 0; u < NACC; ++u) acc[u] = threadIdx.x * 1e-3 + u;
  double a = 1.0 + threadIdx.x * 1e-9, b = 1e-9;
  long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < NACC; ++u) acc[u] = __builtin_fma(acc[u], a, b);
  }
  double s = 0;
#pragma unroll
  for (int u = 0; u < NACC; ++u) s += acc[u];
  long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (s == 12345.678) out[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = r1 - r0; }
}

// dependent chain latencies
__global__ void k_lat(double* out, long long* cyc, int iters, int which) {
  double x = 1.5 + threadIdx.x * 1e-6;
  long long t0 = __builtin_amdgcn_s_memtime();
  if (which == 0) for (int it = 0; it < iters; ++it) x = __builtin_fma(x, 1.0000001, 1e-9);
  if (which == 1) for (int it = 0; it < iters; ++it) x = 1.0 / x + 0.5;
  if (which == 2) for (int it = 0; it < iters; ++it) x = sqrt(x) + 0.5;
  if (which == 3) for (int it = 0; it < iters; ++it) x = __builtin_amdgcn_rcp(x) + 0.5;
  if (which == 4) for (int it = 0; it < iters; ++it) x = __builtin_amdgcn_rsq(x) + 0.5;
  if (which == 5) for (int it = 0; it < iters; ++it) x = exp(-x) + 0.5;
  if (which == 6) for (int it = 0; it < iters; ++it) x = __shfl_xor(x, 1, 64) + 0.5;
  if (which == 7) for (int it = 0; it < iters; ++it) { v4d c = {x, x, x, x}; c = __builtin_amdgcn_mfma_f64_16x16x4f64(x, 1e-3, c, 0, 0, 0); x = c[0]; }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <class F> void run(const char* name, F&& launch, double flop_per_block_iter, int blocks, int iters) {
  double* out; long long* cyc; HIPCHK(hipMalloc((void**)&out, 4096)); HIPCHK(hipMalloc((void**)&cyc, 64));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(out, cyc, 10);
  HIPCHK(hipDeviceSynchronize());
  hipEventRecord(e0); launch(out, cyc, iters); hipEventRecord(e1); HIPCHK(hipEventSynchronize(e1));
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[2]; HIPCHK(hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost));
  printf("%-34s blocks=%4d: %.3f ms  %.1f TFLOP/s   shader cycles/iter %.1f  clock %.2f GHz\n", name, blocks, ms,
         flop_per_block_iter * blocks * iters / (ms * 1e-3) / 1e12, (double)h[0] / iters, h[1] ? (double)h[0] / h[1] * 0.1 : 0.0);
  hipFree(out); hipFree(cyc);
}

int main() {
  const int iters = 20000;
  for (int blocks : {256, 512, 1024}) {
    run("mfma f64 16x16x4, 1 acc/wave", [&](double* o, long long* c, int it) { hipLaunchKernelGGL(k_mfma<1>, dim3(blocks), dim3(256), 0, 0, o, c, it); }, 4 * 1 * 2048.0, blocks, iters);
    run("mfma f64 16x16x4, 2 acc/wave", [&](double* o, long long* c, int it) { hipLaunchKernelGGL(k_mfma<2>, dim3(blocks), dim3(256), 0, 0, o, c, it); }, 4 * 2 * 2048.0, blocks, iters);
    run("mfma f64 16x16x4, 4 acc/wave", [&](double* o, long long* c, int it) { hipLaunchKernelGGL(k_mfma<4>, dim3(blocks), dim3(256), 0, 0, o, c, it); }, 4 * 4 * 2048.0, blocks, iters);
    run("mfma f64 16x16x4, 16 acc/wave", [&](double* o, long long* c, int it) { hipLaunchKernelGGL(k_mfma<16>, dim3(blocks), dim3(256), 0, 0, o, c, it); }, 4 * 16 * 2048.0, blocks, iters / 4);
    run("v_fma_f64, 8 chains/lane", [&](double* o, long long* c, int it) { hipLaunchKernelGGL(k_dfma<8>, dim3(blocks), dim3(256), 0, 0, o, c, it); }, 256 * 8 * 2.0, blocks, iters);
  }
  run("mfma f64, ONE block, 4 acc", [&](double* o, long long* c, int it) { hipLaunchKernelGGL(k_mfma<4>, dim3(1), dim3(256), 0, 0, o, c, it); }, 4 * 4 * 2048.0, 1, iters);
  run("v_fma_f64, ONE block, 8 chains", [&](double* o, long long* c, int it) { hipLaunchKernelGGL(k_dfma<8>, dim3(1), dim3(256), 0, 0, o, c, it); }, 256 * 8 * 2.0, 1, iters);
  const char* names[] = {"fma chain", "1.0/x (IEEE div)", "sqrt", "v_rcp_f64", "v_rsq_f64", "exp(-x)", "shfl_xor", "mfma dependent"};
  for (int w = 0; w < 8; ++w) {
    double* out; long long* cyc; HIPCHK(hipMalloc((void**)&out, 4096)); HIPCHK(hipMalloc((void**)&cyc, 64));
    hipLaunchKernelGGL(k_lat, dim3(1), dim3(64), 0, 0, out, cyc, 2000, w);
    HIPCHK(hipDeviceSynchronize());
    long long h; HIPCHK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    printf("latency %-20s %.1f cycles per dependent op (one wave)\n", names[w], (double)h / 2000);
    hipFree(out); hipFree(cyc);
  }
  return 0;
}
