// microbench2.hip -- clean fp64 MFMA issue-rate probe (inline asm, accumulators pinned).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2);} } while (0)

template <int NACC, bool AGPR>
__global__ __launch_bounds__(256) void k_mfma(double* out, long long* cyc, int iters) {
  v4d acc[NACC];
#pragma unroll
  for (int u = 0; u < NACC; ++u) acc[u] = v4d{0, 0, 0, 0};
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < NACC; ++u) {
      if (AGPR) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(acc[u]) : "v"(a), "v"(b));
      else asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[u]) : "v"(a), "v"(b));
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
#pragma unroll
  for (int u = 0; u < NACC; ++u) s += acc[u][0] + acc[u][1] + acc[u][2] + acc[u][3];
  if (s == 12345.678) out[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = r1 - r0; }
}

template <class F> void run(const char* name, F&& launch, double mfma_per_wave_iter, int blocks, int waves_per_block, int iters) {
  double* out; long long* cyc; HIPCHK(hipMalloc((void**)&out, 4096)); HIPCHK(hipMalloc((void**)&cyc, 64));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(out, cyc, 10);
  HIPCHK(hipDeviceSynchronize());
  hipEventRecord(e0); launch(out, cyc, iters); hipEventRecord(e1); HIPCHK(hipEventSynchronize(e1));
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[2]; HIPCHK(hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost));
  printf("%-28s blocks=%4d x %d waves: %7.3f ms %6.1f TFLOP/s  cycles per MFMA per wave %.1f  clock %.2f GHz\n", name, blocks, waves_per_block, ms,
         mfma_per_wave_iter * 2048.0 * waves_per_block * blocks * iters / (ms * 1e-3) / 1e12, (double)h[0] / iters / mfma_per_wave_iter, h[1] ? (double)h[0] / h[1] * 0.1 : 0.0);
  hipFree(out); hipFree(cyc);
}

int main() {
  const int iters = 4000;
  for (int blocks : {256, 512, 1024}) {
    run("vgpr acc x1", [&](double* o, long long* c, int it) { hipLaunchKernelGGL((k_mfma<1, false>), dim3(blocks), dim3(256), 0, 0, o, c, it); }, 1, blocks, 4, iters);
    run("vgpr acc x2", [&](double* o, long long* c, int it) { hipLaunchKernelGGL((k_mfma<2, false>), dim3(blocks), dim3(256), 0, 0, o, c, it); }, 2, blocks, 4, iters);
    run("vgpr acc x4", [&](double* o, long long* c, int it) { hipLaunchKernelGGL((k_mfma<4, false>), dim3(blocks), dim3(256), 0, 0, o, c, it); }, 4, blocks, 4, iters);
    run("vgpr acc x16", [&](double* o, long long* c, int it) { hipLaunchKernelGGL((k_mfma<16, false>), dim3(blocks), dim3(256), 0, 0, o, c, it); }, 16, blocks, 4, iters);
    run("agpr acc x16", [&](double* o, long long* c, int it) { hipLaunchKernelGGL((k_mfma<16, true>), dim3(blocks), dim3(256), 0, 0, o, c, it); }, 16, blocks, 4, iters);
  }
  run("ONE block vgpr acc x4", [&](double* o, long long* c, int it) { hipLaunchKernelGGL((k_mfma<4, false>), dim3(1), dim3(256), 0, 0, o, c, it); }, 4, 1, 4, iters);
  run("ONE wave vgpr acc x4", [&](double* o, long long* c, int it) { hipLaunchKernelGGL((k_mfma<4, false>), dim3(1), dim3(64), 0, 0, o, c, it); }, 4, 1, 1, iters);
  run("512 thr/block x4 (256 blocks)", [&](double* o, long long* c, int it) { hipLaunchKernelGGL((k_mfma<4, false>), dim3(256), dim3(512), 0, 0, o, c, it); }, 4, 256, 8, iters);
  return 0;
}
