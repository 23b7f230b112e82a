// tools/lab/pipelab.hip -- do fp64 VALU work and fp64 MFMA work share an execution pipe on gfx950?
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o pipelab pipelab.hip
// Per wave and iteration: NM back-to-back independent v_mfma_f64_16x16x4_f64 and NV independent v_fma_f64.  Modes: MFMA only,
// VALU only, both in the same wave, and split (on every SIMD one wave of MFMA only and one of VALU only).  If the time of
// "both" is max(MFMA, VALU) the pipes are separate and an fp64 elementwise epilogue can hide behind another workgroup's
// multiply loop; if it is the sum, every fp64 VALU instruction is taken from the matrix pipe's time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double v4d __attribute__((ext_vector_type(4)));
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2);} } while (0)

template <int NM, int NV, int MODE>   // MODE 0: every wave does both; 1: waves 0-3 MFMA, waves 4-7 VALU (waves w and w+4 share SIMD w)
__global__ __launch_bounds__(512) void k_pipe(double* out, int iters) {
  v4d acc[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) acc[u] = v4d{0, 0, 0, 0};
  double f[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) f[u] = 1.0 + u * 1e-3 + threadIdx.x * 1e-9;
  const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9, c = 1e-9;
  const int wave = threadIdx.x >> 6;
  const bool do_m = MODE == 0 || wave < 4, do_v = MODE == 0 || wave >= 4;
  for (int it = 0; it < iters; ++it) {
    if (do_m) {
#pragma unroll
      for (int u = 0; u < NM; ++u) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[u & 7]) : "v"(a), "v"(b));
    }
    if (do_v) {
#pragma unroll
      for (int u = 0; u < NV; ++u) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(f[u & 15]) : "v"(b), "v"(c));
    }
  }
  double s = 0;
#pragma unroll
  for (int u = 0; u < 8; ++u) s += acc[u][0] + acc[u][3];
#pragma unroll
  for (int u = 0; u < 16; ++u) s += f[u];
  if (s == 12345.678) out[0] = s;
}

// the same with the two kinds interleaved instruction by instruction (one MFMA, then NV / NM FMAs)
template <int NM, int NVPER>
__global__ __launch_bounds__(512) void k_mix(double* out, int iters) {
  v4d acc[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) acc[u] = v4d{0, 0, 0, 0};
  double f[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) f[u] = 1.0 + u * 1e-3 + threadIdx.x * 1e-9;
  const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9, c = 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < NM; ++u) {
      asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[u & 7]) : "v"(a), "v"(b));
#pragma unroll
      for (int w = 0; w < NVPER; ++w) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(f[(u * NVPER + w) & 15]) : "v"(b), "v"(c));
    }
  }
  double s = 0;
#pragma unroll
  for (int u = 0; u < 8; ++u) s += acc[u][0] + acc[u][3];
#pragma unroll
  for (int u = 0; u < 16; ++u) s += f[u];
  if (s == 12345.678) out[0] = s;
}

template <class L> double timeit(L&& launch, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(10); HIPCHK(hipDeviceSynchronize());
  hipEventRecord(e0); launch(iters); hipEventRecord(e1); HIPCHK(hipEventSynchronize(e1));
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0); hipEventDestroy(e1);
  return ms * 1e3;
}

int main() {
  double* out; HIPCHK(hipMalloc((void**)&out, 4096));
  const int iters = 2000, blocks = 256;
  // 8 waves per workgroup, one workgroup per CU: two waves per SIMD
  const double m_only = timeit([&](int it) { hipLaunchKernelGGL((k_pipe<8, 0, 0>), dim3(blocks), dim3(512), 0, 0, out, it); }, iters);
  const double v16 = timeit([&](int it) { hipLaunchKernelGGL((k_pipe<0, 128, 0>), dim3(blocks), dim3(512), 0, 0, out, it); }, iters);
  printf("per iteration and wave: 8 MFMA f64 16x16x4 (nominal 8 x 64 = 512 cycles), 128 v_fma_f64 (nominal 128 x 4 = 512 cycles); 2 waves per SIMD\n");
  printf("MFMA only                          %9.1f us\n", m_only);
  printf("VALU fp64 only                     %9.1f us\n", v16);
  printf("both, same wave, block by block    %9.1f us   (sum %.1f, max %.1f)\n",
         timeit([&](int it) { hipLaunchKernelGGL((k_pipe<8, 128, 0>), dim3(blocks), dim3(512), 0, 0, out, it); }, iters), m_only + v16, m_only > v16 ? m_only : v16);
  printf("both, same wave, interleaved 1:16  %9.1f us\n", timeit([&](int it) { hipLaunchKernelGGL((k_mix<8, 16>), dim3(blocks), dim3(512), 0, 0, out, it); }, iters));
  const double m_half = timeit([&](int it) { hipLaunchKernelGGL((k_pipe<8, 0, 1>), dim3(blocks), dim3(512), 0, 0, out, it); }, iters);
  const double v_half = timeit([&](int it) { hipLaunchKernelGGL((k_pipe<0, 128, 1>), dim3(blocks), dim3(512), 0, 0, out, it); }, iters);
  printf("waves 0-3 MFMA only (4-7 idle)     %9.1f us\n", m_half);
  printf("waves 4-7 VALU only (0-3 idle)     %9.1f us\n", v_half);
  printf("waves 0-3 MFMA, 4-7 VALU, per SIMD %9.1f us   (sum %.1f, max %.1f)\n",
         timeit([&](int it) { hipLaunchKernelGGL((k_pipe<8, 128, 1>), dim3(blocks), dim3(512), 0, 0, out, it); }, iters), m_half + v_half, m_half > v_half ? m_half : v_half);
  // the same question for the transcendental-free integer / fp32 VALU is not asked: the epilogues are fp64
  hipFree(out);
  return 0;
}
