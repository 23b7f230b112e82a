// tools/lab/ratelab.hip -- issue cost of the fp64 VALU instructions the elementwise stages (kernel build, gradient epilogue) are made of.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o ratelab ratelab.hip
// Every wave runs `iters` rounds of 16 independent copies of ONE instruction; 256 CUs x 4 SIMDs x W waves.  Reported: SIMD cycles
// per wave instruction (time x clock / instructions per SIMD), at W = 1 (latency-bound if the instruction's latency exceeds its
// issue cost x 16 chains) and W = 4.  The build kernel's floor counts 27 instructions x 4 cycles per pair and mixture: is every one
// of them a 4-cycle instruction?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2);} } while (0)

template <int OP>
__global__ __launch_bounds__(256) void k_rate(double* out, int iters) {
  double f[16];
  int g[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) { f[u] = 1.0 + u * 1e-3 + threadIdx.x * 1e-9; g[u] = u + (int)threadIdx.x; }
  const double b = 1.0 - 1e-9, c = 1e-9;
  const int e = 1;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(f[u]) : "v"(b), "v"(c));
      if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(f[u]) : "v"(b));
      if (OP == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(f[u]) : "v"(c));
      if (OP == 3) asm volatile("v_max_f64 %0, %0, %1" : "+v"(f[u]) : "v"(c));
      if (OP == 4) asm volatile("v_rndne_f64 %0, %0" : "+v"(f[u]));
      if (OP == 5) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(g[u]) : "v"(f[u]));
      if (OP == 6) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(f[u]) : "v"(e));
      if (OP == 7) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(g[u]) : "v"(e));
      if (OP == 8) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(f[u]) : "v"(g[u]));
      if (OP == 9) asm volatile("v_rcp_f64 %0, %0" : "+v"(f[u]));
      if (OP == 10) asm volatile("v_rsq_f64 %0, %0" : "+v"(f[u]));
      if (OP == 11) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(f[u]) : "v"(b), "v"(c));
      if (OP == 12) asm volatile("v_add_u32 %0, %0, %1" : "+v"(g[u]) : "v"(e));
      if (OP == 13) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(g[u]) : "v"(e), "v"(e));
      if (OP == 14) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(f[u]) : "v"(b));
      if (OP == 15) asm volatile("v_exp_f32 %0, %0" : "+v"(g[u]));
    }
  }
  double s = 0;
#pragma unroll
  for (int u = 0; u < 16; ++u) s += f[u] + g[u];
  if (s == 12345.678) out[0] = s;
}

template <int OP>
static void run(const char* name, double* out, double ghz) {
  hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  for (int W : {1, 2, 4}) {
    const int iters = 20000 / W, blocks = 256 * W;
    hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, out, 100);
    HIPCHK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, out, iters);
    HIPCHK(hipEventRecord(e1, 0));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    const double per_simd = (double)iters * 16 * W;                 // wave instructions issued on one SIMD
    printf("%-16s W=%d  %6.2f cycles per wave instruction (at %.2f GHz)   %.3f ms\n", name, W, ms * 1e-3 * ghz * 1e9 / per_simd, ghz, ms);
  }
}

int main(int argc, char** argv) {
  const double ghz = argc > 1 ? atof(argv[1]) : 2.4;
  double* out; HIPCHK(hipMalloc((void**)&out, 64));
  run<0>("v_fma_f64", out, ghz); run<11>("v_fmac_f64", out, ghz); run<1>("v_mul_f64", out, ghz); run<2>("v_add_f64", out, ghz);
  run<3>("v_max_f64", out, ghz); run<4>("v_rndne_f64", out, ghz); run<5>("v_cvt_i32_f64", out, ghz); run<6>("v_ldexp_f64", out, ghz);
  run<8>("v_cvt_f64_i32", out, ghz); run<7>("v_lshl_add_u32", out, ghz); run<12>("v_add_u32", out, ghz); run<13>("v_fma_f32", out, ghz);
  run<14>("v_pk_fma_f32", out, ghz); run<15>("v_exp_f32", out, ghz); run<9>("v_rcp_f64", out, ghz); run<10>("v_rsq_f64", out, ghz);
  return 0;
}
