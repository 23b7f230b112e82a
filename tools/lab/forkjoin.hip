// tools/lab/forkjoin.hip -- what a fork/join inside a captured graph costs on this machine: 32 steps of
//   serial:    D (27 us, 32 workgroups) -> T (9 us, 200 workgroups) -> F (43 us, 215 workgroups)
//   fork/join: (D -> T) on one branch, F on another, joined before the next step
// with spin kernels of the given durations (1024-thread workgroups holding 147 KB of LDS for D and F, so one per CU).
// Prints us per step for both graphs.   hipcc --offload-arch=gfx950 -O2 -o tools/lab/forkjoin tools/lab/forkjoin.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
template <int LDSB>
__global__ void k_spin(int us, double* sink) {
  __shared__ double buf[LDSB / 8];
  buf[threadIdx.x % (LDSB / 8)] = 1.0;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();            // 100 MHz
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)us * 100ull) __builtin_amdgcn_s_sleep(8);
  if (buf[0] == 123.0) sink[0] = 1.0;
}
int main(int argc, char** argv) {
  const int steps = 32, tD = argc > 1 ? atoi(argv[1]) : 27, tT = argc > 2 ? atoi(argv[2]) : 9, tF = argc > 3 ? atoi(argv[3]) : 43;
  double* sink; CK(hipMalloc(&sink, 64));
  CK(hipFuncSetAttribute((const void*)k_spin<147456>, hipFuncAttributeMaxDynamicSharedMemorySize, 0));
  hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
  hipEvent_t ev[2 * steps + 2]; for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  hipGraph_t g; hipGraphExec_t ge[2];
  for (int mode = 0; mode < 2; ++mode) {
    CK(hipStreamBeginCapture(s1, hipStreamCaptureModeGlobal));
    for (int k = 0; k < steps; ++k) {
      if (mode == 0) {
        hipLaunchKernelGGL(k_spin<147456>, dim3(32), dim3(1024), 0, s1, tD, sink);
        hipLaunchKernelGGL(k_spin<32768>, dim3(200), dim3(512), 0, s1, tT, sink);
        hipLaunchKernelGGL(k_spin<147456>, dim3(215), dim3(1024), 0, s1, tF, sink);
      } else {
        CK(hipEventRecord(ev[2 * k], s1)); CK(hipStreamWaitEvent(s2, ev[2 * k], 0));
        hipLaunchKernelGGL(k_spin<147456>, dim3(215), dim3(1024), 0, s2, tF, sink);
        hipLaunchKernelGGL(k_spin<147456>, dim3(32), dim3(1024), 0, s1, tD, sink);
        hipLaunchKernelGGL(k_spin<32768>, dim3(200), dim3(512), 0, s1, tT, sink);
        CK(hipEventRecord(ev[2 * k + 1], s2)); CK(hipStreamWaitEvent(s1, ev[2 * k + 1], 0));
      }
    }
    CK(hipStreamEndCapture(s1, &g));
    CK(hipGraphInstantiate(&ge[mode], g, nullptr, nullptr, 0));
  }
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int mode = 0; mode < 2; ++mode) {
    CK(hipGraphLaunch(ge[mode], s1)); CK(hipStreamSynchronize(s1));
    CK(hipEventRecord(a, s1));
    for (int r = 0; r < 10; ++r) CK(hipGraphLaunch(ge[mode], s1));
    CK(hipEventRecord(b, s1)); CK(hipStreamSynchronize(s1));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("%s: %.2f us per step (kernels: D %d + T %d%s F %d us)\n", mode ? "fork/join" : "serial   ", ms * 1e3 / 10 / steps, tD, tT, mode ? " beside" : " +", tF);
  }
  return 0;
}
