// tools/lab/eventnode.hip -- can the host wait for an event that is recorded by a node INSIDE a captured graph?
// Graph: kernel A (50 us) -> event record -> kernel B (500 us).  After hipGraphLaunch the host calls hipEventSynchronize on
// that event and prints how long it waited, then how long until the stream is idle.  If event nodes work the first wait is
// ~50 us and the second ~500 us more.  (Measured: the capture drops the record -- 2 nodes, no wait; an explicit
// hipGraphAddEventRecordNode between two child graphs works but adds ~10 us, what two graph launches cost anyway.)   hipcc --offload-arch=gfx950 -O2 -o tools/lab/eventnode tools/lab/eventnode.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void k_spin(int us, double* sink) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)us * 100ull) __builtin_amdgcn_s_sleep(8);
  if (us < 0) sink[0] = 1.0;
}
int main() {
  double* sink; CK(hipMalloc(&sink, 64));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  hipLaunchKernelGGL(k_spin, dim3(8), dim3(256), 0, s, 50, sink);
  hipError_t rc = hipEventRecord(ev, s);
  printf("hipEventRecord inside capture: %s\n", hipGetErrorString(rc));
  hipLaunchKernelGGL(k_spin, dim3(8), dim3(256), 0, s, 500, sink);
  CK(hipStreamEndCapture(s, &g));
  size_t nn = 0; CK(hipGraphGetNodes(g, nullptr, &nn)); printf("graph nodes: %zu\n", nn);
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int rep = 0; rep < 4; ++rep) {
    auto t0 = std::chrono::steady_clock::now();
    CK(hipGraphLaunch(ge, s));
    auto t1 = std::chrono::steady_clock::now();
    hipError_t r2 = hipEventSynchronize(ev);
    auto t2 = std::chrono::steady_clock::now();
    CK(hipStreamSynchronize(s));
    auto t3 = std::chrono::steady_clock::now();
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    printf("rep %d: launch %.0f us, wait for the event %.0f us (%s), then stream idle after %.0f us more\n", rep, us(t0, t1), us(t1, t2), hipGetErrorString(r2), us(t2, t3));
  }
  // second attempt: two captured graphs as child nodes of a parent graph with an explicit event-record node between them
  hipGraph_t g1, g2, gp; hipGraphExec_t gpe;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  hipLaunchKernelGGL(k_spin, dim3(8), dim3(256), 0, s, 50, sink);
  CK(hipStreamEndCapture(s, &g1));
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  hipLaunchKernelGGL(k_spin, dim3(8), dim3(256), 0, s, 500, sink);
  CK(hipStreamEndCapture(s, &g2));
  CK(hipGraphCreate(&gp, 0));
  hipGraphNode_t n1, ne, n2;
  CK(hipGraphAddChildGraphNode(&n1, gp, nullptr, 0, g1));
  hipError_t ra = hipGraphAddEventRecordNode(&ne, gp, &n1, 1, ev);
  printf("hipGraphAddEventRecordNode: %s\n", hipGetErrorString(ra));
  if (ra == hipSuccess) {
    CK(hipGraphAddChildGraphNode(&n2, gp, &ne, 1, g2));
    CK(hipGraphInstantiate(&gpe, gp, nullptr, nullptr, 0));
    for (int rep = 0; rep < 4; ++rep) {
      auto t0 = std::chrono::steady_clock::now();
      CK(hipGraphLaunch(gpe, s));
      auto t1 = std::chrono::steady_clock::now();
      hipError_t r2 = hipEventSynchronize(ev);
      auto t2 = std::chrono::steady_clock::now();
      CK(hipStreamSynchronize(s));
      auto t3 = std::chrono::steady_clock::now();
      auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
      printf("explicit node, rep %d: launch %.0f us, wait for the event %.0f us (%s), then stream idle after %.0f us more\n", rep, us(t0, t1), us(t1, t2), hipGetErrorString(r2), us(t2, t3));
    }
  }
  return 0;
}
