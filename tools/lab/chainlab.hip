// tools/lab/chainlab.hip -- what would ONE launch per block row buy?  The upper bound, as a skeleton of the chain.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o chainlab chainlab.hip
// A block row of the factorisation chain is  [one workgroup works D us: the diagonal block]  ->  [C workgroups work S us each: the
// row solve, one of them hands the next diagonal tile back].  Three ways to run R such rows, work replaced by timed spins:
//   (a) two kernels per row, all of them nodes of one hipGraph (what the library does: k_diag, k_trsm16);
//   (b) one kernel per row: the C consumers are workgroups of the producer's launch and wait for its flag (agent-scope
//       release / acquire), the next row's launch follows;
//   (c) one kernel for all rows (persistent): producer and consumers hand over through flags in both directions.
// Reported: us per row beyond D + S -- the cost of the boundaries / hand-offs alone.  Every spin is bounded.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2);} } while (0)
constexpr int SPIN_MAX = 4000000;

__device__ __forceinline__ void work_us(double us) {           // (s_memrealtime: 100 MHz)
  const long long t0 = __builtin_amdgcn_s_memrealtime(), dt = (long long)(us * 100.0);
  while (__builtin_amdgcn_s_memrealtime() - t0 < dt) __builtin_amdgcn_s_sleep(1);
}
__device__ __forceinline__ bool wait_for(int* flag, int value) {
  __shared__ int ok;
  if (threadIdx.x == 0) {
    int spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < value && ++spins < SPIN_MAX) __builtin_amdgcn_s_sleep(1);
    ok = spins < SPIN_MAX;
  }
  __syncthreads();
  return ok != 0;
}
__device__ __forceinline__ void post(int* flag, int value) {
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(1024) void k_producer(double d_us, double* sink) { work_us(d_us); if (sink && threadIdx.x == 2000) sink[0] = 1.0; }
__global__ __launch_bounds__(1024) void k_consumers(double s_us, double* sink) { work_us(s_us); if (sink && threadIdx.x == 2000) sink[0] = 1.0; }
// (b) workgroup 0 = the producer, 1 .. C = the consumers of this row
__global__ __launch_bounds__(1024) void k_row(double d_us, double s_us, int* flags, int row, int* errors) {
  if (blockIdx.x == 0) { work_us(d_us); post(flags, row + 1); }
  else { if (!wait_for(flags, row + 1)) { if (threadIdx.x == 0) atomicAdd(errors, 1); return; } work_us(s_us); }
}
// (c) all rows in one launch: flags[0] producer -> consumers, flags[16] consumer 1 -> producer
__global__ __launch_bounds__(1024) void k_all(double d_us, double s_us, int* flags, int rows, int* errors) {
  for (int r = 0; r < rows; ++r) {
    if (blockIdx.x == 0) {
      if (r > 0 && !wait_for(flags + 16, r)) { if (threadIdx.x == 0) atomicAdd(errors, 1); return; }
      work_us(d_us); post(flags, r + 1);
    } else {
      if (!wait_for(flags, r + 1)) { if (threadIdx.x == 0) atomicAdd(errors, 1); return; }
      work_us(s_us);
      if (blockIdx.x == 1) post(flags + 16, r + 1);
    }
  }
}

int main(int argc, char** argv) {
  const int rows = argc > 1 ? atoi(argv[1]) : 32;
  int *flags, *errors; double* sink;
  HIPCHK(hipMalloc((void**)&flags, 256)); HIPCHK(hipMalloc((void**)&errors, 4)); HIPCHK(hipMalloc((void**)&sink, 8));
  hipStream_t st; HIPCHK(hipStreamCreate(&st));
  hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  printf("%d block rows; us per row beyond D + S\n", rows);
  for (int C : {16, 32}) for (double D : {0.0, 16.0}) for (double S : {0.0, 5.0}) {
    // (a) graph of 2 * rows kernels
    hipGraph_t g; hipGraphExec_t ge;
    HIPCHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int r = 0; r < rows; ++r) {
      hipLaunchKernelGGL(k_producer, dim3(1), dim3(1024), 0, st, D, (double*)nullptr);
      hipLaunchKernelGGL(k_consumers, dim3(C), dim3(1024), 0, st, S, (double*)nullptr);
    }
    HIPCHK(hipStreamEndCapture(st, &g)); HIPCHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    // (b) graph of rows kernels
    hipGraph_t gb; hipGraphExec_t geb;
    HIPCHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int r = 0; r < rows; ++r) hipLaunchKernelGGL(k_row, dim3(1 + C), dim3(1024), 0, st, D, S, flags, r, errors);
    HIPCHK(hipStreamEndCapture(st, &gb)); HIPCHK(hipGraphInstantiate(&geb, gb, nullptr, nullptr, 0));
    float ms[3] = {0, 0, 0};
    for (int rep = 0; rep < 3; ++rep) {
      HIPCHK(hipEventRecord(e0, st)); HIPCHK(hipGraphLaunch(ge, st)); HIPCHK(hipEventRecord(e1, st)); HIPCHK(hipEventSynchronize(e1));
      HIPCHK(hipEventElapsedTime(&ms[0], e0, e1));
      HIPCHK(hipMemsetAsync(flags, 0, 256, st)); HIPCHK(hipMemsetAsync(errors, 0, 4, st));
      HIPCHK(hipEventRecord(e0, st)); HIPCHK(hipGraphLaunch(geb, st)); HIPCHK(hipEventRecord(e1, st)); HIPCHK(hipEventSynchronize(e1));
      HIPCHK(hipEventElapsedTime(&ms[1], e0, e1));
      HIPCHK(hipMemsetAsync(flags, 0, 256, st));
      HIPCHK(hipEventRecord(e0, st)); hipLaunchKernelGGL(k_all, dim3(1 + C), dim3(1024), 0, st, D, S, flags, rows, errors);
      HIPCHK(hipEventRecord(e1, st)); HIPCHK(hipEventSynchronize(e1));
      HIPCHK(hipEventElapsedTime(&ms[2], e0, e1));
    }
    int err = 0; HIPCHK(hipMemcpy(&err, errors, 4, hipMemcpyDeviceToHost));
    const double base = D + S;
    printf("C=%2d consumers, D=%4.1f us, S=%3.1f us: (a) two kernels per row %6.2f   (b) one kernel per row %6.2f   (c) one kernel for all rows %6.2f   (timeouts %d)\n",
           C, D, S, ms[0] * 1e3 / rows - base, ms[1] * 1e3 / rows - base, ms[2] * 1e3 / rows - base, err);
    hipGraphExecDestroy(ge); hipGraphDestroy(g); hipGraphExecDestroy(geb); hipGraphDestroy(gb);
  }
  return 0;
}
