// handoff.hip -- how long does a producer -> consumer hand-off between workgroups take INSIDE one launch on gfx950?
// (The question behind a dataflow version of the factorisation chain: DESIGN.md section 4, "open question".)
// Pairs of workgroups (neighbouring workgroup ids: different XCDs, whose L2s are not coherent with each other) play
// ping-pong: the producer writes a slab of `bytes`, releases a flag at agent scope; the consumer acquires it, reads the
// slab, checks it and releases an acknowledgement.  Reported: microseconds per one-way hand-off (half a round trip),
// for 1 pair (latency) and for many pairs at once (the per-XCD write-back / invalidate under load).
// Every spin loop is bounded: a lost flag ends the kernel with an error count, never a hang.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int SPIN_MAX = 2000000;

__global__ __launch_bounds__(256) void k_pingpong(double* slabs, int* flags, int slab_doubles, int rounds, int stride, int* errors) {
  const int pair = blockIdx.x / 2, role = blockIdx.x & 1;
  double* slab = slabs + (size_t)pair * slab_doubles;
  int* ready = flags + pair * stride;            // producer -> consumer
  int* ack = flags + pair * stride + stride / 2; // consumer -> producer
  __shared__ int ok;
  for (int r = 1; r <= rounds; ++r) {
    if (role == 0) {
      for (int e = threadIdx.x; e < slab_doubles; e += 256) slab[e] = (double)(r + e);
      __syncthreads();
      if (threadIdx.x == 0) {
        __hip_atomic_store(ready, r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(ack, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < r && ++spins < SPIN_MAX) __builtin_amdgcn_s_sleep(1);
        ok = spins < SPIN_MAX;
      }
      __syncthreads();
      if (!ok) { if (threadIdx.x == 0) atomicAdd(errors, 1); return; }
    } else {
      if (threadIdx.x == 0) {
        int spins = 0;
        while (__hip_atomic_load(ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < r && ++spins < SPIN_MAX) __builtin_amdgcn_s_sleep(1);
        ok = spins < SPIN_MAX;
      }
      __syncthreads();
      if (!ok) { if (threadIdx.x == 0) atomicAdd(errors, 1); return; }
      __atomic_thread_fence(__ATOMIC_ACQUIRE);        // (every wave, not only the polling thread, must see the slab)
      double bad = 0.0;
      for (int e = threadIdx.x; e < slab_doubles; e += 256) bad += (slab[e] != (double)(r + e)) ? 1.0 : 0.0;
      if (bad != 0.0) atomicAdd(errors + 1, 1);
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_store(ack, r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// the same traffic as two dependent launches per hand-off, for comparison
__global__ __launch_bounds__(256) void k_write(double* slabs, int slab_doubles, int r) {
  double* slab = slabs + (size_t)blockIdx.x * slab_doubles;
  for (int e = threadIdx.x; e < slab_doubles; e += 256) slab[e] = (double)(r + e);
}
__global__ __launch_bounds__(256) void k_read(const double* slabs, int slab_doubles, int r, int* errors) {
  const double* slab = slabs + (size_t)blockIdx.x * slab_doubles;
  double bad = 0.0;
  for (int e = threadIdx.x; e < slab_doubles; e += 256) bad += (slab[e] != (double)(r + e)) ? 1.0 : 0.0;
  if (bad != 0.0) atomicAdd(errors + 1, 1);
}

int main() {
  const int rounds = 200, stride = 64;
  for (int kb : {4, 32, 128}) {
    const int slab_doubles = kb * 1024 / 8;
    for (int pairs : {1, 8, 64, 120}) {
      double* slabs; int *flags, *errors;
      hipMalloc((void**)&slabs, sizeof(double) * (size_t)pairs * slab_doubles);
      hipMalloc((void**)&flags, sizeof(int) * pairs * stride);
      hipMalloc((void**)&errors, sizeof(int) * 2);
      hipMemset(flags, 0, sizeof(int) * pairs * stride);
      hipMemset(errors, 0, sizeof(int) * 2);
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipLaunchKernelGGL(k_pingpong, dim3(2 * pairs), dim3(256), 0, 0, slabs, flags, slab_doubles, 2, stride, errors);   // warm-up
      hipDeviceSynchronize();
      hipMemset(flags, 0, sizeof(int) * pairs * stride);
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(k_pingpong, dim3(2 * pairs), dim3(256), 0, 0, slabs, flags, slab_doubles, rounds, stride, errors);
      hipEventRecord(e1, 0);
      hipDeviceSynchronize();
      float ms = 0; hipEventElapsedTime(&ms, e0, e1);
      int herr[2]; hipMemcpy(herr, errors, sizeof(herr), hipMemcpyDeviceToHost);
      // launches: write then read, `rounds` times
      hipEventRecord(e0, 0);
      for (int r = 1; r <= rounds; ++r) {
        hipLaunchKernelGGL(k_write, dim3(pairs), dim3(256), 0, 0, slabs, slab_doubles, r);
        hipLaunchKernelGGL(k_read, dim3(pairs), dim3(256), 0, 0, slabs, slab_doubles, r, errors);
      }
      hipEventRecord(e1, 0);
      hipDeviceSynchronize();
      float ms2 = 0; hipEventElapsedTime(&ms2, e0, e1);
      printf("slab %4d KB, %3d pairs: in-launch hand-off %6.2f us one way (timeouts %d, bad slabs %d);  two dependent launches %6.2f us per write+read\n",
             kb, pairs, ms * 1e3 / rounds / 2.0, herr[0], herr[1], ms2 * 1e3 / rounds);
      hipFree(slabs); hipFree(flags); hipFree(errors);
    }
  }
  return 0;
}
