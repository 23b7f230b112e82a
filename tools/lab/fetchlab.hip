// tools/lab/fetchlab.hip -- what ONE workgroup (one CU) pulls per microsecond from memory that the launch before it wrote.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o fetchlab fetchlab.hip
// The dependent launches of a short light curve (k_diag -> k_trsm -> k_diag -> ... -> k_lauum_grad) each start by reading what the
// previous one stored, with a handful of workgroups on an otherwise idle chip.  Measured here: a producer launch (256 workgroups)
// writes `bytes`; a consumer launch of G workgroups of T threads reads them, every lane with U independent 16-byte loads in
// flight per round; reported per consumer configuration: microseconds from the workgroup's first instruction to its last load's
// arrival (s_memtime, calibrated against the launch's wall time) and the rate per workgroup.
//   G = 1: one CU alone.  G = 4 / 16 / 64: the same bytes spread over more CUs (each reads bytes / G).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2);} } while (0)

typedef double v2d __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_produce(double* buf, size_t n, double seed) {
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) buf[e] = seed + (double)e * 1e-9;
}

// every workgroup reads its contiguous share; lanes read 16-byte pieces, consecutive lanes consecutive pieces, U rounds unrolled
template <int U>
__global__ __launch_bounds__(1024) void k_consume(const double* buf, size_t doubles_per_wg, double* out, long long* ticks) {
  const long long t0 = __builtin_amdgcn_s_memtime();
  const v2d* p = reinterpret_cast<const v2d*>(buf + (size_t)blockIdx.x * doubles_per_wg);
  const size_t pieces = doubles_per_wg / 2;
  double s = 0.0;
  for (size_t base = 0; base < pieces; base += (size_t)blockDim.x * U) {
    v2d x[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t e = base + (size_t)u * blockDim.x + threadIdx.x;
      x[u] = e < pieces ? __builtin_nontemporal_load(p + e) : v2d{0.0, 0.0};
    }
#pragma unroll
    for (int u = 0; u < U; ++u) s += x[u][0] + x[u][1];
  }
  if (s == 12345.678) out[0] = s;                       // (keeps the loads)
  __syncthreads();
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) { ticks[2 * blockIdx.x] = t0; ticks[2 * blockIdx.x + 1] = t1; }
}

template <int U>
static void run(double* buf, double* out, long long* ticks, size_t bytes, int G, int T, double ticks_per_us) {
  const size_t n = bytes / 8;
  std::vector<long long> h(2 * (size_t)G);
  double best = 1e30, first = 0;
  for (int rep = 0; rep < 5; ++rep) {
    hipLaunchKernelGGL(k_produce, dim3(256), dim3(256), 0, 0, buf, n, (double)rep);
    hipLaunchKernelGGL(k_consume<U>, dim3(G), dim3(T), 0, 0, buf, n / G, out, ticks);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(h.data(), ticks, sizeof(long long) * h.size(), hipMemcpyDeviceToHost));
    double worst = 0;
    for (int g = 0; g < G; ++g) { const double us = (double)(h[2 * g + 1] - h[2 * g]) / ticks_per_us; if (us > worst) worst = us; }
    if (rep == 0) first = worst;
    if (worst < best) best = worst;
  }
  printf("bytes %7zu  G=%3d  T=%4d  U=%2d : longest workgroup %6.2f us (first run %6.2f)  -> %6.1f GB/s per workgroup, %7.1f GB/s in all\n",
         bytes, G, T, U, best, first, (double)bytes / G / best * 1e-3, (double)bytes / best * 1e-3);
}

// Column panels of a row-major matrix: workgroup g reads `rows` rows of block column (g % cols_per_row) -- 1 KB per row: wavefront w
// the 256 bytes at w * 256, lanes 0..15 one 16-byte piece each, lane >> 4 one of four consecutive rows per load, U loads in flight
// -- starting at row (g / cols_per_row) * rows; consecutive rows are `pitch` bytes apart.  Is a power-of-two pitch (ld = 4096
// doubles = 32 KB at N=4096) slower than a padded one?
template <int U>
__global__ __launch_bounds__(256) void k_panel(const char* buf, size_t pitch, int rows, int cols_per_row, double* out, long long* ticks) {
  const long long t0 = __builtin_amdgcn_s_memtime();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const char* p = buf + (size_t)(blockIdx.x / cols_per_row) * rows * pitch + (size_t)(blockIdx.x % cols_per_row) * 1024 + w * 256 + (lane & 15) * 16
                  + (size_t)(lane >> 4) * pitch;
  double s = 0.0;
  for (int r = 0; r < rows; r += 4 * U) {
    v2d x[U];
#pragma unroll
    for (int u = 0; u < U; ++u) x[u] = *reinterpret_cast<const v2d*>(p + (size_t)(r + 4 * u) * pitch);
#pragma unroll
    for (int u = 0; u < U; ++u) s += x[u][0] + x[u][1];
  }
  if (s == 12345.678) out[0] = s;
  __syncthreads();
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) { ticks[2 * blockIdx.x] = t0; ticks[2 * blockIdx.x + 1] = t1; }
}
static void run_panel(char* buf, size_t bufbytes, double* out, long long* ticks, size_t pitch, int rows, int G, int cols_per_row, double tpu) {
  if ((size_t)((G + cols_per_row - 1) / cols_per_row) * rows * pitch > bufbytes || (size_t)cols_per_row * 1024 > pitch) { printf("(skipped: buffer)\n"); return; }
  std::vector<long long> h(2 * (size_t)G);
  hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  double best = 1e30, best_ms = 1e30;
  for (int rep = 0; rep < 5; ++rep) {
    hipLaunchKernelGGL(k_produce, dim3(1024), dim3(256), 0, 0, (double*)buf, bufbytes / 8, (double)rep);
    HIPCHK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_panel<8>, dim3(G), dim3(256), 0, 0, buf, pitch, rows, cols_per_row, out, ticks);
    HIPCHK(hipEventRecord(e1, 0));
    HIPCHK(hipDeviceSynchronize());
    float ms = 0; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    HIPCHK(hipMemcpy(h.data(), ticks, sizeof(long long) * h.size(), hipMemcpyDeviceToHost));
    double worst = 0;
    for (int g = 0; g < G; ++g) { const double us = (double)(h[2 * g + 1] - h[2 * g]) / tpu; if (us > worst) worst = us; }
    if (worst < best) best = worst;
    if (ms < best_ms) best_ms = ms;
  }
  const double bytes = (double)G * rows * 1024.0;
  printf("panel: pitch %7zu B  rows %5d  G=%4d (%2d block columns side by side): longest workgroup %7.2f us, launch %7.2f us -> %7.1f GB/s in all\n",
         pitch, rows, G, cols_per_row, best, best_ms * 1e3, bytes / (best_ms * 1e-3) * 1e-9);
}

__global__ void k_spin(long long* ticks, long long n) {
  const long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < n) {}
  ticks[0] = t0; ticks[1] = __builtin_amdgcn_s_memtime();
}

int main() {
  double* buf; double* out; long long* ticks;
  const size_t bufbytes = (size_t)160 << 20;
  HIPCHK(hipMalloc((void**)&buf, bufbytes)); HIPCHK(hipMalloc((void**)&out, 64)); HIPCHK(hipMalloc((void**)&ticks, sizeof(long long) * 4096));
  // ticks of s_memtime per microsecond: a kernel that spins for 2 000 000 ticks, timed with events
  hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_spin, dim3(1), dim3(1), 0, 0, ticks, 1000LL);
  HIPCHK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(k_spin, dim3(1), dim3(1), 0, 0, ticks, 2000000LL);
  HIPCHK(hipEventRecord(e1, 0)); HIPCHK(hipEventSynchronize(e1));
  float ms = 0; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  const double tpu = 2000000.0 / (ms * 1e3);
  printf("s_memtime: %.1f ticks per microsecond (2 000 000 ticks in %.3f ms)\n", tpu, ms);
  for (size_t kb : {32, 64, 128, 256, 512}) {
    run<4>(buf, out, ticks, kb << 10, 1, 256, tpu);
    run<8>(buf, out, ticks, kb << 10, 1, 256, tpu);
    run<8>(buf, out, ticks, kb << 10, 1, 1024, tpu);
  }
  for (int G : {4, 16, 64}) { run<8>(buf, out, ticks, 128u << 10, G, 256, tpu); run<8>(buf, out, ticks, 512u << 10, G, 256, tpu); }
  // the whole chip, for scale
  run<8>(buf, out, ticks, 8u << 20, 256, 1024, tpu);
  // column panels at the pitches the matrix has: N=256 (2 KB), N=1024 (8 KB), N=4096 (32 KB), and the same plus 128 / 256 B
  for (size_t pitch : {(size_t)2048, (size_t)2048 + 128, (size_t)8192, (size_t)8192 + 128, (size_t)32768, (size_t)32768 + 128, (size_t)32768 + 256}) {
    run_panel((char*)buf, bufbytes, out, ticks, pitch, 128, 1, 1, tpu);                       // one workgroup, one block: latency regime
    run_panel((char*)buf, bufbytes, out, ticks, pitch, 128, 2 * (int)(pitch / 2048 > 16 ? 16 : pitch / 2048), (int)(pitch / 1024 > 32 ? 32 : pitch / 1024), tpu);
  }
  for (size_t pitch : {(size_t)32768, (size_t)32768 + 128, (size_t)32768 + 256}) {
    run_panel((char*)buf, bufbytes, out, ticks, pitch, 1024, 128, 32, tpu);                   // 32 block columns x 4 row ranges: 128 MB
    run_panel((char*)buf, bufbytes, out, ticks, pitch, 512, 256, 32, tpu);
    run_panel((char*)buf, bufbytes, out, ticks, pitch, 512, 256, 8, tpu);                     // 8 block columns, 32 row ranges
  }
  return 0;
}
