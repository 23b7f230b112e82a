// tools/lab/gemmlab.hip -- where the tile GEMM loop loses its cycles: the loop of pgm_gemm.h with parts switched off.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=fast -I../../pgmuvi_amd/csrc -o gemmlab gemmlab.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include <algorithm>
#include "pgm_gemm.h"

template <class C, bool GLOAD, bool SSTORE, bool BARRIER>
__device__ __forceinline__ void gemm_lab(double* __restrict__ lds, int nkb, const double* pa0, const double* pb0, int64_t ld, v4d (&acc)[C::TM][C::TN]) {
  const int t = threadIdx.x;
  const WavePos wp = wave_pos<C>();
  constexpr int KC = C::KC;
  const int nchunks = nkb * (NB / KC);
  constexpr int D = C::PF;
  v2d ra[D][C::VA], rb[D][C::VB];
  auto gload = [&](int c, v2d (&xa)[C::VA], v2d (&xb)[C::VB]) {
    const int kr = c * KC;
#pragma unroll
    for (int s = 0; s < C::VA; ++s) { const int e = t + C::NT * s, row = e / (C::BM / 2), c2 = e % (C::BM / 2);
      xa[s] = *reinterpret_cast<const v2d*>(pa0 + (int64_t)(kr + row) * ld + 2 * c2); }
#pragma unroll
    for (int s = 0; s < C::VB; ++s) { const int e = t + C::NT * s, row = e / (C::BN / 2), c2 = e % (C::BN / 2);
      xb[s] = *reinterpret_cast<const v2d*>(pb0 + (int64_t)(kr + row) * ld + 2 * c2); }
  };
  auto sstore = [&](int buf, const v2d (&xa)[C::VA], const v2d (&xb)[C::VB]) {
    double* As = lds + buf * C::STAGE; double* Bs = As + KC * C::PA;
#pragma unroll
    for (int s = 0; s < C::VA; ++s) { const int e = t + C::NT * s, row = e / (C::BM / 2), c2 = e % (C::BM / 2);
      *reinterpret_cast<v2d*>(As + row * C::PA + 2 * c2) = xa[s]; }
#pragma unroll
    for (int s = 0; s < C::VB; ++s) { const int e = t + C::NT * s, row = e / (C::BN / 2), c2 = e % (C::BN / 2);
      *reinterpret_cast<v2d*>(Bs + row * C::PB + 2 * c2) = xb[s]; }
  };
#pragma unroll
  for (int u = 0; u < D; ++u) gload(u, ra[u], rb[u]);
  sstore(0, ra[0], rb[0]);
  sstore(1, ra[0], rb[0]);
  __syncthreads();
  for (int c0 = 0; c0 < nchunks; c0 += D) {
#pragma unroll
    for (int u = 0; u < D; ++u) {
      const int c = c0 + u;
      if (GLOAD && c + D < nchunks) gload(c + D, ra[u], rb[u]);
      const double* As = lds + (c & 1) * C::STAGE;
      const double* Bs = As + KC * C::PA;
#pragma unroll
      for (int kk = 0; kk < KC / 4; ++kk) {
        const int krow = kk * 4 + (wp.lane >> 4);
        double a[C::TM], b[C::TN];
#pragma unroll
        for (int ti = 0; ti < C::TM; ++ti) a[ti] = As[krow * C::PA + wp.m0 + ti * 16 + (wp.lane & 15)];
#pragma unroll
        for (int tj = 0; tj < C::TN; ++tj) b[tj] = Bs[krow * C::PB + wp.n0 + tj * 16 + (wp.lane & 15)];
#pragma unroll
        for (int ti = 0; ti < C::TM; ++ti)
#pragma unroll
          for (int tj = 0; tj < C::TN; ++tj) acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ti], b[tj], acc[ti][tj], 0, 0, 0);
      }
      if (SSTORE && c + 1 < nchunks) sstore((c + 1) & 1, ra[(u + 1) % D], rb[(u + 1) % D]);
      if (BARRIER) __syncthreads();
    }
  }
}

template <class C, int WPS, bool GLOAD, bool SSTORE, bool BARRIER>
__global__ __launch_bounds__(C::NT, WPS) void k_lab(const double* A, int64_t ld, int nkb, double* out) {
  __shared__ __attribute__((aligned(16))) double lds[C::LDS_DOUBLES];
  v4d acc[C::TM][C::TN];
  acc_zero<C>(acc);
  const int64_t off = (int64_t)(blockIdx.x % 8) * 2 * C::BM;
  gemm_lab<C, GLOAD, SSTORE, BARRIER>(lds, nkb, A + off, A + off + C::BM, ld, acc);
  double s = 0.0;
#pragma unroll
  for (int ti = 0; ti < C::TM; ++ti)
#pragma unroll
    for (int tj = 0; tj < C::TN; ++tj) s += acc[ti][tj][0] + acc[ti][tj][3];
  if (s == 1.2345) out[0] = s;
}


// LDS-DMA staging: global_load_lds_dwordx4 writes each 1-KB operand row straight into its (padded) LDS row -- no staging
// registers, no ds_write.  Two LDS buffers: the DMA of chunk c+1 is issued after the barrier that freed its buffer, runs
// behind the MFMAs of chunk c and is waited for (vmcnt(0)) in front of the next barrier.
template <class C>
__device__ __forceinline__ void gemm_dma(double* __restrict__ lds, int nkb, const double* pa0, const double* pb0, int64_t ld, v4d (&acc)[C::TM][C::TN]) {
  static_assert(C::BM == 128 && C::BN == 128, "one wave instruction per operand row");
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const WavePos wp = wave_pos<C>();
  constexpr int KC = C::KC, NW = C::NT / 64, RPW = KC / NW;     // rows of each operand per wavefront and chunk
  const int nchunks = nkb * (NB / KC);
  typedef const double __attribute__((address_space(1)))* gptr;
  typedef double __attribute__((address_space(3)))* lptr;
  auto dma = [&](int c, int buf) {
    double* As = lds + buf * C::STAGE; double* Bs = As + KC * C::PA;
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      const int row = wave * RPW + r;
      __builtin_amdgcn_global_load_lds((gptr)(pa0 + (int64_t)(c * KC + row) * ld + 2 * lane), (lptr)(As + row * C::PA), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr)(pb0 + (int64_t)(c * KC + row) * ld + 2 * lane), (lptr)(Bs + row * C::PB), 16, 0, 0);
    }
  };
  dma(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int c = 0; c < nchunks; ++c) {
    if (c + 1 < nchunks) dma(c + 1, (c + 1) & 1);
    const double* As = lds + (c & 1) * C::STAGE;
    const double* Bs = As + KC * C::PA;
#pragma unroll
    for (int kk = 0; kk < KC / 4; ++kk) {
      const int krow = kk * 4 + (wp.lane >> 4);
      double a[C::TM], b[C::TN];
#pragma unroll
      for (int ti = 0; ti < C::TM; ++ti) a[ti] = As[krow * C::PA + wp.m0 + ti * 16 + (wp.lane & 15)];
#pragma unroll
      for (int tj = 0; tj < C::TN; ++tj) b[tj] = Bs[krow * C::PB + wp.n0 + tj * 16 + (wp.lane & 15)];
#pragma unroll
      for (int ti = 0; ti < C::TM; ++ti)
#pragma unroll
        for (int tj = 0; tj < C::TN; ++tj) acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ti], b[tj], acc[ti][tj], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
}

template <class C, int WPS>
__global__ __launch_bounds__(C::NT, WPS) void k_lab_dma(const double* A, int64_t ld, int nkb, double* out) {
  __shared__ __attribute__((aligned(16))) double lds[C::LDS_DOUBLES];
  v4d acc[C::TM][C::TN];
  acc_zero<C>(acc);
  const int64_t off = (int64_t)(blockIdx.x % 8) * 2 * C::BM;
  gemm_dma<C>(lds, nkb, A + off, A + off + C::BM, ld, acc);
  double s = 0.0;
#pragma unroll
  for (int ti = 0; ti < C::TM; ++ti)
#pragma unroll
    for (int tj = 0; tj < C::TN; ++tj) s += acc[ti][tj][0] + acc[ti][tj][3];
  if (s == 1.2345) out[0] = s;
  if (blockIdx.x == 0 && nkb == 1) out[8 + threadIdx.x] = acc[0][0][0] + acc[C::TM - 1][C::TN - 1][3];      // (check against the register-staged loop)
}
template <class C, int WPS>
__global__ __launch_bounds__(C::NT, WPS) void k_lab_ref(const double* A, int64_t ld, int nkb, double* out) {
  __shared__ __attribute__((aligned(16))) double lds[C::LDS_DOUBLES];
  v4d acc[C::TM][C::TN];
  acc_zero<C>(acc);
  gemm_lab<C, true, true, true>(lds, nkb, A, A + C::BM, ld, acc);
  if (blockIdx.x == 0) out[8 + threadIdx.x] = acc[0][0][0] + acc[C::TM - 1][C::TN - 1][3];
}

template <class C, int WPS>
void run_dma(const char* name, const double* A, int64_t ld, double* out, int blocks, int nkb) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 6; ++rep) {
    if (rep == 1) hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_lab_dma<C, WPS>), dim3(blocks), dim3(C::NT), 0, 0, A, ld, nkb, out);
  }
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double tf = 2.0 * C::BM * C::BN * NB * nkb * blocks / (ms * 1e-3) / 1e12;
  printf("%-64s blocks %4d nkb %2d: %8.1f us  %6.2f TFLOP/s = %.3f of 78.6%s\n", name, blocks, nkb, ms * 1e3, tf, tf / 78.6, hipGetLastError() == hipSuccess ? "" : " FAILED");
}

// cold operands: every workgroup multiplies k-blocks of its own (rows that no other workgroup touches, 1 GB in all)
template <class C, int WPS>
__global__ __launch_bounds__(C::NT, WPS) void k_lab_cold(const double* A, int64_t ld, int nkb, int64_t rows, double* out) {
  __shared__ __attribute__((aligned(16))) double lds[C::LDS_DOUBLES];
  v4d acc[C::TM][C::TN];
  acc_zero<C>(acc);
  const int64_t r0 = ((int64_t)blockIdx.x * nkb * NB) % (rows - (int64_t)nkb * NB);
  const int64_t c0 = (blockIdx.x % 8) * 2 * C::BM;
  gemm_lab<C, true, true, true>(lds, nkb, A + r0 * ld + c0, A + r0 * ld + c0 + C::BM, ld, acc);
  double s = 0.0;
#pragma unroll
  for (int ti = 0; ti < C::TM; ++ti)
#pragma unroll
    for (int tj = 0; tj < C::TN; ++tj) s += acc[ti][tj][0] + acc[ti][tj][3];
  if (s == 1.2345) out[0] = s;
}
template <class C, int WPS>
void run_cold(const char* name, const double* A, int64_t ld, int64_t rows, double* out, int blocks, int nkb) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 6; ++rep) {
    if (rep == 1) hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_lab_cold<C, WPS>), dim3(blocks), dim3(C::NT), 0, 0, A, ld, nkb, rows, out);
  }
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double tf = 2.0 * C::BM * C::BN * NB * nkb * blocks / (ms * 1e-3) / 1e12;
  printf("%-64s blocks %4d nkb %2d: %8.1f us  %6.2f TFLOP/s = %.3f of 78.6%s\n", name, blocks, nkb, ms * 1e3, tf, tf / 78.6, hipGetLastError() == hipSuccess ? "" : " FAILED");
}

template <class C, int WPS, bool G, bool S, bool B>
void run(const char* name, const double* A, int64_t ld, double* out, int blocks, int nkb) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 6; ++rep) {
    if (rep == 1) hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_lab<C, WPS, G, S, B>), dim3(blocks), dim3(C::NT), 0, 0, A, ld, nkb, out);
  }
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double tf = 2.0 * C::BM * C::BN * NB * nkb * blocks / (ms * 1e-3) / 1e12;
  printf("%-64s blocks %4d nkb %2d: %8.1f us  %6.2f TFLOP/s = %.3f of 78.6%s\n", name, blocks, nkb, ms * 1e3, tf, tf / 78.6, hipGetLastError() == hipSuccess ? "" : " FAILED");
}

int main() {
  const int64_t ld = 4096;
  double* A; double* out;
  hipMalloc(&A, sizeof(double) * ld * 2048); hipMalloc(&out, 8192);
  std::vector<double> h((size_t)ld * 2048);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 1.0 + 1e-3 * (double)((i * 2654435761u) % 1000);
  hipMemcpy(A, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice);
  using Big = TileCfg<128, 128, 64, 64, 2>;
  using Big1 = TileCfg<128, 128, 64, 64, 1>;
  using W8 = TileCfg<128, 128, 64, 32, 2, 512>;
  using F16 = TileCfg<128, 128, 32, 32, 2, 1024, 16>;
  using Sm = TileCfg<64, 64, 32, 32, 4>;
  {  // the LDS-DMA loop computes what the register-staged loop computes
    std::vector<double> r1(256), r2(256);
    hipLaunchKernelGGL((k_lab_ref<Big, 2>), dim3(1), dim3(256), 0, 0, A, ld, 1, out); hipMemcpy(r1.data(), out + 8, 256 * 8, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL((k_lab_dma<Big, 2>), dim3(1), dim3(256), 0, 0, A, ld, 1, out); hipMemcpy(r2.data(), out + 8, 256 * 8, hipMemcpyDeviceToHost);
    double md = 0; for (int i = 0; i < 256; ++i) md = std::max(md, std::abs(r1[i] - r2[i]));
    printf("LDS-DMA loop vs register-staged loop: max |difference| %.3e (values ~%.3e)\n", md, r1[0]);
  }
  {  // operands streamed from memory nobody else touches (1 GB: beyond the L2s and most of the Infinity Cache)
    const int64_t rows = 32768;
    double* Acold = nullptr;
    if (hipMalloc(&Acold, sizeof(double) * ld * rows) == hipSuccess) {
      hipMemset(Acold, 0, sizeof(double) * ld * rows);
      using F16p4 = TileCfg<128, 128, 32, 32, 4, 1024, 16>;
      using Big4 = TileCfg<128, 128, 64, 64, 4>;
      for (int nkb : {2, 4, 16}) {
        run_cold<F16, 4>("COLD operands, 16 waves 32x32/wave PF2, 1 WG/CU", Acold, ld, rows, out, 255, nkb);
        run_cold<F16p4, 4>("COLD operands, 16 waves 32x32/wave PF4, 1 WG/CU", Acold, ld, rows, out, 255, nkb);
        run_cold<Big, 2>("COLD operands, 4 waves 64x64/wave PF2, 2 WG/CU", Acold, ld, rows, out, 510, nkb);
        run_cold<Big4, 2>("COLD operands, 4 waves 64x64/wave PF4, 2 WG/CU", Acold, ld, rows, out, 510, nkb);
        // the same with a row pitch that is not a power of two (4096 + 16 doubles): do the 32-KB-strided rows of a panel collide?
        run_cold<Big, 2>("COLD operands, 4 waves 64x64/wave PF2, 2 WG/CU, pitch 4112", Acold, ld + 16, rows - 256, out, 510, nkb);
        run_cold<F16, 4>("COLD operands, 16 waves 32x32/wave PF2, 1 WG/CU, pitch 4112", Acold, ld + 16, rows - 256, out, 255, nkb);
      }
      hipFree(Acold);
    }
  }
  for (int nkb : {4, 16}) {
    run_dma<Big, 2>("4 waves 64x64/wave, LDS-DMA staging, 2 WG/CU", A, ld, out, 512, nkb);
    run_dma<W8, 2>("8 waves 64x32/wave, LDS-DMA staging, 1 WG/CU by regs", A, ld, out, 256, nkb);
    run<Big, 2, true, true, true>("4 waves 64x64/wave PF2, 2 WG/CU: full loop", A, ld, out, 512, nkb);
    run<Big, 2, false, true, true>("   no global loads", A, ld, out, 512, nkb);
    run<Big, 2, false, false, true>("   no global loads, no LDS stores", A, ld, out, 512, nkb);
    run<Big, 2, false, false, false>("   no global loads, no LDS stores, no barriers", A, ld, out, 512, nkb);
    run<Big, 2, true, true, true>("   full loop, 1 WG/CU (256 blocks)", A, ld, out, 256, nkb);
    run<Big, 2, false, false, false>("   bare MFMA + LDS reads, 1 WG/CU (256 blocks)", A, ld, out, 256, nkb);
    run<Big1, 2, true, true, true>("4 waves 64x64/wave PF1, 2 WG/CU: full loop", A, ld, out, 512, nkb);
    run<W8, 2, true, true, true>("8 waves 64x32/wave PF2, 1 WG/CU by regs: full loop", A, ld, out, 256, nkb);
    run<W8, 2, false, false, false>("   bare MFMA + LDS reads", A, ld, out, 256, nkb);
    run<F16, 4, true, true, true>("16 waves 32x32/wave PF2, 1 WG/CU: full loop", A, ld, out, 256, nkb);
    run<F16, 4, false, false, true>("   no global loads, no LDS stores", A, ld, out, 256, nkb);
    run<F16, 4, false, false, false>("   bare MFMA + LDS reads", A, ld, out, 256, nkb);
    run<Sm, 4, true, true, true>("4 waves 32x32/wave (64x64 tile) PF4, 4 WG/CU: full loop", A, ld, out, 1024, nkb);
    run<Sm, 4, false, false, false>("   bare MFMA + LDS reads", A, ld, out, 1024, nkb);
  }
  return 0;
}
