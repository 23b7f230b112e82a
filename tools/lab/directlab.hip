// tools/lab/directlab.hip -- the tile GEMM with NO LDS: every wavefront loads its MFMA fragments straight from memory.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=fast -I../../pgmuvi_amd/csrc -o directlab directlab.hip
// acc[m][n] += sum_k A[k][m] B[k][n] on row-major k-major operands.  A wavefront owns a 64x64 sub-tile as 4x4 MFMA tiles, but
// with the rows (and columns) of the sub-tile dealt to the MFMA tiles round-robin: MFMA tile ti holds rows m0 + 4 i + ti
// (i = 0..15), so that lane (k = l>>4, i = l&15) needs A[k][m0 + 4 i .. 4 i + 3] -- 32 contiguous bytes -- and one k-step
// (4 rows) of all four A fragments is two 16-byte loads per lane, 4 x 512 contiguous bytes per wavefront.  The same for B.
// No staging registers -> LDS stores -> barrier -> LDS reads: the loop is 4 loads and 16 MFMAs per k-step, wavefronts never
// wait for one another, and the prefetch distance is PD k-steps of registers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include "pgm_gemm.h"

typedef unsigned v4u __attribute__((ext_vector_type(4)));

template <int PD, int NW, int WPS = (NW == 4 ? 2 : 1), bool RMW = false>   // NW wavefronts per workgroup: 4 (2x2 of 64x64) or 8 (4x2 of 32(m)x64(n)); RMW: the C tile is read first and written last
__global__ __launch_bounds__(NW * 64, WPS) void k_direct(const double* A, int64_t ld, int nkb, int cold, int64_t rows, double* out, double* Ct = nullptr) {
  constexpr int TM = (NW == 4) ? 4 : 2, TN = 4;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m0 = (wave / 2) * (16 * TM), n0 = (wave % 2) * 64;
  const int64_t r0 = cold ? ((int64_t)blockIdx.x * nkb * NB) % (rows - (int64_t)nkb * NB) : 0;
  const int64_t c0 = (blockIdx.x % 8) * 256;
  const double* pa = A + r0 * ld + c0;
  const double* pb = pa + 128;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)pa, 0, 0x7fffffff, 0x00027000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)pb, 0, 0x7fffffff, 0x00027000);
  const int g = lane >> 4, i = lane & 15;
  const int voa = (g * (int)ld + m0 + TM * i) * 8, vob = (g * (int)ld + n0 + TN * i) * 8;
  const int step = 4 * (int)ld * 8;                              // bytes per k-step
  v4d acc[TM][TN];
  double* cbase = RMW ? Ct + (int64_t)blockIdx.x * 128 * 128 : nullptr;     // (a private, contiguous 128x128 tile per workgroup)
#pragma unroll
  for (int ti = 0; ti < TM; ++ti)
#pragma unroll
    for (int tj = 0; tj < TN; ++tj) acc[ti][tj] = v4d{0, 0, 0, 0};
  if (RMW) {
#pragma unroll
    for (int ti = 0; ti < TM; ++ti)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double* q = cbase + (m0 + TM * (g + 4 * r) + ti) * 128 + n0 + TN * i;
        const v2d x0 = *reinterpret_cast<const v2d*>(q), x1 = *reinterpret_cast<const v2d*>(q + 2);
        acc[ti][0][r] = -x0[0]; acc[ti][1][r] = -x0[1]; acc[ti][2][r] = -x1[0]; acc[ti][3][r] = -x1[1];
      }
  }
  double a[PD][TM], b[PD][TN];
  auto load = [&](int u, int ks) {
    const int so = ks * step;
    if (TM == 4) {
      const v4u x0 = __builtin_amdgcn_raw_buffer_load_b128(ra, voa, so, 0), x1 = __builtin_amdgcn_raw_buffer_load_b128(ra, voa + 16, so, 0);
      a[u][0] = __builtin_bit_cast(v2d, x0)[0]; a[u][1] = __builtin_bit_cast(v2d, x0)[1];
      a[u][2 % TM] = __builtin_bit_cast(v2d, x1)[0]; a[u][3 % TM] = __builtin_bit_cast(v2d, x1)[1];
    } else {
      const v4u x0 = __builtin_amdgcn_raw_buffer_load_b128(ra, voa, so, 0);
      a[u][0] = __builtin_bit_cast(v2d, x0)[0]; a[u][1] = __builtin_bit_cast(v2d, x0)[1];
    }
    const v4u y0 = __builtin_amdgcn_raw_buffer_load_b128(rb, vob, so, 0), y1 = __builtin_amdgcn_raw_buffer_load_b128(rb, vob + 16, so, 0);
    b[u][0] = __builtin_bit_cast(v2d, y0)[0]; b[u][1] = __builtin_bit_cast(v2d, y0)[1];
    b[u][2] = __builtin_bit_cast(v2d, y1)[0]; b[u][3] = __builtin_bit_cast(v2d, y1)[1];
  };
  const int nks = nkb * (NB / 4);
#pragma unroll
  for (int u = 0; u < PD; ++u) load(u, u);
  for (int ks = 0; ks < nks; ks += PD) {
#pragma unroll
    for (int u = 0; u < PD; ++u) {
#pragma unroll
      for (int ti = 0; ti < TM; ++ti)
#pragma unroll
        for (int tj = 0; tj < TN; ++tj) acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][ti], b[u][tj], acc[ti][tj], 0, 0, 0);
      if (ks + u + PD < nks) load(u, ks + u + PD);
    }
  }
  if (RMW) {
#pragma unroll
    for (int ti = 0; ti < TM; ++ti)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double* q = cbase + (m0 + TM * (g + 4 * r) + ti) * 128 + n0 + TN * i;
        *reinterpret_cast<v2d*>(q) = v2d{-acc[ti][0][r], -acc[ti][1][r]}; *reinterpret_cast<v2d*>(q + 2) = v2d{-acc[ti][2][r], -acc[ti][3][r]};
      }
    return;
  }
  double s = 0.0;
#pragma unroll
  for (int ti = 0; ti < TM; ++ti)
#pragma unroll
    for (int tj = 0; tj < TN; ++tj) s += acc[ti][tj][0] + acc[ti][tj][1] + acc[ti][tj][2] + acc[ti][tj][3];
  if (s == 1.2345) out[0] = s;
  if (blockIdx.x == 0 && nkb == 1 && !cold) {                    // check: the sum of the whole 128x128 tile
    s += __shfl_xor(s, 32, 64); s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 8, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 1, 64);
    if (lane == 0) out[16 + wave] = s;
  }
}

template <class C, int WPS>
__global__ __launch_bounds__(C::NT, WPS) void k_ref(const double* A, int64_t ld, int nkb, int cold, int64_t rows, double* out) {
  __shared__ __attribute__((aligned(16))) double lds[C::LDS_DOUBLES];
  v4d acc[C::TM][C::TN];
  acc_zero<C>(acc);
  const int64_t r0 = cold ? ((int64_t)blockIdx.x * nkb * NB) % (rows - (int64_t)nkb * NB) : 0;
  const int64_t c0 = (blockIdx.x % 8) * 256;
  const double* pa0 = A + r0 * ld + c0;
  gemm_tn<C>(lds, nkb, [&](int kb, const double*& pa, int64_t& lda, const double*& pb, int64_t& ldb) {
    pa = pa0 + (int64_t)kb * NB * ld; lda = ld; pb = pa + 128; ldb = ld; }, acc);
  double s = 0.0;
#pragma unroll
  for (int ti = 0; ti < C::TM; ++ti)
#pragma unroll
    for (int tj = 0; tj < C::TN; ++tj) s += acc[ti][tj][0] + acc[ti][tj][1] + acc[ti][tj][2] + acc[ti][tj][3];
  if (s == 1.2345) out[0] = s;
  if (blockIdx.x == 0 && nkb == 1 && !cold) {
    const int lane = threadIdx.x & 63;
    s += __shfl_xor(s, 32, 64); s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 8, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 1, 64);
    if (lane == 0) out[32 + (threadIdx.x >> 6)] = s;
  }
}

template <class L> void timeit(const char* name, L&& launch, int blocks, int nkb) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 6; ++rep) { if (rep == 1) hipEventRecord(e0, 0); launch(); }
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double tf = 2.0 * 128 * 128 * NB * nkb * blocks / (ms * 1e-3) / 1e12;
  printf("%-58s blocks %4d nkb %2d: %8.1f us  %6.2f TFLOP/s = %.3f of 78.6%s\n", name, blocks, nkb, ms * 1e3, tf, tf / 78.6, hipGetLastError() == hipSuccess ? "" : " FAILED");
}

int main() {
  const int64_t ld = 4096, rows = 32768;
  double* A; double* out;
  if (hipMalloc(&A, sizeof(double) * ld * rows) != hipSuccess) return 1;
  hipMalloc(&out, 8192);
  std::vector<double> h((size_t)ld * 256);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 1.0 + 1e-3 * (double)((i * 2654435761u) % 1000);
  hipMemset(A, 0, sizeof(double) * ld * rows);
  hipMemcpy(A, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice);
  using Big = TileCfg<128, 128, 64, 64, 2>;
  {
    hipMemset(out, 0, 8192);
    hipLaunchKernelGGL((k_direct<4, 4>), dim3(1), dim3(256), 0, 0, A, ld, 1, 0, rows, out);
    hipLaunchKernelGGL((k_ref<Big, 2>), dim3(1), dim3(256), 0, 0, A, ld, 1, 0, rows, out);
    double r[64]; hipMemcpy(r, out, 64 * 8, hipMemcpyDeviceToHost);
    const double d = r[16] + r[17] + r[18] + r[19], f = r[32] + r[33] + r[34] + r[35];
    printf("tile sum: direct %.12e  LDS-staged %.12e  relative difference %.2e\n", d, f, std::fabs(d - f) / std::fabs(f));
  }
  for (int cold = 0; cold < 2; ++cold)
    for (int nkb : {2, 4, 16}) {
      const char* tag = cold ? "COLD" : "hot ";
      char name[128];
      snprintf(name, sizeof name, "%s LDS-staged 4 waves 64x64 PF2, 2 WG/CU", tag);
      timeit(name, [&] { hipLaunchKernelGGL((k_ref<Big, 2>), dim3(512), dim3(256), 0, 0, A, ld, nkb, cold, rows, out); }, 512, nkb);
      snprintf(name, sizeof name, "%s direct 4 waves 64x64, PD 2, 2 WG/CU", tag);
      timeit(name, [&] { hipLaunchKernelGGL((k_direct<2, 4>), dim3(512), dim3(256), 0, 0, A, ld, nkb, cold, rows, out); }, 512, nkb);
      snprintf(name, sizeof name, "%s direct 4 waves 64x64, PD 4, 2 WG/CU", tag);
      timeit(name, [&] { hipLaunchKernelGGL((k_direct<4, 4>), dim3(512), dim3(256), 0, 0, A, ld, nkb, cold, rows, out); }, 512, nkb);
      snprintf(name, sizeof name, "%s direct 4 waves 64x64, PD 8 (may spill), 2 WG/CU", tag);
      timeit(name, [&] { hipLaunchKernelGGL((k_direct<8, 4>), dim3(512), dim3(256), 0, 0, A, ld, nkb, cold, rows, out); }, 512, nkb);
      snprintf(name, sizeof name, "%s direct 8 waves 32x64, PD 8, 1 WG/CU (2 waves/SIMD)", tag);
      timeit(name, [&] { hipLaunchKernelGGL((k_direct<8, 8>), dim3(256), dim3(512), 0, 0, A, ld, nkb, cold, rows, out); }, 256, nkb);
      snprintf(name, sizeof name, "%s direct 8 waves 32x64, PD 16, 1 WG/CU (2 waves/SIMD)", tag);
      timeit(name, [&] { hipLaunchKernelGGL((k_direct<16, 8>), dim3(256), dim3(512), 0, 0, A, ld, nkb, cold, rows, out); }, 256, nkb);
    }
  {  // trailing-update shape: private C tile read first and written last, operands cold, k-depth 512 / 1024; 2 or 3 workgroups per CU
    double* Ct; hipMalloc(&Ct, sizeof(double) * 128 * 128 * 4096); hipMemset(Ct, 0, sizeof(double) * 128 * 128 * 4096);
    for (int nkb : {1, 2, 4, 8}) {
      timeit("COLD + C tile RMW, direct PD 4, 2 WG/CU, 2048 tiles", [&] { hipLaunchKernelGGL((k_direct<4, 4, 2, true>), dim3(2048), dim3(256), 0, 0, A, ld, nkb, 1, rows, out, Ct); }, 2048, nkb);
      timeit("COLD + C tile RMW, direct PD 2, 2 WG/CU, 2048 tiles", [&] { hipLaunchKernelGGL((k_direct<2, 4, 2, true>), dim3(2048), dim3(256), 0, 0, A, ld, nkb, 1, rows, out, Ct); }, 2048, nkb);
      timeit("COLD + C tile RMW, direct PD 2, 3 WG/CU, 2048 tiles", [&] { hipLaunchKernelGGL((k_direct<2, 4, 3, true>), dim3(2048), dim3(256), 0, 0, A, ld, nkb, 1, rows, out, Ct); }, 2048, nkb);
      timeit("COLD + C tile RMW, direct PD 4, 3 WG/CU (spills?), 2048 tiles", [&] { hipLaunchKernelGGL((k_direct<4, 4, 3, true>), dim3(2048), dim3(256), 0, 0, A, ld, nkb, 1, rows, out, Ct); }, 2048, nkb);
      timeit("COLD no RMW, direct PD 4, 2 WG/CU, 2048 tiles", [&] { hipLaunchKernelGGL((k_direct<4, 4, 2, false>), dim3(2048), dim3(256), 0, 0, A, ld, nkb, 1, rows, out, Ct); }, 2048, nkb);
    }
  }
  return 0;
}
