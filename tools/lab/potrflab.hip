// tools/lab/potrflab.hip -- the dependent chain of the 16x16 factorisation inside k_diag (diag_potrf16), alone on one wavefront.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o potrflab potrflab.hip
// A 128x128 diagonal block is 8 of these in a row, and nothing else of the block can start before each is done: 16 dependent
// pivots, each  pivot -> 1/sqrt -> Newton step -> scaled row -> next row's update -> next pivot.  Variants of how the pivot and
// the multipliers reach the other lanes:
//   0  v_readlane into scalar registers (rounds 1-3)
//   1  DPP row_newbcast operands of v_fmac_f64 (same arithmetic, same bits)
//   2  as 1 with the Newton step folded into the scaling of the row (one dependent instruction less; other rounding)
//   3 .. 6  timing only: without the Newton step / the 1/sqrt / the rank-4 updates between the mini-panels / the inverse image
// Prints clock ticks (s_memtime) per factorisation, and the largest difference to variant 0 and to a host factorisation.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2);} } while (0)
typedef double v4d __attribute__((ext_vector_type(4)));
typedef unsigned v2u_sw __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double readlane_d(double x, int l) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_readlane(lo, l);
  hi = __builtin_amdgcn_readlane(hi, l);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rsqrt_nr(double d) {
  const double y = __builtin_amdgcn_rsq(d);
  const double e = __builtin_fma(-d * y, y, 1.0);
  return __builtin_fma(0.5 * y, e, y);
}
__device__ __forceinline__ void rows_to_all(double x, double (&p)[4]) {
  unsigned h[2] = {(unsigned)__double2loint(x), (unsigned)__double2hiint(x)};
  unsigned o[4][2];
#pragma unroll
  for (int w = 0; w < 2; ++w) {
    const v2u_sw r = __builtin_amdgcn_permlane32_swap(h[w], h[w], false, false);
    const v2u_sw lo = __builtin_amdgcn_permlane16_swap(r[0], r[0], false, false);
    const v2u_sw hi = __builtin_amdgcn_permlane16_swap(r[1], r[1], false, false);
    o[0][w] = lo[0]; o[1][w] = lo[1]; o[2][w] = hi[0]; o[3][w] = hi[1];
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) p[q] = __hiloint2double((int)o[q][1], (int)o[q][0]);
}
// acc += (lane L of my 16-lane row of a) * b     /  acc -= ...
template <int L> __device__ __forceinline__ void fmac_bcast(double& acc, double a, double b) {
  asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a), "v"(b), "n"(L));
}
template <int L> __device__ __forceinline__ void fnmac_bcast(double& acc, double a, double b) {
  asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a), "v"(b), "n"(L));
}

template <int VAR, int M, int PL>
__device__ __forceinline__ void pivot(double (&pa)[4], double (&pb)[4]) {
  constexpr int LP = 4 * M + PL;
  if constexpr (VAR == 0 || VAR >= 3) {   // (3 .. 6: timing-only variants of 0)
    const double dp = readlane_d(pa[PL], LP);
    // (timing-only variants: 3 the hardware seed without its Newton step, 4 a multiplication in place of the whole 1/sqrt)
    const double rs = VAR == 3 ? __builtin_amdgcn_rsq(dp) : VAR == 4 ? dp * 0.0625 : rsqrt_nr(dp);
    pa[PL] *= rs;
    if (VAR != 6) pb[PL] *= rs;                     // (6: timing only, the factor without its inverse image -- is the chain bound by issue or by latency?)
#pragma unroll
    for (int ql = PL + 1; ql < 4; ++ql) {
      const double mult = readlane_d(pa[PL], 4 * M + ql);
      pa[ql] = __builtin_fma(-mult, pa[PL], pa[ql]);
      if (VAR != 6) pb[ql] = __builtin_fma(-mult, pb[PL], pb[ql]);
    }
  } else if constexpr (VAR == 1) {
    const double rs = rsqrt_nr(pa[PL]);           // every lane for its own element; lane LP of each row holds the pivot's
    double sa = 0.0, sb = 0.0;
    fmac_bcast<LP>(sa, rs, pa[PL]);
    fmac_bcast<LP>(sb, rs, pb[PL]);
    pa[PL] = sa; pb[PL] = sb;
    if constexpr (PL + 1 < 4) { fnmac_bcast<4 * M + PL + 1>(pa[PL + 1], sa, sa); fnmac_bcast<4 * M + PL + 1>(pb[PL + 1], sa, sb); }
    if constexpr (PL + 2 < 4) { fnmac_bcast<4 * M + PL + 2>(pa[PL + 2], sa, sa); fnmac_bcast<4 * M + PL + 2>(pb[PL + 2], sa, sb); }
    if constexpr (PL + 3 < 4) { fnmac_bcast<4 * M + PL + 3>(pa[PL + 3], sa, sa); fnmac_bcast<4 * M + PL + 3>(pb[PL + 3], sa, sb); }
  } else {
    const double d = pa[PL];
    const double y = __builtin_amdgcn_rsq(d);
    const double e = __builtin_fma(-d * y, y, 1.0);
    const double hy = 0.5 * y;
    double ya = 0.0, yb = 0.0, ha = 0.0, hb = 0.0;
    fmac_bcast<LP>(ya, y, pa[PL]);                // row * y
    fmac_bcast<LP>(ha, hy, pa[PL]);               // row * y/2
    fmac_bcast<LP>(ya, e, ha);                    // row * (y + y e / 2)
    fmac_bcast<LP>(yb, y, pb[PL]);
    fmac_bcast<LP>(hb, hy, pb[PL]);
    fmac_bcast<LP>(yb, e, hb);
    pa[PL] = ya; pb[PL] = yb;
    if constexpr (PL + 1 < 4) { fnmac_bcast<4 * M + PL + 1>(pa[PL + 1], ya, ya); fnmac_bcast<4 * M + PL + 1>(pb[PL + 1], ya, yb); }
    if constexpr (PL + 2 < 4) { fnmac_bcast<4 * M + PL + 2>(pa[PL + 2], ya, ya); fnmac_bcast<4 * M + PL + 2>(pb[PL + 2], ya, yb); }
    if constexpr (PL + 3 < 4) { fnmac_bcast<4 * M + PL + 3>(pa[PL + 3], ya, ya); fnmac_bcast<4 * M + PL + 3>(pb[PL + 3], ya, yb); }
  }
}

template <int VAR, int M>
__device__ __forceinline__ void panel(v4d& ua, v4d& va, double (&fa)[4], double (&fb)[4], int g) {
  double pa[4], pb[4];
  rows_to_all(ua[M], pa);
  if (VAR != 6) rows_to_all(va[M], pb); else { pb[0] = pb[1] = pb[2] = pb[3] = 0.0; }
  pivot<VAR, M, 0>(pa, pb); pivot<VAR, M, 1>(pa, pb); pivot<VAR, M, 2>(pa, pb); pivot<VAR, M, 3>(pa, pb);
  const double ra = (g == 0) ? pa[0] : (g == 1) ? pa[1] : (g == 2) ? pa[2] : pa[3];
  const double rb = (g == 0) ? pb[0] : (g == 1) ? pb[1] : (g == 2) ? pb[2] : pb[3];
  fa[M] = ra; fb[M] = rb;
  if (M < 3 && VAR != 5) {                          // (5: timing only, the pivots without the rank-4 updates between the panels)
    ua = __builtin_amdgcn_mfma_f64_16x16x4f64(-ra, ra, ua, 0, 0, 0);
    if (VAR != 6) va = __builtin_amdgcn_mfma_f64_16x16x4f64(-ra, rb, va, 0, 0, 0);
  }
}

template <int VAR>
__global__ __launch_bounds__(64) void k_potrf(const double* __restrict__ A, double* __restrict__ out, long long* ticks, int iters) {
  const int lane = threadIdx.x, g = lane >> 4, n = lane & 15;
  v4d u0;
#pragma unroll
  for (int r = 0; r < 4; ++r) u0[r] = A[(g + 4 * r) * 16 + n];
  double fa[4] = {0, 0, 0, 0}, fb[4] = {0, 0, 0, 0};
  double carry = 0.0;
  const long long t0 = __builtin_amdgcn_s_memtime();
#pragma clang loop unroll(disable)
  for (int it = 0; it < iters; ++it) {
    v4d ua, va;
#pragma unroll
    for (int r = 0; r < 4; ++r) { ua[r] = u0[r] + carry; va[r] = (g + 4 * r == n) ? 1.0 : 0.0; }
    panel<VAR, 0>(ua, va, fa, fb, g); panel<VAR, 1>(ua, va, fa, fb, g); panel<VAR, 2>(ua, va, fa, fb, g); panel<VAR, 3>(ua, va, fa, fb, g);
    carry = readlane_d(fa[3], 63) * 1e-300;        // (the last pivot: the next round waits for the whole chain)
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
#pragma unroll
  for (int r = 0; r < 4; ++r) { out[(g + 4 * r) * 16 + n] = fa[r]; out[256 + (g + 4 * r) * 16 + n] = fb[r]; }
  if (lane == 0) ticks[0] = t1 - t0;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  std::vector<double> A(256), U(256), V(256);
  srand(7);
  std::vector<double> B(256);
  for (auto& b : B) b = rand() / (double)RAND_MAX - 0.5;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
    double s = (i == j) ? 4.0 : 0.0;
    for (int k = 0; k < 16; ++k) s += B[i * 16 + k] * B[j * 16 + k];
    A[i * 16 + j] = s;
  }
  // host: A = U^T U, V = U^-T
  std::vector<double> W = A;
  for (int j = 0; j < 16; ++j) {
    const double d = std::sqrt(W[j * 16 + j]);
    for (int c = j; c < 16; ++c) W[j * 16 + c] /= d;
    for (int i = j + 1; i < 16; ++i) for (int c = i; c < 16; ++c) W[i * 16 + c] -= W[j * 16 + i] * W[j * 16 + c];
  }
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) U[i * 16 + j] = j >= i ? W[i * 16 + j] : 0.0;
  for (int c = 0; c < 16; ++c) for (int i = 0; i < 16; ++i) {       // U^T V[:, c] = e_c
    double s = (i == c) ? 1.0 : 0.0;
    for (int k = 0; k < i; ++k) s -= U[k * 16 + i] * V[k * 16 + c];
    V[i * 16 + c] = s / U[i * 16 + i];
  }
  double *dA, *dO; long long* dT;
  HIPCHK(hipMalloc((void**)&dA, 256 * 8)); HIPCHK(hipMalloc((void**)&dO, 512 * 8)); HIPCHK(hipMalloc((void**)&dT, 8));
  HIPCHK(hipMemcpy(dA, A.data(), 256 * 8, hipMemcpyHostToDevice));
  std::vector<double> ref(512), got(512);
  for (int var = 0; var < 7; ++var) {
    long long t = 0;
    hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      HIPCHK(hipEventRecord(e0, 0));
      if (var == 0) hipLaunchKernelGGL(k_potrf<0>, dim3(1), dim3(64), 0, 0, dA, dO, dT, iters);
      if (var == 1) hipLaunchKernelGGL(k_potrf<1>, dim3(1), dim3(64), 0, 0, dA, dO, dT, iters);
      if (var == 2) hipLaunchKernelGGL(k_potrf<2>, dim3(1), dim3(64), 0, 0, dA, dO, dT, iters);
      if (var == 3) hipLaunchKernelGGL(k_potrf<3>, dim3(1), dim3(64), 0, 0, dA, dO, dT, iters);
      if (var == 4) hipLaunchKernelGGL(k_potrf<4>, dim3(1), dim3(64), 0, 0, dA, dO, dT, iters);
      if (var == 5) hipLaunchKernelGGL(k_potrf<5>, dim3(1), dim3(64), 0, 0, dA, dO, dT, iters);
      if (var == 6) hipLaunchKernelGGL(k_potrf<6>, dim3(1), dim3(64), 0, 0, dA, dO, dT, iters);
      HIPCHK(hipEventRecord(e1, 0));
      HIPCHK(hipDeviceSynchronize());
      HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    }
    HIPCHK(hipMemcpy(&t, dT, 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(got.data(), dO, 512 * 8, hipMemcpyDeviceToHost));
    if (var == 0) ref = got;
    double dv = 0, dh = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
      if (j >= i) { dv = std::fmax(dv, std::fabs(got[i * 16 + j] - ref[i * 16 + j])); dh = std::fmax(dh, std::fabs(got[i * 16 + j] - U[i * 16 + j])); }
      if (j <= i) { dv = std::fmax(dv, std::fabs(got[256 + i * 16 + j] - ref[256 + i * 16 + j])); dh = std::fmax(dh, std::fabs(got[256 + i * 16 + j] - V[i * 16 + j])); }
    }
    printf("variant %d: %8.1f ticks, %7.1f ns per 16x16 factorisation (%.1f ns per pivot)   max|d| to variant 0 %.2e, to the host's %.2e\n",
           var, (double)t / iters, ms * 1e6 / iters, ms * 1e6 / iters / 16, dv, dh);
  }
  return 0;
}
