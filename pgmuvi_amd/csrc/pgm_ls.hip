// pgm_ls.hip -- Lomb-Scargle seeding of the spectral-mixture frequencies.
//
// Replaces astropy.timeseries.LombScargle(t, y, dy).power(freq) as pgmuvi calls it from
// Lightcurve.fit_LS (/root/reference/pgmuvi/lightcurve.py:4320, 4504-4511): the floating-mean
// ("generalised", Zechmeister & Kuerster 2009) periodogram, standard normalisation, exact sums in fp64,
//   P(f) = [SS YC^2 + CC YS^2 - 2 CS YC YS] / [YY (CC SS - CS^2)]
// with w_i = dy_i^-2 / sum, y centred on its weighted mean, C = sum w cos, S = sum w sin,
// YC = sum w y cos, YS = sum w y sin, CC = sum w cos^2 - C^2, SS = sum w sin^2 - S^2, CS = sum w cos sin - C S
// (the C, S corrections only with fit_mean).  One thread per frequency, the points staged through LDS in
// chunks; batch of light curves on gridDim.z sharing one frequency grid.  The work is N x Nf sincos
// evaluations (2e8 at N=4096 with pgmuvi's 5 x Nyquist, 5 samples per peak grid): VALU-bound, ~0.4 ms.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/pgmuvi_hip.h"

namespace {

constexpr int LS_THREADS = 256;
constexpr int LS_CHUNK = 512;

// scratch layout per light curve: w[n], wy[n], yy (light curve b at scratch + b * stride: 2 n + 1 for the exact sums, the FFT form's
// longer record for the fast path)
__global__ __launch_bounds__(LS_THREADS) void k_ls_prepare(const double* __restrict__ y, const double* __restrict__ dy,
                                                           int64_t n, double* __restrict__ scratch, int64_t stride) {
  const int b = blockIdx.x, t = threadIdx.x;
  const double* yb = y + (int64_t)b * n;
  const double* db = dy ? dy + (int64_t)b * n : nullptr;
  double* w = scratch + (int64_t)b * stride;
  double* wy = w + n;
  __shared__ double red[LS_THREADS];
  auto reduce = [&](double v) {
    red[t] = v;
    __syncthreads();
    for (int s = LS_THREADS / 2; s > 0; s >>= 1) { if (t < s) red[t] += red[t + s]; __syncthreads(); }
    const double r = red[0];
    __syncthreads();
    return r;
  };
  double sw = 0.0;
  for (int64_t i = t; i < n; i += LS_THREADS) { const double d = db ? db[i] : 1.0; sw += 1.0 / (d * d); }
  sw = reduce(sw);
  double sy = 0.0;
  for (int64_t i = t; i < n; i += LS_THREADS) { const double d = db ? db[i] : 1.0; const double wi = 1.0 / (d * d) / sw; w[i] = wi; sy += wi * yb[i]; }
  sy = reduce(sy);
  double yy = 0.0;
  for (int64_t i = t; i < n; i += LS_THREADS) { const double yc = yb[i] - sy; wy[i] = w[i] * yc; yy += w[i] * yc * yc; }
  yy = reduce(yy);
  if (t == 0) wy[n] = yy;
}

__global__ __launch_bounds__(LS_THREADS) void k_lomb_scargle(const double* __restrict__ tt, const double* __restrict__ scratch,
                                                             int64_t n, const double* __restrict__ freq, int64_t nf,
                                                             int fit_mean, double* __restrict__ power) {
  const int b = blockIdx.z;
  const int64_t m = (int64_t)blockIdx.x * LS_THREADS + threadIdx.x;
  const double* tb = tt + (int64_t)b * n;
  const double* w = scratch + (int64_t)b * (2 * n + 1);
  const double* wy = w + n;
  __shared__ double ts[LS_CHUNK], ws[LS_CHUNK], wys[LS_CHUNK];
  const double f = (m < nf) ? freq[m] : 0.0;
  double C = 0.0, S = 0.0, YC = 0.0, YS = 0.0, CC = 0.0, CS = 0.0;
  for (int64_t i0 = 0; i0 < n; i0 += LS_CHUNK) {
    const int cnt = (int)((n - i0 < LS_CHUNK) ? n - i0 : LS_CHUNK);
    __syncthreads();
    for (int e = threadIdx.x; e < cnt; e += LS_THREADS) { ts[e] = tb[i0 + e]; ws[e] = w[i0 + e]; wys[e] = wy[i0 + e]; }
    __syncthreads();
    for (int e = 0; e < cnt; ++e) {
      const double x = f * ts[e];
      const double r = x - rint(x);                 // phase in cycles, |r| <= 1/2
      double s, c;
      sincospi(2.0 * r, &s, &c);
      const double wc = ws[e] * c;
      C += wc; S = __builtin_fma(ws[e], s, S);
      YC = __builtin_fma(wys[e], c, YC); YS = __builtin_fma(wys[e], s, YS);
      CC = __builtin_fma(wc, c, CC); CS = __builtin_fma(wc, s, CS);
    }
  }
  if (m >= nf) return;
  const double yy = wy[n];
  double SS = 1.0 - CC;
  if (fit_mean) { CC -= C * C; SS -= S * S; CS -= C * S; }
  const double D = CC * SS - CS * CS;
  power[(int64_t)b * nf + m] = (SS * YC * YC + CC * YS * YS - 2.0 * CS * YC * YS) / (yy * D);
}

// ---------------------------------------------------------------------------------------------------------------
// The FFT approximation (Press & Rybicki 1989) astropy's method='auto' resolves to on pgmuvi's grids (a regular grid of
// more than 200 frequencies): the reference's recorded outputs come from it, and it differs from the exact sums by up
// to 1e-2 in the power at the high-frequency end -- enough to reorder near-equal peaks.  Same floating-mean periodogram
// in its tau form; the three pairs of trigonometric sums
//     (Sh, Ch) = sum w y {sin, cos}(2 pi f t),   (S2, C2) = sum w {sin, cos}(4 pi f t),   (S, C) = sum w {sin, cos}(2 pi f t)
// on f_k = f0 + k df come from one inverse FFT each of the samples spread ("extirpolated") onto a regular grid of
// nfft = 2^ceil(log2(oversampling nf)) points with 4-point Lagrange weights (astropy's defaults: oversampling 5, M = 4).
// Kernels: prepare (weights, centring, t_min) -> spread (atomic adds into the three grids) -> log2(nfft) Stockham radix-2
// passes (three transforms per launch) -> combine.  Scratch per light curve: [w n | wy n | yy, tmin | 6 nfft complex].
// Every step is ONE launch for the whole batch (batch on gridDim.z / gridDim.x); the scratch is zeroed with one memset.
// The spread adds with fp64 atomics, so two runs agree to round-off only (the exact-sum path is bit-reproducible).
// ---------------------------------------------------------------------------------------------------------------
typedef double2 cplx;

__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
// exp(2 pi i x), argument reduced to |r| <= 1/2 first
__device__ __forceinline__ cplx cis2pi(double x) {
  const double r = x - rint(x);
  double s, c;
  sincospi(2.0 * r, &s, &c);
  return make_double2(c, s);
}

__host__ __device__ inline int64_t ls_fast_stride(int64_t n, int64_t nfft) { return 2 * n + 2 + 12 * nfft; }
inline int64_t ls_fast_nfft(int64_t nf, int oversampling) {
  int64_t nfft = 8;                                           // (the 4-point spread needs a few grid points)
  while (nfft < nf * oversampling) nfft <<= 1;
  return nfft;
}

__global__ __launch_bounds__(LS_THREADS) void k_ls_tmin(const double* __restrict__ tt, int64_t n, double* __restrict__ scratch,
                                                       int64_t stride) {
  const int b = blockIdx.x, t = threadIdx.x;
  __shared__ double red[LS_THREADS];
  double m = 1e300;
  for (int64_t i = t; i < n; i += LS_THREADS) { const double v = tt[(int64_t)b * n + i]; if (isfinite(v)) m = fmin(m, v); }   // (a NaN or an infinity among the times does not become the origin)
  red[t] = m;
  __syncthreads();
  for (int s = LS_THREADS / 2; s > 0; s >>= 1) { if (t < s) red[t] = fmin(red[t], red[t + s]); __syncthreads(); }
  if (t == 0) scratch[(int64_t)b * stride + 2 * n + 1] = (red[0] < 1e300) ? red[0] : 0.0;
}

// which = blockIdx.y: 0: h = w y, frequencies f;  1: h = w, frequencies 2 f;  2: h = w, frequencies f
__global__ __launch_bounds__(LS_THREADS) void k_ls_spread(const double* __restrict__ tt, double* __restrict__ scratch, int64_t n,
                                                         int64_t stride, double f0, double df, int64_t nfft) {
  const int b = blockIdx.z, which = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * LS_THREADS + threadIdx.x;
  if (i >= n) return;
  double* sc = scratch + (int64_t)b * stride;
  const double t0 = sc[2 * n + 1];
  const double fac = (which == 1) ? 2.0 : 1.0;
  const double h = (which == 0) ? sc[n + i] : sc[i];
  const double dt = tt[(int64_t)b * n + i] - t0;
  cplx hv = make_double2(h, 0.0);
  if (f0 > 0.0) hv = cmul(hv, cis2pi(fac * f0 * dt));
  double x = fmod(dt * (double)nfft * (fac * df), (double)nfft);
  double* grid = sc + 2 * n + 2 + (int64_t)which * 4 * nfft;          // (ping buffer of transform `which`; pong follows it)
  if (x == floor(x)) {
    const int64_t k = (int64_t)x;
    atomicAdd(grid + 2 * k, hv.x); atomicAdd(grid + 2 * k + 1, hv.y);
    return;
  }
  int64_t lo = (int64_t)(x - 2.0);
  if (lo < 0) lo = 0;
  if (lo > nfft - 4) lo = nfft - 4;
  const double d0 = x - (double)lo, d1 = d0 - 1.0, d2 = d0 - 2.0, d3 = d0 - 3.0;
  const double wgt[4] = {d1 * d2 * d3 * (-1.0 / 6.0), d0 * d2 * d3 * 0.5, d0 * d1 * d3 * (-0.5), d0 * d1 * d2 * (1.0 / 6.0)};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    atomicAdd(grid + 2 * (lo + j), hv.x * wgt[j]);
    atomicAdd(grid + 2 * (lo + j) + 1, hv.y * wgt[j]);
  }
}

// one Stockham radix-2 pass of the unnormalised inverse DFT (e^{+2 pi i k n / N}), sub-transform length p -> 2p
__global__ __launch_bounds__(LS_THREADS) void k_ls_fft_pass(double* __restrict__ scratch, int64_t n, int64_t stride, int64_t nfft,
                                                           int64_t p, int from_pong) {
  const int b = blockIdx.z, which = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * LS_THREADS + threadIdx.x;
  const int64_t half = nfft >> 1;
  if (i >= half) return;
  cplx* ping = reinterpret_cast<cplx*>(scratch + (int64_t)b * stride + 2 * n + 2 + (int64_t)which * 4 * nfft);
  cplx* pong = ping + nfft;
  const cplx* in = from_pong ? pong : ping;
  cplx* out = from_pong ? ping : pong;
  const int64_t k = i & (p - 1);
  const int64_t j = ((i - k) << 1) + k;
  double s, c;
  sincospi((double)k / (double)p, &s, &c);
  const cplx u0 = in[i], u1 = cmul(in[i + half], make_double2(c, s));
  out[j] = make_double2(u0.x + u1.x, u0.y + u1.y);
  out[j + p] = make_double2(u0.x - u1.x, u0.y - u1.y);
}

__global__ __launch_bounds__(LS_THREADS) void k_ls_combine(const double* __restrict__ scratch, int64_t n, int64_t stride, int64_t nfft,
                                                          int result_in_pong, double f0, double df, int64_t nf, int fit_mean,
                                                          double* __restrict__ power) {
  const int b = blockIdx.z;
  const int64_t k = (int64_t)blockIdx.x * LS_THREADS + threadIdx.x;
  if (k >= nf) return;
  const double* sc = scratch + (int64_t)b * stride;
  const double yy = sc[2 * n], t0 = sc[2 * n + 1];
  auto tr = [&](int which) -> cplx {
    const cplx* base = reinterpret_cast<const cplx*>(sc + 2 * n + 2 + (int64_t)which * 4 * nfft) + (result_in_pong ? nfft : 0);
    const double fac = (which == 1) ? 2.0 : 1.0;
    return cmul(base[k], cis2pi(t0 * fac * (f0 + df * (double)k)));
  };
  const cplx zh = tr(0), z2 = tr(1);
  const double Sh = zh.y, Ch = zh.x, S2 = z2.y, C2 = z2.x;
  double S = 0.0, C = 0.0, tan2;
  if (fit_mean) {
    const cplx z1 = tr(2);
    S = z1.y; C = z1.x;
    tan2 = (S2 - 2.0 * S * C) / (C2 - (C * C - S * S));
  } else {
    tan2 = S2 / C2;
  }
  const double C2w = 1.0 / sqrt(1.0 + tan2 * tan2), S2w = tan2 * C2w;
  const double Cw = sqrt(0.5) * sqrt(1.0 + C2w);
  const double Sw = sqrt(0.5) * ((S2w > 0.0) - (S2w < 0.0)) * sqrt(1.0 - C2w);
  const double YC = Ch * Cw + Sh * Sw, YS = Sh * Cw - Ch * Sw;
  double CC = 0.5 * (1.0 + C2 * C2w + S2 * S2w), SS = 0.5 * (1.0 - C2 * C2w - S2 * S2w);
  if (fit_mean) {
    const double a = C * Cw + S * Sw, bb = S * Cw - C * Sw;
    CC -= a * a; SS -= bb * bb;
  }
  power[(int64_t)b * nf + k] = (YC * YC / CC + YS * YS / SS) / yy;
}

}  // namespace

extern "C" int64_t pgm_lomb_scargle_fast_scratch_doubles(int64_t n, int64_t nf, int oversampling) {
  if (n < 1 || nf < 1 || oversampling < 1) return 0;
  return ls_fast_stride(n, ls_fast_nfft(nf, oversampling));
}

extern "C" int pgm_lomb_scargle_fast_f64(const double* t, const double* y, const double* dy, int64_t n, int batch,
                                         double f0, double df, int64_t nf, int fit_mean, int oversampling,
                                         double* scratch, double* power, void* stream) {
  if (!t) return -1;
  if (!y) return -2;
  if (n < 3) return -4;
  if (batch < 1) return -5;
  if (!(f0 >= 0.0)) return -6;
  if (!(df > 0.0)) return -7;
  if (nf < 1) return -8;
  if (oversampling < 1) return -10;
  if (!scratch) return -11;
  if (!power) return -12;
  const int64_t nfft = ls_fast_nfft(nf, oversampling);
  const int64_t stride = ls_fast_stride(n, nfft);
  hipStream_t st = (hipStream_t)stream;
  // the three grids start at zero (one memset of the whole scratch, records included), then weights, centred w y and yy by the
  // exact path's prepare kernel on this layout, one workgroup per light curve
  hipMemsetAsync(scratch, 0, sizeof(double) * (size_t)batch * (size_t)stride, st);
  hipLaunchKernelGGL(k_ls_prepare, dim3(batch), dim3(LS_THREADS), 0, st, y, dy, n, scratch, stride);
  hipLaunchKernelGGL(k_ls_tmin, dim3(batch), dim3(LS_THREADS), 0, st, t, n, scratch, stride);
  hipLaunchKernelGGL(k_ls_spread, dim3((unsigned)((n + LS_THREADS - 1) / LS_THREADS), 3, batch), dim3(LS_THREADS), 0, st,
                     t, scratch, n, stride, f0, df, nfft);
  int from_pong = 0;
  for (int64_t p = 1; p < nfft; p <<= 1) {
    hipLaunchKernelGGL(k_ls_fft_pass, dim3((unsigned)((nfft / 2 + LS_THREADS - 1) / LS_THREADS), 3, batch), dim3(LS_THREADS), 0, st,
                       scratch, n, stride, nfft, p, from_pong);
    from_pong ^= 1;
  }
  hipLaunchKernelGGL(k_ls_combine, dim3((unsigned)((nf + LS_THREADS - 1) / LS_THREADS), 1, batch), dim3(LS_THREADS), 0, st,
                     scratch, n, stride, nfft, from_pong, f0, df, nf, fit_mean, power);
  return hipGetLastError() == hipSuccess ? 0 : -99;
}

extern "C" int pgm_lomb_scargle_f64(const double* t, const double* y, const double* dy, int64_t n, int batch,
                                    const double* freq, int64_t nf, int fit_mean, double* scratch, double* power,
                                    void* stream) {
  if (!t) return -1;
  if (!y) return -2;
  if (n < 3) return -4;
  if (batch < 1) return -5;
  if (!freq) return -6;
  if (nf < 1) return -7;
  if (!scratch) return -9;
  if (!power) return -10;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_ls_prepare, dim3(batch), dim3(LS_THREADS), 0, st, y, dy, n, scratch, 2 * n + 1);
  hipLaunchKernelGGL(k_lomb_scargle, dim3((unsigned)((nf + LS_THREADS - 1) / LS_THREADS), 1, batch), dim3(LS_THREADS), 0, st,
                     t, scratch, n, freq, nf, fit_mean, power);
  return hipGetLastError() == hipSuccess ? 0 : -99;
}
