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

// scratch layout per light curve: w[n], wy[n], yy
__global__ __launch_bounds__(LS_THREADS) void k_ls_prepare(const double* __restrict__ y, const double* __restrict__ dy,
                                                           int64_t n, double* __restrict__ scratch) {
  const int b = blockIdx.x, t = threadIdx.x;
  const double* yb = y + (int64_t)b * n;
  const double* db = dy ? dy + (int64_t)b * n : nullptr;
  double* w = scratch + (int64_t)b * (2 * n + 1);
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

}  // namespace

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
  hipLaunchKernelGGL(k_ls_prepare, dim3(batch), dim3(LS_THREADS), 0, st, y, dy, n, scratch);
  hipLaunchKernelGGL(k_lomb_scargle, dim3((unsigned)((nf + LS_THREADS - 1) / LS_THREADS), 1, batch), dim3(LS_THREADS), 0, st,
                     t, scratch, n, freq, nf, fit_mean, power);
  return hipGetLastError() == hipSuccess ? 0 : -99;
}
