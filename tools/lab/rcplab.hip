// tools/lab/rcplab.hip -- accuracy of the fp64 reciprocal / reciprocal-square-root seeds plus Newton steps.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double* d, double* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double x = d[i];
  const double y0 = __builtin_amdgcn_rcp(x);
  const double e0 = __builtin_fma(-x, y0, 1.0);
  const double y1 = __builtin_fma(y0, e0, y0);
  const double e1 = __builtin_fma(-x, y1, 1.0);
  const double y2 = __builtin_fma(y1, e1, y1);
  const double r0 = __builtin_amdgcn_rsq(x);
  const double f0 = __builtin_fma(-x * r0, r0, 1.0);
  const double r1 = __builtin_fma(0.5 * r0, f0, r0);
  out[i * 5 + 0] = y0; out[i * 5 + 1] = y1; out[i * 5 + 2] = y2; out[i * 5 + 3] = r0; out[i * 5 + 4] = r1;
}
int main() {
  const int n = 1 << 20;
  std::vector<double> h(n), o((size_t)n * 5);
  for (int i = 0; i < n; ++i) h[i] = std::exp(-20.0 + 40.0 * (double)((i * 2654435761u) % 1000003) / 1000003.0);
  double *d, *out; hipMalloc(&d, n * 8); hipMalloc(&out, (size_t)n * 40);
  hipMemcpy(d, h.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, out, n);
  hipMemcpy(o.data(), out, (size_t)n * 40, hipMemcpyDeviceToHost);
  double m[5] = {0, 0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
    const long double x = h[i];
    const long double t[5] = {1.0L / x, 1.0L / x, 1.0L / x, 1.0L / sqrtl(x), 1.0L / sqrtl(x)};
    for (int q = 0; q < 5; ++q) { const double e = (double)fabsl(((long double)o[(size_t)i * 5 + q] - t[q]) / t[q]); if (e > m[q]) m[q] = e; }
  }
  printf("max relative error: rcp seed %.3e, +1 Newton %.3e, +2 Newton %.3e; rsq seed %.3e, +1 Newton %.3e\n", m[0], m[1], m[2], m[3], m[4]);
  return 0;
}
