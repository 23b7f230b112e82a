// evalloop.cpp -- runs a few evaluations of B N-point light curves (for rocprofv3 timelines).  tools/evalloop [n] [reps] [need_grad] [q] [batch] [d]
// (d = 2: config 4's shape -- x = (time, wavelength of one of 8 bands), the 2-D spectral-mixture kernel, product over the dimensions)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <vector>
#include "../include/pgmuvi_hip.h"
int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 4096, reps = argc > 2 ? atoi(argv[2]) : 5, ng = argc > 3 ? atoi(argv[3]) : 1, q = argc > 4 ? atoi(argv[4]) : 4;
  const int B = argc > 5 ? atoi(argv[5]) : 1;
  const int d = argc > 6 ? atoi(argv[6]) : 1;
  pgm_ws* ws; if (pgm_workspace_create(&ws, 0, n, q > 0 ? q : 6, d, B)) return 1;
  std::vector<double> x((size_t)n * B * d), y((size_t)n * B), m((size_t)n * B, 0.0), nz((size_t)n * B, 0.01), w((size_t)q * B), mu((size_t)q * B * d), v((size_t)q * B * d);
  for (int b = 0; b < B; ++b)
    for (int a = 0; a < q; ++a) {
      w[b * q + a] = 0.5 / (a + 1);
      mu[(b * q + a) * d] = 1.0 / (150.0 - 13.0 * a + b); v[(b * q + a) * d] = mu[(b * q + a) * d] / 10.0;
      if (d == 2) { mu[(b * q + a) * d + 1] = 0.5; v[(b * q + a) * d + 1] = 0.3; }       // (the example's wavelength dimension: examples/2d_multiwavelength_example.py:83-91)
    }
  for (int b = 0; b < B; ++b)
    for (int i = 0; i < n; ++i) {
      const size_t e = (size_t)b * n + i;
      const double t = 3450.0 * i / n + 0.3 * sin(i + b);
      x[e * d] = t; if (d == 2) x[e * d + 1] = 0.45 + 0.25 * (i % 8);                  // 8 bands between 0.45 and 2.2
      y[e] = sin(t / (20 + b)) * (d == 2 ? 1.0 + 0.3 * (i % 8) : 1.0) + 0.1 * cos(i * 0.7);
    }
  auto dev = [](std::vector<double>& h) { double* p; hipMalloc((void**)&p, 8 * h.size()); hipMemcpy(p, h.data(), 8 * h.size(), hipMemcpyHostToDevice); return p; };
  double *dx = dev(x), *dy = dev(y), *dm = dev(m), *dn = dev(nz), *dw = dev(w), *dmu = dev(mu), *dv = dev(v);
  double* out; hipMalloc((void**)&out, 8 * (64 + 3 * (size_t)n) * B); int* info; hipMalloc((void**)&info, 4 * B);
  hipStream_t st; hipStreamCreate(&st);
  if (q <= 0) {
    // q = 0: a composed stationary kernel instead of the spectral mixture -- ScaleKernel(Periodic * RBF) + ScaleKernel(RBF),
    // the reference's PeriodicPlusStochasticGPModel (pgm_mll_kernel_value_grad_f64)
    pgm_workspace_destroy(ws);
    if (pgm_workspace_create(&ws, 0, n, 6, 1, 1)) return 1;
    pgm_kernel_program G{};
    G.nleaf = 3; G.nterm = 2; G.nparam = 6;
    G.kind[0] = 5; G.dims[0] = 1; G.par[0] = 0;      // periodic [p, lambda]
    G.kind[1] = 1; G.dims[1] = 1; G.par[1] = 2;      // rbf [l]
    G.kind[2] = 1; G.dims[2] = 1; G.par[2] = 3;      // rbf [l]
    G.tmask[0] = 3; G.tnscale[0] = 1; G.tscale[0][0] = 4;
    G.tmask[1] = 4; G.tnscale[1] = 1; G.tscale[1][0] = 5;
    std::vector<double> th = {150.0, 1.3, 750.0, 20.0, 0.8, 0.2};
    double* dth = dev(th);
    auto rung = [&]() { pgm_mll_kernel_value_grad_f64(ws, 1, dx, dy, dm, dn, nullptr, n, 1, &G, dth, 0.0, ng, out, out + 1, out + 16, out + 16 + n, info, st); };
    rung(); hipStreamSynchronize(st);
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) rung();
    hipStreamSynchronize(st);
    double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / reps;
    std::vector<double> h(16);
    hipMemcpy(h.data(), out, 8 * 16, hipMemcpyDeviceToHost);
    printf("n=%d generic kernel (periodic x rbf + rbf) need_grad=%d: %.3f ms/eval  mll=%.12f  g_theta = %.6e %.6e %.6e %.6e %.6e %.6e\n", n, ng, ms, h[0], h[1], h[2], h[3], h[4], h[5], h[6]);
    return 0;
  }
  // batch 1: the layout of old (one block of 16 + 3n doubles); batches: [mll B | g_w | g_mu | g_v | g_noise | g_mean]
  double* o = out;
  auto run = [&]() {
    if (B == 1) pgm_mll_value_grad_f64(ws, dx, dy, dm, dn, 0, n, d, dw, dmu, dv, q, 0, 0, ng, out, out + 1, out + 1 + q, out + 1 + q + q * d, out + 64, out + 64 + n, info, st);
    else pgm_mll_value_grad_batched_f64(ws, B, dx, dy, dm, dn, nullptr, n, d, dw, dmu, dv, q, 0, 0, ng, o, o + B, o + B + (size_t)B * q, o + B + (size_t)B * q * (1 + d),
                                        o + B + (size_t)B * q * (1 + 2 * d), o + B + (size_t)B * q * (1 + 2 * d) + (size_t)B * n, info, st);
  };
  run(); hipStreamSynchronize(st);
  auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < reps; ++r) run();
  hipStreamSynchronize(st);
  double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / reps;
  std::vector<double> h((64 + 3 * (size_t)n) * (size_t)B);       // (= the allocation of `out`)
  hipMemcpy(h.data(), out, 8 * h.size(), hipMemcpyDeviceToHost);
  double gs = 0.0, gn = 0.0;                                   // gradient fingerprints: hyper-parameters, per-point noise
  for (int a = 1; a < 1 + q * (1 + 2 * d) && ng; ++a) gs += h[(size_t)a] * (1.0 + 0.1 * a);
  for (int i = 0; i < n && ng; ++i) gn += h[64 + (size_t)i] * (1.0 + 1e-3 * (i % 97));
  if (B > 1) { printf("n=%d batch=%d need_grad=%d: %.3f ms/call = %.1f evals/s  mll[0]=%.12f mll[B-1]=%.12f\n", n, B, ng, ms, B / ms * 1e3, h[0], h[B - 1]); return 0; }
  printf("n=%d need_grad=%d: %.3f ms/eval  mll=%.12f  gsum=%.12e  gnoise=%.12e\n", n, ng, ms, h[0], gs, gn);
  return 0;
}
