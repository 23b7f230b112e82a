// selftest.cpp -- standalone GPU check of libpgmuvi_hip.so against a naive CPU
// implementation written here (no torch, no oracle): fast bring-up diagnostics for
// the kernels.  Build: make -C tools ; run on the GPU box: tools/selftest [nmax]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../include/pgmuvi_hip.h"

using std::vector;
static const double PI = 3.14159265358979323846;

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

struct Problem {
  int n, d, q, order;
  vector<double> x, y, mean, noise, w, mu, v;
};

static Problem make_problem(int n, int d, int q, int order, unsigned seed) {
  std::mt19937_64 g(seed);
  std::uniform_real_distribution<double> U(0.0, 1.0);
  std::normal_distribution<double> Nrm(0.0, 1.0);
  Problem p; p.n = n; p.d = d; p.q = q; p.order = order;
  p.x.resize((size_t)n * d); p.y.resize(n); p.mean.resize(n); p.noise.resize(n);
  p.w.resize(q); p.mu.resize((size_t)q * d); p.v.resize((size_t)q * d);
  for (int i = 0; i < n; ++i) {
    for (int k = 0; k < d; ++k) p.x[(size_t)i * d + k] = (k == 0) ? 3450.0 * U(g) : 0.45 + 1.75 * U(g);
    p.mean[i] = 0.1;
    p.noise[i] = 0.01 * (0.5 + U(g));
  }
  for (int a = 0; a < q; ++a) {
    p.w[a] = 0.1 + 0.5 * U(g);
    for (int k = 0; k < d; ++k) {
      p.mu[(size_t)a * d + k] = (k == 0) ? 1.0 / (30.0 + 300.0 * U(g)) : 0.5 * U(g);
      p.v[(size_t)a * d + k] = (k == 0) ? p.mu[(size_t)a * d] / (5.0 + 10.0 * U(g)) : 0.1 + 0.3 * U(g);
    }
  }
  for (int i = 0; i < n; ++i) {
    double s = 0.0;
    for (int a = 0; a < q; ++a) s += std::sqrt(2 * p.w[a]) * std::sin(2 * PI * p.mu[(size_t)a * d] * p.x[(size_t)i * d] + a);
    p.y[i] = s + 0.1 * Nrm(g) + 0.1;
  }
  return p;
}

// ---- naive CPU reference ---------------------------------------------------
struct CpuOut { double mll; vector<double> gw, gmu, gv, gnoise, gmean; bool ok; };

static double kpair(const Problem& p, int i, int j, vector<double>* dparts = nullptr) {
  const int d = p.d, q = p.q;
  vector<double> E(q * d), C(q * d);
  for (int a = 0; a < q; ++a)
    for (int k = 0; k < d; ++k) {
      const double xi = p.x[(size_t)i * d + k], xj = p.x[(size_t)j * d + k];
      const double ds = xi * p.v[a * d + k] - xj * p.v[a * d + k];
      E[a * d + k] = std::exp(-2 * PI * PI * ds * ds);
      C[a * d + k] = std::cos(2 * PI * (xi * p.mu[a * d + k] - xj * p.mu[a * d + k]));
    }
  double K;
  if (p.order == 0) { K = 1; for (int k = 0; k < d; ++k) { double S = 0; for (int a = 0; a < q; ++a) S += p.w[a] * E[a * d + k] * C[a * d + k]; K *= S; } }
  else { K = 0; for (int a = 0; a < q; ++a) { double pr = 1; for (int k = 0; k < d; ++k) pr *= E[a * d + k] * C[a * d + k]; K += p.w[a] * pr; } }
  (void)dparts;
  return K;
}

static CpuOut cpu_eval(const Problem& p, double jitter, double noise_scalar) {
  const int n = p.n, d = p.d, q = p.q;
  CpuOut o; o.ok = true;
  vector<double> A((size_t)n * n);
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) A[(size_t)i * n + j] = kpair(p, i, j) + (i == j ? p.noise[i] + jitter + noise_scalar : 0.0);
  vector<double> L = A;   // lower Cholesky in place
  for (int j = 0; j < n; ++j) {
    double s = L[(size_t)j * n + j];
    for (int k = 0; k < j; ++k) s -= L[(size_t)j * n + k] * L[(size_t)j * n + k];
    if (!(s > 0)) { o.ok = false; return o; }
    const double ljj = std::sqrt(s); L[(size_t)j * n + j] = ljj;
    for (int i = j + 1; i < n; ++i) {
      double t = L[(size_t)i * n + j];
      for (int k = 0; k < j; ++k) t -= L[(size_t)i * n + k] * L[(size_t)j * n + k];
      L[(size_t)i * n + j] = t / ljj;
    }
  }
  vector<double> r(n), z(n), al(n);
  for (int i = 0; i < n; ++i) r[i] = p.y[i] - p.mean[i];
  double logdet = 0, zz = 0;
  for (int i = 0; i < n; ++i) { double t = r[i]; for (int k = 0; k < i; ++k) t -= L[(size_t)i * n + k] * z[k]; z[i] = t / L[(size_t)i * n + i]; zz += z[i] * z[i]; logdet += 2 * std::log(L[(size_t)i * n + i]); }
  for (int i = n - 1; i >= 0; --i) { double t = z[i]; for (int k = i + 1; k < n; ++k) t -= L[(size_t)k * n + i] * al[k]; al[i] = t / L[(size_t)i * n + i]; }
  o.mll = -0.5 * (zz + logdet + n * std::log(2 * PI)) / n;
  // inverse: Linv then Ainv = Linv^T Linv
  vector<double> Li((size_t)n * n, 0.0);
  for (int c = 0; c < n; ++c) {
    for (int i = c; i < n; ++i) { double t = (i == c) ? 1.0 : 0.0; for (int k = c; k < i; ++k) t -= L[(size_t)i * n + k] * Li[(size_t)k * n + c]; Li[(size_t)i * n + c] = t / L[(size_t)i * n + i]; }
  }
  o.gw.assign(q, 0); o.gmu.assign(q * d, 0); o.gv.assign(q * d, 0); o.gnoise.assign(n, 0); o.gmean.assign(n, 0);
  const double hn = 0.5 / n;
  vector<double> E(q * d), C(q * d), S(q * d), TAU(d), Sd(d);
  for (int i = 0; i < n; ++i) {
    o.gmean[i] = al[i] / n;
    for (int j = 0; j < n; ++j) {
      double ainv = 0; for (int k = (i > j ? i : j); k < n; ++k) ainv += Li[(size_t)k * n + i] * Li[(size_t)k * n + j];
      const double G = al[i] * al[j] - ainv;
      if (i == j) o.gnoise[i] = hn * G;
      for (int k = 0; k < d; ++k) { TAU[k] = p.x[(size_t)i * d + k] - p.x[(size_t)j * d + k]; Sd[k] = 0; }
      for (int a = 0; a < q; ++a) for (int k = 0; k < d; ++k) {
        const double xi = p.x[(size_t)i * d + k], xj = p.x[(size_t)j * d + k];
        const double ds = xi * p.v[a * d + k] - xj * p.v[a * d + k];
        const double ang = 2 * PI * (xi * p.mu[a * d + k] - xj * p.mu[a * d + k]);
        E[a * d + k] = std::exp(-2 * PI * PI * ds * ds); C[a * d + k] = std::cos(ang); S[a * d + k] = std::sin(ang);
        Sd[k] += p.w[a] * E[a * d + k] * C[a * d + k];
      }
      for (int a = 0; a < q; ++a) {
        double pall = 1; for (int k = 0; k < d; ++k) pall *= E[a * d + k] * C[a * d + k];
        if (p.order != 0) o.gw[a] += hn * G * pall;
        for (int k = 0; k < d; ++k) {
          double oth = 1;
          for (int k2 = 0; k2 < d; ++k2) if (k2 != k) oth *= (p.order == 0) ? Sd[k2] : E[a * d + k2] * C[a * d + k2];
          if (p.order == 0) o.gw[a] += hn * G * oth * E[a * d + k] * C[a * d + k];
          o.gmu[a * d + k] += hn * G * oth * (-2 * PI * TAU[k] * p.w[a] * E[a * d + k] * S[a * d + k]);
          o.gv[a * d + k] += hn * G * oth * (-4 * PI * PI * p.v[a * d + k] * TAU[k] * TAU[k] * p.w[a] * E[a * d + k] * C[a * d + k]);
        }
      }
    }
  }
  return o;
}

// ---- device helpers --------------------------------------------------------
template <class T> static T* dev(const vector<T>& h) { T* p; HIPCHK(hipMalloc((void**)&p, sizeof(T) * (h.size() ? h.size() : 1))); if (h.size()) HIPCHK(hipMemcpy(p, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice)); return p; }
template <class T> static vector<T> host(const T* p, size_t n) { vector<T> h(n); HIPCHK(hipMemcpy(h.data(), p, sizeof(T) * n, hipMemcpyDeviceToHost)); return h; }
static double relerr(const vector<double>& a, const vector<double>& b) {
  double num = 0, den = 0; for (size_t i = 0; i < a.size(); ++i) { num = std::fmax(num, std::fabs(a[i] - b[i])); den = std::fmax(den, std::fabs(b[i])); }
  return num / (den + 1e-300);
}

struct GpuOut { double mll; vector<double> gw, gmu, gv, gnoise, gmean; int info; };

static GpuOut gpu_eval(pgm_ws* ws, const Problem& p, double jitter, double noise_scalar, int need_grad) {
  double *x = dev(p.x), *y = dev(p.y), *m = dev(p.mean), *nz = dev(p.noise), *w = dev(p.w), *mu = dev(p.mu), *v = dev(p.v);
  vector<double> z1(1), zq(p.q), zqd(p.q * p.d), zn(p.n); vector<int> zi(1);
  double *mll = dev(z1), *gw = dev(zq), *gmu = dev(zqd), *gv = dev(zqd), *gn = dev(zn), *gm = dev(zn); int* info = dev(zi);
  int rc = pgm_mll_value_grad_f64(ws, x, y, m, nz, noise_scalar, p.n, p.d, w, mu, v, p.q, p.order, jitter, need_grad, mll, gw, gmu, gv, gn, gm, info, nullptr);
  HIPCHK(hipDeviceSynchronize());
  if (rc != 0) printf("  pgm_mll_value_grad_f64 rc=%d\n", rc);
  GpuOut o; o.mll = host(mll, 1)[0]; o.gw = host(gw, p.q); o.gmu = host(gmu, p.q * p.d); o.gv = host(gv, p.q * p.d);
  o.gnoise = host(gn, p.n); o.gmean = host(gm, p.n); o.info = host(info, 1)[0];
  for (void* q : {(void*)x, (void*)y, (void*)m, (void*)nz, (void*)w, (void*)mu, (void*)v, (void*)mll, (void*)gw, (void*)gmu, (void*)gv, (void*)gn, (void*)gm, (void*)info}) HIPCHK(hipFree(q));
  return o;
}

static int check(const char* name, pgm_ws* ws, const Problem& p, double noise_scalar = 0.0) {
  CpuOut c = cpu_eval(p, 0.0, noise_scalar);
  GpuOut g = gpu_eval(ws, p, 0.0, noise_scalar, 1);
  GpuOut g0 = gpu_eval(ws, p, 0.0, noise_scalar, 0);
  const double dm = std::fabs(c.mll - g.mll), dm0 = std::fabs(c.mll - g0.mll);
  const double ew = relerr(g.gw, c.gw), emu = relerr(g.gmu, c.gmu), ev = relerr(g.gv, c.gv), en = relerr(g.gnoise, c.gnoise), em = relerr(g.gmean, c.gmean);
  const bool ok = c.ok && g.info == 0 && dm < 1e-9 && dm0 < 1e-9 && ew < 1e-7 && emu < 1e-7 && ev < 1e-7 && en < 1e-7 && em < 1e-7;
  printf("%-28s n=%5d d=%d q=%d ord=%d  mll=% .12f |dmll|=%.2e (value-only %.2e)  rel: w %.1e mu %.1e v %.1e noise %.1e mean %.1e info=%d  %s\n",
         name, p.n, p.d, p.q, p.order, g.mll, dm, dm0, ew, emu, ev, en, em, g.info, ok ? "OK" : "FAIL");
  return ok ? 0 : 1;
}

int main(int argc, char** argv) {
  int nmax = argc > 1 ? atoi(argv[1]) : 4096;
  int fails = 0;
  printf("%s\n", pgm_version());
  double tf = 0; pgm_probe_mfma_f64(0, &tf);
  printf("fp64 MFMA issue probe: %.1f TFLOP/s\n", tf);

  pgm_ws* ws = nullptr;
  int rc = pgm_workspace_create(&ws, 0, 1024, 8, 2, 4);
  if (rc) { printf("workspace_create rc=%d\n", rc); return 2; }

  // dense kernel entry point
  {
    Problem p = make_problem(150, 2, 3, 0, 7);
    double *x = dev(p.x), *w = dev(p.w), *mu = dev(p.mu), *v = dev(p.v), *nz = dev(p.noise);
    vector<double> Kh((size_t)150 * 150); double* K = dev(Kh);
    for (int order = 0; order < 2; ++order) {
      p.order = order;
      rc = pgm_sm_kernel_f64(x, 150, x, 150, 2, w, mu, v, 3, nz, 0.25, order, K, 150, nullptr);
      HIPCHK(hipDeviceSynchronize());
      Kh = host(K, Kh.size());
      double e = 0; for (int i = 0; i < 150; ++i) for (int j = 0; j < 150; ++j) e = std::fmax(e, std::fabs(Kh[(size_t)i * 150 + j] - (kpair(p, i, j) + (i == j ? p.noise[i] + 0.25 : 0))));
      printf("pgm_sm_kernel_f64 order %d rc=%d max|err|=%.2e %s\n", order, rc, e, e < 1e-12 ? "OK" : "FAIL");
      fails += !(e < 1e-12);
    }
  }
  fails += check("1-D single block", ws, make_problem(64, 1, 4, 0, 1));
  fails += check("1-D one full block", ws, make_problem(128, 1, 1, 0, 2));
  fails += check("1-D ragged (200)", ws, make_problem(200, 1, 4, 0, 3));
  fails += check("1-D 3 blocks", ws, make_problem(384, 1, 4, 0, 4));
  fails += check("1-D ragged (700)", ws, make_problem(700, 1, 3, 0, 5));
  fails += check("2-D prod-of-sums", ws, make_problem(300, 2, 3, 0, 6));
  fails += check("2-D sum-of-prods", ws, make_problem(300, 2, 3, 1, 6));
  fails += check("scalar noise", ws, make_problem(257, 1, 2, 0, 8), 0.05);

  // batched == singles
  {
    const int B = 3, n = 300, q = 4;
    vector<Problem> ps; for (int b = 0; b < B; ++b) ps.push_back(make_problem(n, 1, q, 0, 100 + b));
    vector<double> X, Y, M, NZ, W, MU, V;
    for (auto& p : ps) { X.insert(X.end(), p.x.begin(), p.x.end()); Y.insert(Y.end(), p.y.begin(), p.y.end()); M.insert(M.end(), p.mean.begin(), p.mean.end());
      NZ.insert(NZ.end(), p.noise.begin(), p.noise.end()); W.insert(W.end(), p.w.begin(), p.w.end()); MU.insert(MU.end(), p.mu.begin(), p.mu.end()); V.insert(V.end(), p.v.begin(), p.v.end()); }
    double *x = dev(X), *y = dev(Y), *m = dev(M), *nz = dev(NZ), *w = dev(W), *mu = dev(MU), *v = dev(V);
    vector<double> zb(B), zq(B * q), zn((size_t)B * n); vector<int> zi(B);
    double *mll = dev(zb), *gw = dev(zq), *gmu = dev(zq), *gv = dev(zq), *gn = dev(zn), *gm = dev(zn); int* info = dev(zi);
    rc = pgm_mll_value_grad_batched_f64(ws, B, x, y, m, nz, nullptr, n, 1, w, mu, v, q, 0, 0.0, 1, mll, gw, gmu, gv, gn, gm, info, nullptr);
    HIPCHK(hipDeviceSynchronize());
    auto hm = host(mll, B); auto hgw = host(gw, B * q); auto hgmu = host(gmu, B * q);
    for (int b = 0; b < B; ++b) {
      GpuOut g = gpu_eval(ws, ps[b], 0.0, 0.0, 1);
      // (the value is the same whoever shares the call; the gradient sums are formed per work item, and a call's work items
      //  depend on how many light curves it has: equal to rounding, relative to the gradient's size)
      double e = std::fabs(g.mll - hm[b]), scale = 1.0;
      for (int a = 0; a < q; ++a) scale = std::fmax(scale, std::fmax(std::fabs(g.gw[a]), std::fabs(g.gmu[a])));
      for (int a = 0; a < q; ++a) e = std::fmax(e, (std::fabs(g.gw[a] - hgw[b * q + a]) + std::fabs(g.gmu[a] - hgmu[b * q + a])) / scale);
      printf("batched[%d] vs single: rc=%d max diff %.2e (relative to the largest gradient) %s\n", b, rc, e, e == 0.0 ? "OK(bitwise)" : (e < 1e-12 ? "OK" : "FAIL"));
      fails += !(e < 1e-12);
    }
  }
  // ragged == singles: light curves of different lengths in one call (pgm_mll_value_grad_ragged_f64), padded to a common pitch
  {
    const int B = 5, q = 4; const int lens[B] = {300, 70, 420, 129, 300};
    const int64_t S = 448;
    vector<Problem> ps; for (int b = 0; b < B; ++b) ps.push_back(make_problem(lens[b], 1, q, 0, 200 + b));
    vector<double> X((size_t)B * S), Y((size_t)B * S), M((size_t)B * S), NZ((size_t)B * S), W, MU, V;
    vector<int64_t> nh(B);
    for (int b = 0; b < B; ++b) {
      nh[b] = lens[b];
      for (int i = 0; i < lens[b]; ++i) { X[b * S + i] = ps[b].x[i]; Y[b * S + i] = ps[b].y[i]; M[b * S + i] = ps[b].mean[i]; NZ[b * S + i] = ps[b].noise[i]; }
      W.insert(W.end(), ps[b].w.begin(), ps[b].w.end()); MU.insert(MU.end(), ps[b].mu.begin(), ps[b].mu.end()); V.insert(V.end(), ps[b].v.begin(), ps[b].v.end());
    }
    double *x = dev(X), *y = dev(Y), *m = dev(M), *nz = dev(NZ), *w = dev(W), *mu = dev(MU), *v = dev(V);
    vector<double> zb(B), zq(B * q), zn((size_t)B * S); vector<int> zi(B);
    double *mll = dev(zb), *gw = dev(zq), *gmu = dev(zq), *gv = dev(zq), *gn = dev(zn), *gm = dev(zn); int* info = dev(zi);
    vector<int> set_of(B), nb_of(B);
    const int nsets = pgm_ragged_plan(nh.data(), B, 64, set_of.data(), nb_of.data());
    rc = pgm_mll_value_grad_ragged_f64(ws, B, x, y, m, nz, nullptr, nh.data(), S, 1, w, mu, v, q, 0, 0.0, 1, mll, gw, gmu, gv, gn, gm, info, nullptr);
    HIPCHK(hipDeviceSynchronize());
    auto hm = host(mll, B); auto hgw = host(gw, B * q); auto hgn = host(gn, (size_t)B * S);
    for (int b = 0; b < B; ++b) {
      GpuOut g = gpu_eval(ws, ps[b], 0.0, 0.0, 1);
      const double ev = std::fabs(g.mll - hm[b]);
      double eg = 0;
      for (int a = 0; a < q; ++a) eg = std::fmax(eg, std::fabs(g.gw[a] - hgw[b * q + a]) / (std::fabs(g.gw[a]) + 1e-300));
      for (int i = 0; i < lens[b]; ++i) eg = std::fmax(eg, std::fabs(g.gnoise[i] - hgn[b * S + i]) / (std::fabs(g.gnoise[i]) + 1e-9));
      printf("ragged[%d] n=%3d (set %d of %d, %d block rows) vs single: rc=%d |dmll| %.2e %s, gradients rel %.1e %s\n", b, lens[b], set_of[b], nsets, nb_of[set_of[b]],
             rc, ev, ev == 0.0 ? "OK(bitwise)" : "FAIL", eg, eg < 1e-9 ? "OK" : "FAIL");
      fails += !(ev == 0.0) + !(eg < 1e-9);
    }
  }
  // the sampler's potential on the device (pgm_pot_*): U and dU/dz for two chains against the same formula assembled here from the
  // batched evaluation at theta = (c, exp z): U = -(N mll + sum log Normal(z; loc, scale)), dU/dz by the chain rule
  {
    const int B = 2, n = 300, q = 3, P = 1 + 3 * q;
    vector<Problem> ps; for (int b = 0; b < B; ++b) ps.push_back(make_problem(n, 1, q, 0, 300 + b));
    vector<double> X, Y, NZ, Z((size_t)B * P), LOC((size_t)B * P), SC((size_t)B * P), W, MU, V, M;
    for (int b = 0; b < B; ++b) {
      auto& p = ps[b];
      X.insert(X.end(), p.x.begin(), p.x.end()); Y.insert(Y.end(), p.y.begin(), p.y.end()); NZ.insert(NZ.end(), p.noise.begin(), p.noise.end());
      double* z = &Z[(size_t)b * P];
      z[0] = 0.07 - 0.1 * b;
      for (int a = 0; a < q; ++a) { z[1 + a] = std::log(p.w[a]); z[1 + q + a] = std::log(p.mu[a]); z[1 + 2 * q + a] = std::log(p.v[a]); }
      for (int e = 0; e < P; ++e) { LOC[(size_t)b * P + e] = e == 0 ? 0.1 : -1.0 - 0.3 * e; SC[(size_t)b * P + e] = e == 0 ? 0.5 : 1.5 + 0.1 * e; }
      W.insert(W.end(), p.w.begin(), p.w.end()); MU.insert(MU.end(), p.mu.begin(), p.mu.end()); V.insert(V.end(), p.v.begin(), p.v.end());
      for (int i = 0; i < n; ++i) M.push_back(z[0]);
    }
    double *x = dev(X), *y = dev(Y), *nz = dev(NZ), *w = dev(W), *mu = dev(MU), *v = dev(V), *m = dev(M);
    vector<double> zb(B), zq(B * q), zn((size_t)B * n); vector<int> zi(B);
    double *mll = dev(zb), *gw = dev(zq), *gmu = dev(zq), *gv = dev(zq), *gn = dev(zn), *gm = dev(zn); int* info = dev(zi);
    rc = pgm_mll_value_grad_batched_f64(ws, B, x, y, m, nz, nullptr, n, 1, w, mu, v, q, 0, 0.0, 1, mll, gw, gmu, gv, gn, gm, info, nullptr);
    HIPCHK(hipDeviceSynchronize());
    auto hm = host(mll, B); auto hgw = host(gw, B * q); auto hgmu = host(gmu, B * q); auto hgv = host(gv, B * q); auto hgm = host(gm, (size_t)B * n);
    pgm_pot* pot = nullptr;
    int rcp = pgm_pot_create(&pot, ws, B, x, y, nz, n, 1, q, 0, LOC.data(), SC.data());
    vector<double> U(B), G((size_t)B * P); vector<int> inf(B);
    for (int rep = 0; rep < 2 && rcp == 0; ++rep) rcp = pgm_pot_eval(pot, Z.data(), U.data(), G.data(), inf.data(), nullptr);
    for (int b = 0; b < B; ++b) {
      const double* z = &Z[(size_t)b * P];
      double lp = 0, gsum = 0;
      for (int i = 0; i < n; ++i) gsum += hgm[(size_t)b * n + i];
      vector<double> gref(P);
      for (int e = 0; e < P; ++e) {
        const double t = (z[e] - LOC[(size_t)b * P + e]) / SC[(size_t)b * P + e];
        lp += -0.5 * t * t - std::log(SC[(size_t)b * P + e]) - 0.91893853320467274178;
        double gth = e == 0 ? gsum : (e < 1 + q ? hgw[b * q + e - 1] : e < 1 + 2 * q ? hgmu[b * q + e - 1 - q] : hgv[b * q + e - 1 - 2 * q]);
        if (e > 0) gth *= std::exp(z[e]);
        gref[e] = -(n * gth - t / SC[(size_t)b * P + e]);
      }
      const double Uref = -(n * hm[b] + lp);
      double eg = 0;
      for (int e = 0; e < P; ++e) eg = std::fmax(eg, std::fabs(G[(size_t)b * P + e] - gref[e]) / (std::fabs(gref[e]) + 1e-9));
      const double eu = std::fabs(U[b] - Uref) / (std::fabs(Uref) + 1.0);
      printf("potential[%d]: rc=%d U=%.9f |dU| rel %.1e %s, dU/dz rel %.1e %s, info %d\n", b, rcp, U[b], eu, eu < 1e-12 ? "OK" : "FAIL", eg, eg < 1e-9 ? "OK" : "FAIL", inf[b]);
      fails += !(rcp == 0 && eu < 1e-12) + !(eg < 1e-9);
    }
    if (pot) pgm_pot_destroy(pot);
  }
  // non-PD detection
  {
    Problem p = make_problem(200, 1, 2, 0, 9);
    for (auto& nzv : p.noise) nzv = -5.0;
    GpuOut g = gpu_eval(ws, p, 0.0, 0.0, 1);
    printf("non-PD: info=%d mll=%f %s\n", g.info, g.mll, (g.info > 0 && std::isnan(g.mll)) ? "OK" : "FAIL");
    fails += !(g.info > 0);
  }
  pgm_workspace_destroy(ws);

  // timing at scale
  for (int n : {1024, 2048, 4096}) {
    if (n > nmax) break;
    rc = pgm_workspace_create(&ws, 0, n, 4, 1, 1);
    if (rc) { printf("workspace_create(%d) rc=%d\n", n, rc); return 2; }
    Problem p = make_problem(n, 1, 4, 0, 11);
    double *x = dev(p.x), *y = dev(p.y), *m = dev(p.mean), *nz = dev(p.noise), *w = dev(p.w), *mu = dev(p.mu), *v = dev(p.v);
    vector<double> z1(1), zq(4), zn(n); vector<int> zi(1);
    double *mll = dev(z1), *gw = dev(zq), *gmu = dev(zq), *gv = dev(zq), *gn = dev(zn), *gm = dev(zn); int* info = dev(zi);
    for (int need_grad = 1; need_grad >= 0; --need_grad) {
      for (int it = 0; it < 3; ++it) pgm_mll_value_grad_f64(ws, x, y, m, nz, 0, n, 1, w, mu, v, 4, 0, 0, need_grad, mll, gw, gmu, gv, gn, gm, info, nullptr);
      HIPCHK(hipDeviceSynchronize());
      const int reps = 10;
      auto t0 = std::chrono::steady_clock::now();
      for (int it = 0; it < reps; ++it) pgm_mll_value_grad_f64(ws, x, y, m, nz, 0, n, 1, w, mu, v, 4, 0, 0, need_grad, mll, gw, gmu, gv, gn, gm, info, nullptr);
      HIPCHK(hipDeviceSynchronize());
      double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / reps;
      const double flops = need_grad ? (double)n * n * n : (double)n * n * n / 3;
      printf("n=%d need_grad=%d: %.3f ms/eval (%.1f evals/s, %.2f TFLOP/s algorithmic) mll=%.9f info=%d\n", n, need_grad, ms, 1e3 / ms, flops / ms / 1e9, host(mll, 1)[0], host(info, 1)[0]);
    }
    pgm_profile_enable(ws, 1);
    for (int it = 0; it < 5; ++it) pgm_mll_value_grad_f64(ws, x, y, m, nz, 0, n, 1, w, mu, v, 4, 0, 0, 1, mll, gw, gmu, gv, gn, gm, info, nullptr);
    double ms[16]; int64_t cnt[16]; pgm_profile_read(ws, ms, cnt);
    pgm_profile_enable(ws, 0);
    for (int ph = 0; ph < pgm_profile_phases(); ++ph) printf("   phase %-16s %8.3f ms/eval  (%lld launches/eval)\n", pgm_profile_phase_name(ph), ms[ph] / 5, (long long)cnt[ph] / 5);
    pgm_workspace_destroy(ws);
  }
  printf("selftest: %d failure(s)\n", fails);
  return fails ? 1 : 0;
}
