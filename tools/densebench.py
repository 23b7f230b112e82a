#!/usr/bin/env python3
"""The reference's non-spectral-mixture models: a quasi-periodic exact GP (ScaleKernel(Periodic * RBF), pgmuvi/gps.py:915-935)
at N points, model -> mll -> backward per iteration on the fused generic-kernel path (pgm_mll_kernel_value_grad_f64: build,
sweep and gradient contraction in the library) and, for comparison, on the dense back-end (kernel matrix and its pull-back
by torch, factorisation + dmll/dA by the library)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pgmuvi_amd import _hip, gpytorch as g, synthetic as syn
dev = torch.device("cuda:0")
K = g.kernels
for n in (1024, 4096):
    t, y, e = syn.cfg2(n_obs=n)
    x, yy, nz = t.double().to(dev), y.double().to(dev), (e.double() ** 2).to(dev)
    lik = g.likelihoods.FixedNoiseGaussianLikelihood(nz)
    class M(g.models.ExactGP):
        def __init__(s):
            super().__init__(x, yy, lik); s.mean_module = g.means.ConstantMean()
            per, rbf = K.PeriodicKernel(), K.RBFKernel(); per.period_length = 150.0; rbf.lengthscale = 750.0
            s.covar_module = K.ScaleKernel(K.ProductKernel(per, rbf))
        def forward(s, xx): return g.distributions.MultivariateNormal(s.mean_module(xx), s.covar_module(xx))
    m = M().double().to(dev); m.train(); lik.train()
    mll = g.mlls.ExactMarginalLogLikelihood(lik, m)
    def it(dense=False):
        m.zero_grad(); out = m(x)
        if dense: _ = out.lazy_covariance_matrix.K              # materialised: the dense back-end takes over
        l = -mll(out, yy); l.backward(); return l
    res = {}
    for dense in (False, True):
        it(dense); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): l = it(dense)
        torch.cuda.synchronize(); res[dense] = ((time.perf_counter() - t0) / 10, float(l))
    dt = res[False][0]
    print(f"n={n}: fused generic path {res[False][0]*1e3:.2f} ms/iter (loss {res[False][1]:.9f}); dense back-end {res[True][0]*1e3:.2f} ms/iter (loss {res[True][1]:.9f})")
    from pgmuvi_amd.gpytorch.kernels import compile_program
    prog = compile_program(m.covar_module, 1); th = prog.theta().detach()
    f = lambda: _hip.mll_kernel_value_grad(x, yy, yy.mean().expand(n), nz, None, prog, th)
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): f()
    torch.cuda.synchronize(); print(f"      pgm_mll_kernel_value_grad_f64 alone {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms")
    A = m.covar_module(x).to_dense().detach() + torch.diag(nz); r = yy - yy.mean()
    _hip.mll_dense(A, r); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): _hip.mll_dense(A, r)
    torch.cuda.synchronize(); dk = (time.perf_counter() - t0) / 10
    print(f"n={n}: quasi-periodic model->mll->backward {dt*1e3:.2f} ms/iter (loss {float(l):.5f}); pgm_mll_dense_f64 alone {dk*1e3:.2f} ms")
