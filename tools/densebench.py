#!/usr/bin/env python3
"""Dense back-end timing: a quasi-periodic exact GP (ScaleKernel(Periodic * RBF), pgmuvi/gps.py:915-935) at N points:
model -> mll -> backward per iteration (kernel matrix and its pull-back by torch, factorisation + dmll/dA by the HIP library)."""
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
    def it():
        m.zero_grad(); l = -mll(m(x), yy); l.backward(); return l
    it(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): l = it()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    A = m.covar_module(x).to_dense().detach() + torch.diag(nz); r = yy - yy.mean()
    _hip.mll_dense(A, r); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): _hip.mll_dense(A, r)
    torch.cuda.synchronize(); dk = (time.perf_counter() - t0) / 10
    print(f"n={n}: quasi-periodic model->mll->backward {dt*1e3:.2f} ms/iter (loss {float(l):.5f}); pgm_mll_dense_f64 alone {dk*1e3:.2f} ms")
