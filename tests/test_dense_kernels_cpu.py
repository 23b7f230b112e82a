"""CPU tests of the dense back-end's host side: the shim's non-spectral-mixture kernels
against the oracle's formulas, the dense MLL autograd node and eval-mode prediction (HIP calls replaced by torch
stand-ins: no GPU here), and the reference's own non-SM models (pgmuvi/gps.py:1075-1342) built on the shim."""
import math
import os
import subprocess
import sys
import textwrap
from unittest import mock

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import _oracle_backend as ob  # noqa: E402
from oracle import sm_mll_oracle as orc  # noqa: E402
from pgmuvi_amd import _hip, gpytorch as g, synthetic as syn  # noqa: E402

D = torch.float64
K = g.kernels


def _x(n=30, d=1, seed=0):
    gen = torch.Generator().manual_seed(seed)
    return torch.rand(n, d, generator=gen, dtype=D) * 100.0


def test_kernel_formulas_and_parameter_surface():
    x1, x2 = _x(17), _x(11, seed=1)
    rbf = K.RBFKernel().double(); rbf.lengthscale = 13.0
    assert torch.allclose(rbf(x1, x2).to_dense(), orc.rbf(x1, x2, 13.0), rtol=1e-13)
    assert rbf.raw_lengthscale.shape == (1, 1) and abs(float(rbf.lengthscale) - 13.0) < 1e-12
    for nu in (0.5, 1.5, 2.5):
        m = K.MaternKernel(nu=nu).double(); m.lengthscale = 20.0
        assert torch.allclose(m(x1, x2).to_dense(), orc.matern(x1, x2, 20.0, nu), rtol=1e-12)
    with pytest.raises(RuntimeError):
        K.MaternKernel(nu=1.0)
    per = K.PeriodicKernel().double(); per.period_length = 37.0; per.lengthscale = 0.8
    assert torch.allclose(per(x1, x2).to_dense(), orc.periodic(x1, x2, 37.0, 0.8), rtol=1e-12)
    rq = K.RQKernel().double(); rq.lengthscale = 9.0; rq.alpha = 1.7
    assert torch.allclose(rq(x1, x2).to_dense(), orc.rq(x1, x2, 9.0, 1.7), rtol=1e-12)
    # defaults are softplus(0), like every GPyTorch kernel parameter
    assert abs(float(K.ScaleKernel(K.RBFKernel()).outputscale) - math.log(2.0)) < 1e-7
    # composition: ScaleKernel(Periodic * RBF) -- pgmuvi/gps.py:915-935 -- and sums
    qp = K.ScaleKernel(K.ProductKernel(K.PeriodicKernel(), K.RBFKernel())).double()
    qp.base_kernel.kernels[0].period_length = 37.0
    qp.base_kernel.kernels[1].lengthscale = 185.0
    qp.outputscale = 2.5
    ref = 2.5 * orc.periodic(x1, x1, 37.0, math.log(2.0)) * orc.rbf(x1, x1, 185.0)
    assert torch.allclose(qp(x1).to_dense(), ref, rtol=1e-12)
    assert [n for n, _ in qp.named_parameters()] == ["raw_outputscale", "base_kernel.kernels.0.raw_lengthscale",
                                                     "base_kernel.kernels.0.raw_period_length", "base_kernel.kernels.1.raw_lengthscale"]
    s = (K.ScaleKernel(K.RBFKernel()) + K.ScaleKernel(K.MaternKernel(nu=1.5))).double()
    assert isinstance(s, K.AdditiveKernel) and isinstance(K.RBFKernel() * K.RBFKernel(), K.ProductKernel)
    assert torch.allclose(s(x1).to_dense(), sum(k(x1).to_dense() for k in s.kernels))
    # active_dims re-registered as a buffer, the separable 2-D construction of pgmuvi/gps.py:1320-1336
    t, w = K.ScaleKernel(K.MaternKernel(nu=1.5)).double(), K.ScaleKernel(K.RBFKernel()).double()
    t.register_buffer("active_dims", torch.tensor([0], dtype=torch.long))
    w.register_buffer("active_dims", torch.tensor([1], dtype=torch.long))
    X = torch.cat([_x(12), _x(12, seed=5) / 50.0], dim=1)
    sep = (t * w)(X).to_dense()
    ls = math.log(2.0)
    assert torch.allclose(sep, ls * orc.matern(X[:, :1], X[:, :1], ls, 1.5) * ls * orc.rbf(X[:, 1:], X[:, 1:], ls), rtol=1e-12)
    # the spectral-mixture kernel inside a sum: same numbers as its fused formula
    smk = K.SpectralMixtureKernel(num_mixtures=2).double()
    smk.mixture_means = torch.tensor([0.02, 0.05], dtype=D).reshape(2, 1, 1)
    smk.mixture_scales = torch.tensor([0.003, 0.01], dtype=D).reshape(2, 1, 1)
    smk.mixture_weights = torch.tensor([0.7, 0.2], dtype=D)
    mix = (smk + K.ScaleKernel(K.RBFKernel(ard_num_dims=1))).double()
    ref = orc.sm_kernel(x1, x1, smk.mixture_weights, smk.mixture_means.reshape(2, 1), smk.mixture_scales.reshape(2, 1)) + ls * orc.rbf(x1, x1, ls)
    assert torch.allclose(mix(x1).to_dense(), ref.to(D), rtol=1e-11, atol=1e-13)


def _matern_model(x, y, lik):
    class M(g.models.ExactGP):
        def __init__(self):
            super().__init__(x, y, lik)
            self.mean_module = g.means.ConstantMean()
            base = K.MaternKernel(nu=1.5); base.lengthscale = 25.0
            self.covar_module = K.ScaleKernel(base)

        def forward(self, xx):
            return g.distributions.MultivariateNormal(self.mean_module(xx), self.covar_module(xx))
    return M().double()


def test_dense_mll_backward_and_prediction_with_standins():
    t, y, e = syn.cfg2(n_obs=60)
    x, y, noise = t.double(), y.double(), e.double() ** 2
    for lik in (g.likelihoods.FixedNoiseGaussianLikelihood(noise), g.likelihoods.GaussianLikelihood().double()):
        m = _matern_model(x, y, lik)
        m.train(); lik.train()
        with mock.patch.object(_hip, "mll_dense", ob.mll_dense), mock.patch.object(_hip, "mll_kernel_value_grad", ob.mll_kernel_value_grad), mock.patch.object(_hip, "require_gpu", lambda *a, **k: None):
            mll = g.mlls.ExactMarginalLogLikelihood(lik, m)
            loss = -mll(m(x), y)
            loss.backward()
        # oracle: same raw parameters through the same transforms, autograd on the dense graph
        raw = {n: p.detach().clone().requires_grad_(True) for n, p in m.named_parameters()}
        osc = orc.positive(raw["covar_module.raw_outputscale"])
        ell = orc.positive(raw["covar_module.base_kernel.raw_lengthscale"]).reshape(())
        nz = noise if "likelihood.noise_covar.raw_noise" not in raw else orc.greater_than(raw["likelihood.noise_covar.raw_noise"], 1e-4).reshape(())
        ref = -orc.mll_dense(osc * orc.matern(x, x, ell, 1.5), y, raw["mean_module.raw_constant"], nz)
        ref.backward()
        assert abs(float(loss.detach()) - float(ref.detach())) < 1e-11
        for n, p in m.named_parameters():
            assert torch.allclose(p.grad, raw[n].grad, rtol=1e-8, atol=1e-11), n
    # eval mode through the dense prediction entry point
    lik = g.likelihoods.FixedNoiseGaussianLikelihood(noise)
    m = _matern_model(x, y, lik)
    m.eval(); lik.eval()
    xs = torch.linspace(float(x.min()), float(x.max()), 40, dtype=D)
    with mock.patch.object(_hip, "mll_dense", ob.mll_dense), mock.patch.object(_hip, "mll_kernel_value_grad", ob.mll_kernel_value_grad), mock.patch.object(_hip, "predict_dense", ob.predict_dense), torch.no_grad():
        pred = m(xs)
    ls = math.log(2.0)
    Kxx, Kxs = ls * orc.matern(x, x, 25.0, 1.5), ls * orc.matern(x, xs, 25.0, 1.5)
    pm, pv = orc.posterior_dense(Kxx, Kxs, torch.full((40,), ls, dtype=D), y, torch.zeros((), dtype=D), noise, torch.zeros((), dtype=D))
    assert torch.allclose(pred.mean, pm, atol=1e-10) and torch.allclose(pred.variance, pv, atol=1e-10)


REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "pgmuvi")), reason="reference checkout not present")
def test_reference_non_sm_models_train_through_the_shim():
    """pgmuvi/gps.py's MaternGPModel, QuasiPeriodicGPModel, PeriodicPlusStochasticGPModel and the separable 2-D model,
    unmodified, built on the shim's kernels and trained by the mirror of trainers.train (dense stand-in)."""
    prog = textwrap.dedent("""
        import sys, warnings, torch, numpy as np
        sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)
        from unittest import mock
        import pgmuvi_amd
        from pgmuvi_amd import _hip, synthetic as syn
        import _oracle_backend as ob
        pgmuvi_amd.install_as_gpytorch()
        warnings.simplefilter("ignore")
        import gpytorch, pgmuvi.gps as gps
        from pgmuvi_amd.trainers import train      # (the reference's loop needs a Lightcurve for its results bookkeeping)
        t, y, e = syn.cfg2(n_obs=50)
        x, y, nz = t.double(), y.double(), e.double() ** 2
        with mock.patch.object(_hip, "mll_dense", ob.mll_dense), mock.patch.object(_hip, "mll_kernel_value_grad", ob.mll_kernel_value_grad), mock.patch.object(_hip, "require_gpu", lambda *a, **k: None):
            for make in (lambda l: gps.MaternGPModel(x, y, l, nu=1.5), lambda l: gps.QuasiPeriodicGPModel(x, y, l, period=150.0),
                         lambda l: gps.PeriodicPlusStochasticGPModel(x, y, l, period=150.0)):
                lik = gpytorch.likelihoods.FixedNoiseGaussianLikelihood(nz)
                m = make(lik).double()
                res = train(model=m, likelihood=lik, train_x=x, train_y=y, maxiter=6, lr=0.05, optim="AdamW", progress=False)
                L = [float(v) for v in res["loss"]]
                assert len(L) == 6 and all(np.isfinite(L)) and L[-1] < L[0], (type(m).__name__, L)
            X, Y, E = syn.cfg4(n_per_band=6)
            lik = gpytorch.likelihoods.FixedNoiseGaussianLikelihood(E.double() ** 2)
            m = gps.SeparableGPModel(X.double(), Y.double(), lik).double()
            res = train(model=m, likelihood=lik, train_x=X.double(), train_y=Y.double(), maxiter=4, lr=0.05, optim="AdamW", progress=False)
            assert all(np.isfinite(float(v)) for v in res["loss"])
        print("NONSM_OK")
    """ % (ROOT, os.path.join(ROOT, "tests"), REF))
    r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "NONSM_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
