import sys, time, torch, warnings
sys.path.insert(0, '/root/repo')
from pgmuvi_amd import gpytorch as g, synthetic as syn
from pgmuvi_amd.trainers import train
dev = torch.device("cuda:0")
for n in (1024, 4096):
    t, y, e = syn.cfg2(n_obs=n)
    x, yy, nz = t.double().to(dev), y.double().to(dev), (e.double() ** 2).to(dev)
    lik = g.likelihoods.FixedNoiseGaussianLikelihood(nz)
    class M(g.models.ExactGP):
        def __init__(s):
            super().__init__(x, yy, lik); s.mean_module = g.means.ConstantMean(); s.covar_module = g.kernels.SpectralMixtureKernel(num_mixtures=4)
        def forward(s, xx): return g.distributions.MultivariateNormal(s.mean_module(xx), s.covar_module(xx))
    m = M().double().to(dev)
    h = syn.cfg_hypers(2, y.double())
    m.initialize(**{"covar_module.mixture_weights": h["w"].to(dev), "covar_module.mixture_means": h["mu"].to(dev), "covar_module.mixture_scales": h["v"].to(dev)})
    train(model=m, likelihood=lik, train_x=x, train_y=yy, maxiter=5, lr=0.001, optim="AdamW", progress=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = train(model=m, likelihood=lik, train_x=x, train_y=yy, maxiter=50, lr=0.001, optim="AdamW", progress=False)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    print(f"n={n}: train() {dt*1e3:.3f} ms/iter ({1/dt:.1f} it/s), loss {float(res['loss'][0]):.5f} -> {float(res['loss'][-1]):.5f}")
    # pure surface eval without the trainer's bookkeeping
    mll = g.mlls.ExactMarginalLogLikelihood(lik, m); m.train()
    for _ in range(3): l = -mll(m(x), yy); l.backward()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30):
        m.zero_grad(); l = -mll(m(x), yy); l.backward()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
    print(f"n={n}: model->mll->backward {dt*1e3:.3f} ms/eval")
    with g.settings.check_cholesky_info(False):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30):
            m.zero_grad(); l = -mll(m(x), yy); l.backward()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
    print(f"n={n}: same without the info readback {dt*1e3:.3f} ms/eval")

from pgmuvi_amd.trainers import train_device
for n in (256, 1024, 4096):
    t, y, e = syn.cfg2(n_obs=n)
    x, yy, nz = t.double().to(dev), y.double().to(dev), (e.double() ** 2).to(dev)
    def build():
        lik = g.likelihoods.FixedNoiseGaussianLikelihood(nz)
        class M(g.models.ExactGP):
            def __init__(s):
                super().__init__(x, yy, lik); s.mean_module = g.means.ConstantMean(); s.covar_module = g.kernels.SpectralMixtureKernel(num_mixtures=4)
            def forward(s, xx): return g.distributions.MultivariateNormal(s.mean_module(xx), s.covar_module(xx))
        m = M().double().to(dev)
        h = syn.cfg_hypers(2, y.double())
        m.initialize(**{"covar_module.mixture_weights": h["w"].to(dev), "covar_module.mixture_means": h["mu"].to(dev), "covar_module.mixture_scales": h["v"].to(dev)})
        return m, lik
    m1, l1 = build(); m2, l2 = build()
    r1 = train(model=m1, likelihood=l1, train_x=x, train_y=yy, maxiter=60, lr=0.001, optim="AdamW", progress=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r2 = train_device(model=m2, likelihood=l2, train_x=x, train_y=yy, maxiter=60, lr=0.001, optim="AdamW", check_every=20)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    import numpy as np
    d = max(abs(float(a) - float(b)) for a, b in zip(r1["loss"], r2["loss"]))
    m3, l3 = build()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    train(model=m3, likelihood=l3, train_x=x, train_y=yy, maxiter=60, lr=0.001, optim="AdamW", progress=False)
    torch.cuda.synchronize(); dt1 = time.perf_counter() - t0
    print(f"n={n}: train_device {dt/60*1e3:.3f} ms/iter incl. capture (train: {dt1/60*1e3:.3f}); max |loss diff| vs train() {d:.2e}; last losses {float(r1['loss'][-1]):.8f} {float(r2['loss'][-1]):.8f}")

from pgmuvi_amd.trainers import train_native
for n in (89, 256, 1024, 4096):
    t, y, e = syn.cfg2(n_obs=n)
    x, yy, nz = t.double().to(dev), y.double().to(dev), (e.double() ** 2).to(dev)
    lik = g.likelihoods.FixedNoiseGaussianLikelihood(nz)
    class M(g.models.ExactGP):
        def __init__(s):
            super().__init__(x, yy, lik); s.mean_module = g.means.ConstantMean(); s.covar_module = g.kernels.SpectralMixtureKernel(num_mixtures=4)
        def forward(s, xx): return g.distributions.MultivariateNormal(s.mean_module(xx), s.covar_module(xx))
    m = M().double().to(dev)
    h = syn.cfg_hypers(2, y.double())
    m.initialize(**{"covar_module.mixture_weights": h["w"].to(dev), "covar_module.mixture_means": h["mu"].to(dev), "covar_module.mixture_scales": h["v"].to(dev)})
    train_native(model=m, likelihood=lik, train_x=x, train_y=yy, maxiter=5, lr=0.001, optim="AdamW")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = train_native(model=m, likelihood=lik, train_x=x, train_y=yy, maxiter=200, lr=0.001, optim="AdamW", check_every=50)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"n={n}: train_native {dt/200*1e3:.3f} ms/iter incl. graph capture ({200/dt:.0f} it/s); loss {r['loss'][0]:.6f} -> {r['loss'][-1]:.6f}")
