"""Host cost of the optimiser step inside the reference-shaped train() at N=1024: torch's default (foreach) AdamW against
fused=True, and the loss trajectories of the two side by side."""
import sys, time, torch
sys.path.insert(0, '/root/repo')
from pgmuvi_amd import gpytorch as g, synthetic as syn, trainers
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
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
variants = {"foreach": lambda params, lr, eps, **kw: torch.optim.AdamW(params, lr=lr, eps=eps, foreach=True, **{k: v for k, v in kw.items() if k != "fused"}),
            "fused (train()'s choice on the GPU)": lambda params, lr, eps, **kw: torch.optim.AdamW(params, lr=lr, eps=eps, **dict(kw, fused=True)),
            "single-tensor": lambda params, lr, eps, **kw: torch.optim.AdamW(params, lr=lr, eps=eps, foreach=False, **{k: v for k, v in kw.items() if k != "fused"})}
res = {}
for name, mk in variants.items():
    trainers._OPTIMISERS["AdamW"] = mk
    m, lik = build()
    trainers.train(model=m, likelihood=lik, train_x=x, train_y=yy, maxiter=5, lr=0.001, optim="AdamW", progress=False)
    m, lik = build()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = trainers.train(model=m, likelihood=lik, train_x=x, train_y=yy, maxiter=200, lr=0.001, optim="AdamW", progress=False)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
    res[name] = [float(v) for v in r["loss"]]
    print(f"n={n} {name}: {dt*1e3:.3f} ms/iter ({1/dt:.0f} it/s), last loss {res[name][-1]:.12f}")
ref = res["foreach"]
for name in res:
    print(name, "max |loss - foreach|", max(abs(a - b) for a, b in zip(res[name], ref)))
