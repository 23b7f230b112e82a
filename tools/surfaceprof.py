"""Where the host time of one reference-shaped iteration (model(x) -> mll -> backward -> optimiser step -> loss.cpu()) goes:
cProfile over 200 iterations at a size where the device work is short.   python tools/surfaceprof.py [n]"""
import cProfile, pstats, sys, time, torch
sys.path.insert(0, '/root/repo')
from pgmuvi_amd import gpytorch as g, synthetic as syn
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda:0")
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
m.train(); lik.train()
mll = g.mlls.ExactMarginalLogLikelihood(lik, m)
opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
def it():
    opt.zero_grad(); out = m(x); loss = -mll(out, yy); loss.backward(); opt.step(); return loss.cpu().detach().numpy()
for _ in range(5): it()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): it()
torch.cuda.synchronize(); print(f"n={n}: {(time.perf_counter()-t0)/200*1e3:.3f} ms/iteration")
pr = cProfile.Profile(); pr.enable()
for _ in range(200): it()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
