"""Host profile of the reference-shaped fit loop (pgmuvi_amd.trainers.train) at N points: cProfile over `iters` iterations.
   python tools/trainprof.py [n] [iters]"""
import cProfile, pstats, sys, time, torch
sys.path.insert(0, '/root/repo')
from pgmuvi_amd import gpytorch as g, synthetic as syn, trainers
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
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
trainers.train(model=m, likelihood=lik, train_x=x, train_y=yy, maxiter=20, miniter=20, lr=1e-3, optim="AdamW", progress=False)
torch.cuda.synchronize(); t0 = time.perf_counter()
trainers.train(model=m, likelihood=lik, train_x=x, train_y=yy, maxiter=iters, miniter=iters, lr=1e-3, optim="AdamW", progress=False)
torch.cuda.synchronize(); print(f"n={n}: train() {(time.perf_counter()-t0)/iters*1e3:.3f} ms/iteration")
pr = cProfile.Profile(); pr.enable()
trainers.train(model=m, likelihood=lik, train_x=x, train_y=yy, maxiter=iters, miniter=iters, lr=1e-3, optim="AdamW", progress=False)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
st.sort_stats("tottime").print_stats("pgmuvi_amd", 30)
