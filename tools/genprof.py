"""Host profile of one iteration of a composed-kernel model (quasi-periodic: ScaleKernel(Periodic * RBF)) through the Python
surface at N points:   python tools/genprof.py [n] [iters]"""
import cProfile, pstats, sys, time, torch
sys.path.insert(0, '/root/repo')
from pgmuvi_amd import gpytorch as g, synthetic as syn
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda:0"); K = g.kernels
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
opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
def it():
    opt.zero_grad(); out = m(x); l = -mll(out, yy); l.backward(); opt.step(); return l.cpu()
for _ in range(10): it()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(iters): it()
torch.cuda.synchronize(); print(f"n={n}: {(time.perf_counter() - t0) / iters * 1e3:.3f} ms/iteration")
pr = cProfile.Profile(); pr.enable()
for _ in range(iters): it()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(40)
