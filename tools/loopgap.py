"""Where the GPU idles between two iterations of the reference-shaped loop (N points): host time from the return of
loss.cpu() to the next evaluation's launch, phase by phase.   python tools/loopgap.py [n]"""
import sys, time, torch, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pgmuvi_amd import gpytorch as g, synthetic as syn, _hip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
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
real = _hip.mll_value_grad
stamp = {}
def timed(*a, **k):
    stamp["call"] = time.perf_counter(); r = real(*a, **k); stamp["ret"] = time.perf_counter(); return r
_hip.mll_value_grad = timed
import pgmuvi_amd.mll_function as mf
names = ["zero_grad", "model(x)", "mll() up to the C call", "C call (launch)", "status wait + rest of mll()", "backward", "optimizer.step", "loss.cpu()", "4 x param.cpu()"]
acc = np.zeros(len(names)); iters = 300
for i in range(iters + 20):
    t0 = time.perf_counter(); opt.zero_grad()
    t1 = time.perf_counter(); out = m(x)
    t2 = time.perf_counter(); loss = -mll(out, yy)
    t3 = time.perf_counter(); loss.backward()
    t4 = time.perf_counter(); opt.step()
    t5 = time.perf_counter(); v = loss.cpu().detach().numpy()
    t6 = time.perf_counter()
    for name, p in m.named_parameters(): p.cpu().detach().numpy()
    t7 = time.perf_counter()
    if i >= 20:
        acc += [t1 - t0, t2 - t1, stamp["call"] - t2, stamp["ret"] - stamp["call"], t3 - stamp["ret"], t4 - t3, t5 - t4, t6 - t5, t7 - t6]
tot = acc.sum() / iters
print(f"n={n}: {tot*1e3:.3f} ms per iteration")
for nm, a in zip(names, acc / iters): print(f"  {nm:32s} {a*1e6:8.1f} us")
