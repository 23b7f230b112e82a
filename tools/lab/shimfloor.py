"""How much of a reference-shaped iteration (model(x) -> mll -> backward -> AdamW step -> loss.cpu()) is the drop-in surface's own
Python, and how much torch's?  Same N=89, Q=4 problem three ways: (a) through the surface; (b) the same HIP call with the softplus chain
rule written out on raw tensors (no Module/__call__/distribution objects, no autograd graph) + torch's AdamW + loss.cpu();
(c) optimiser step and loss.cpu() alone.   python tools/lab/shimfloor.py [n]"""
import sys, time, torch
sys.path.insert(0, '/root/repo')
from pgmuvi_amd import _hip, gpytorch as g, synthetic as syn
n = int(sys.argv[1]) if len(sys.argv) > 1 else 89
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
def a():
    opt.zero_grad(); out = m(x); loss = -mll(out, yy); loss.backward(); opt.step(); return loss.cpu().detach().numpy()
raw = [torch.zeros(4, dtype=torch.float64, device=dev, requires_grad=True), torch.full((4, 1, 1), -4.0, dtype=torch.float64, device=dev, requires_grad=True),
       torch.full((4, 1, 1), -6.0, dtype=torch.float64, device=dev, requires_grad=True), torch.zeros((), dtype=torch.float64, device=dev, requires_grad=True)]
opt2 = torch.optim.AdamW(raw, lr=1e-3)
xx = x.reshape(n, 1)
def b():
    opt2.zero_grad()
    with torch.no_grad():
        w, mu, v = torch.nn.functional.softplus(raw[0]), torch.nn.functional.softplus(raw[1]), torch.nn.functional.softplus(raw[2])
        out = _hip.mll_value_grad(xx, yy, raw[3].expand(n), nz, None, w, mu.reshape(4, 1), v.reshape(4, 1), 0, 0.0, True)
        raw[0].grad = -out["g_w"] * torch.sigmoid(raw[0]); raw[1].grad = (-out["g_mu"] * torch.sigmoid(raw[1]).reshape(4, 1)).reshape(4, 1, 1)
        raw[2].grad = (-out["g_v"] * torch.sigmoid(raw[2]).reshape(4, 1)).reshape(4, 1, 1); raw[3].grad = -out["g_mean"].sum()
    opt2.step()
    return (-out["mll"]).cpu().numpy()
lossbuf = torch.zeros((), dtype=torch.float64, device=dev)
def c():
    opt2.step(); return lossbuf.cpu().numpy()
for name, f in (("(a) through the surface", a), ("(b) raw tensors, chain rule by hand", b), ("(c) AdamW step + loss.cpu() alone", c)):
    for _ in range(20): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(500): f()
    torch.cuda.synchronize(); print(f"n={n} {name}: {(time.perf_counter() - t0) / 500 * 1e6:.0f} us per iteration")
