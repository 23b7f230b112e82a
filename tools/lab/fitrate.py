"""Steady-state rate of the device-resident fit (train_native) at several sizes: 1000 iterations, one read of the log at the end,
set-up (workspace, graph capture) reported separately.   python tools/lab/fitrate.py [n ...]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pgmuvi_amd import _hip, trainers, synthetic as syn
from pgmuvi_amd import gpytorch as g
dev = torch.device("cuda:0"); D = torch.float64
acc = {}
def wrap(cls, name):
    fn = getattr(cls, name)
    def w(*a, **k):
        t0 = time.perf_counter(); r = fn(*a, **k); acc[name] = acc.get(name, 0) + time.perf_counter() - t0; return r
    setattr(cls, name, w)
for nm in ("__init__", "run", "read"):
    wrap(_hip.NativeFit, nm)
for n in [int(a) for a in sys.argv[1:]] or [89, 256, 512, 1024, 2048]:
    t, y, e = syn.cfg2(n_obs=n)
    x, yy, nz = t.double().to(dev), y.double().to(dev), (e.double() ** 2).to(dev)
    h = syn.cfg_hypers(2, y.double())
    def make():
        lik = g.likelihoods.FixedNoiseGaussianLikelihood(nz)
        class M(g.models.ExactGP):
            def __init__(s):
                super().__init__(x, yy, lik); s.mean_module = g.means.ConstantMean(); s.covar_module = g.kernels.SpectralMixtureKernel(num_mixtures=4)
            def forward(s, xx): return g.distributions.MultivariateNormal(s.mean_module(xx), s.covar_module(xx))
        m = M().double().to(dev)
        m.initialize(**{"covar_module.mixture_weights": h["w"].to(dev), "covar_module.mixture_means": h["mu"].to(dev), "covar_module.mixture_scales": h["v"].to(dev)})
        return m, lik
    m, lik = make(); trainers.train_native(model=m, likelihood=lik, train_x=x, train_y=yy, maxiter=16, lr=1e-3, optim="AdamW", stop=None, check_every=16)
    acc.clear(); m, lik = make(); torch.cuda.synchronize(); t0 = time.perf_counter()
    trainers.train_native(model=m, likelihood=lik, train_x=x, train_y=yy, maxiter=1000, lr=1e-3, optim="AdamW", stop=None, check_every=1000)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"n={n}: {acc['read'] * 1e3:.3f} us per iteration on the device ({1000 / acc['read']:.0f} it/s); set-up {acc['__init__'] * 1e3:.2f} ms, enqueue {acc['run'] * 1e3:.2f} ms, whole call {dt * 1e3:.1f} ms")
