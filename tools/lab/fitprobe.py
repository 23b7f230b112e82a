"""Where the wall time of train_native at N=89 goes: create / run (enqueue) / read (wait + copy) / host bookkeeping."""
import os, sys, time, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pgmuvi_amd import _hip, trainers
from pgmuvi_amd import gpytorch as g
dev = torch.device("cuda:0")
p = np.load(os.path.join(ROOT, "tests", "golden", "notebook_pin_1d.npz"))
D = torch.float64
x, y, noise = (torch.as_tensor(p[k], dtype=D).to(dev) for k in ("x", "y", "noise"))
def make():
    lik = g.likelihoods.FixedNoiseGaussianLikelihood(noise)
    class Model(g.models.ExactGP):
        def __init__(self):
            super().__init__(x, y, lik)
            self.mean_module = g.means.ConstantMean(); self.covar_module = g.kernels.SpectralMixtureKernel(num_mixtures=2)
        def forward(self, xx): return g.distributions.MultivariateNormal(self.mean_module(xx), self.covar_module(xx))
    return Model().to(D).to(dev), lik
acc = {}
def wrap(cls, name):
    fn = getattr(cls, name)
    def w(*a, **k):
        t0 = time.perf_counter(); r = fn(*a, **k); acc[name] = acc.get(name, 0) + time.perf_counter() - t0; acc[name + "_n"] = acc.get(name + "_n", 0) + 1; return r
    setattr(cls, name, w)
for nm in ("__init__", "run", "read", "close"):
    wrap(_hip.NativeFit, nm)
for ce in (250, 1000):
    m, lik = make(); trainers.train_native(model=m, likelihood=lik, train_x=x, train_y=y, maxiter=20, lr=0.05, optim="AdamW", stop=None, check_every=ce)
    acc.clear()
    m, lik = make(); torch.cuda.synchronize(); t0 = time.perf_counter()
    trainers.train_native(model=m, likelihood=lik, train_x=x, train_y=y, maxiter=1000, lr=0.05, optim="AdamW", stop=None, check_every=ce)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("check_every", ce, "wall ms", round(dt * 1e3, 2), {k: (round(v * 1e3, 2) if not k.endswith("_n") else v) for k, v in acc.items()})
