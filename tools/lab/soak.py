"""tools/lab/soak.py -- 600 evaluations over ten alternating problem sizes on one device (workspace reuse, graph cache, every
schedule variant in turn): every result must equal the first one of its size bit for bit.   python tools/lab/soak.py  (GPU box)"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pgmuvi_amd import _hip, synthetic as syn
dev = torch.device("cuda:0"); D = torch.float64
sizes = [89, 4096, 256, 2048, 1000, 5120, 130, 3000, 4352, 640]
data = {}
for n in sizes:
    t, y, e = syn.cfg2(n_obs=n); h = syn.cfg_hypers(2, y.double())
    data[n] = (t.double().reshape(-1, 1).to(dev), y.double().to(dev), torch.full((n,), float(h["mean"]), dtype=D, device=dev), (e.double() ** 2).to(dev),
               h["w"].to(dev), h["mu"].reshape(4, 1).to(dev), h["v"].reshape(4, 1).to(dev))
ref = {}
t0 = time.time(); bad = 0
for rep in range(60):
    for n in sizes:
        x, y, m, nz, w, mu, v = data[n]
        o = _hip.mll_value_grad(x, y, m, nz, None, w, mu, v, 0, 0.0, True)
        key = (float(o["mll"]), float(o["g_w"].sum()), float(o["g_noise"].sum()), int(o["info"]))
        if n not in ref: ref[n] = key
        elif ref[n] != key: bad += 1; print("MISMATCH", rep, n, ref[n], key)
torch.cuda.synchronize()
print(f"{60 * len(sizes)} evaluations over {len(sizes)} alternating sizes in {time.time() - t0:.1f} s, {bad} mismatches against the first result of each size; memory {torch.cuda.memory_allocated() / 2**20:.0f} MiB")

# ragged batches on one workspace, alternating: the length table is re-uploaded and the launch graphs of the trimmed sets are
# re-captured whenever the lengths change (their class sizes are baked into the grids); every result must repeat bit for bit
from pgmuvi_amd.batch import evaluate_ragged, pad_curves, ragged_lengths
batches = []
for seed, (B, lo, hi) in enumerate([(96, 200, 1500), (40, 100, 900), (96, 300, 1400)]):
    lens = ragged_lengths(B, lo, hi, seed=seed + 11)
    curves = []
    for i, n in enumerate(lens):
        (t, y, e), per = syn.cfg3_lightcurve(i, n_obs=n)
        h = syn.cfg_hypers(3, y.double(), lead_period=per)
        curves.append(dict(x=t.double(), y=y.double(), noise=e.double() ** 2, mean=h["mean"], w=h["w"], mu=h["mu"], v=h["v"]))
    batches.append(pad_curves(curves, device=dev))
ws = _hip.Workspace(dev, 1500, 4, 1, 96)
def run(k):
    p, lens = batches[k]
    return _hip.mll_value_grad_ragged(p["x"], p["y"], p["mean"], p["noise"], None, lens, p["w"], p["mu"], p["v"], 0, 0.0, True, workspace=ws)
refs, bad = {}, 0
for k in range(3):
    o = run(k); torch.cuda.synchronize()
    refs[k] = (o["mll"].clone(), o["g_mu"].clone(), o["g_noise"].clone())
t0 = time.time()
for rep in range(20): o = run(0)
torch.cuda.synchronize(); same = (time.time() - t0) / 20 * 1e3
t0 = time.time()
for rep in range(20):
    for k in range(3):
        o = run(k); torch.cuda.synchronize()
        if not (torch.equal(o["mll"], refs[k][0]) and torch.equal(o["g_mu"], refs[k][1]) and torch.equal(o["g_noise"], refs[k][2]) and int(o["info"].abs().max()) == 0):
            bad += 1; print("RAGGED MISMATCH", rep, k)
alt = (time.time() - t0) / 60 * 1e3
t0 = time.time()
for rep in range(20): o = run(0)
torch.cuda.synchronize(); same2 = (time.time() - t0) / 20 * 1e3
print(f"ragged: 60 alternating calls of three batches on one workspace, {bad} mismatches; {alt:.2f} ms per call alternating (every call uploads its "
      f"table and re-captures its launch graphs) against {same:.2f} / {same2:.2f} ms for the first batch repeated")

# round 5: the sampler's potential (pgm_pot_*: a captured tick with workspace addresses and the early-pass tables baked in) interleaved
# with plain evaluations and small batches on the SAME cached workspaces -- every result must repeat bit for bit
import numpy as np
from pgmuvi_amd import mcmc
def chains(C, n):
    xs, ys, ns = [], [], []
    for c in range(C):
        (t, y, e), _ = syn.cfg3_lightcurve(5000 + c, n_obs=n)
        xs.append(t.double().reshape(n, 1)); ys.append(y.double()); ns.append(e.double() ** 2)
    return torch.stack(xs).to(dev), torch.stack(ys).to(dev), torch.stack(ns).to(dev)
x1, y1, n1 = chains(1, 2048); x8, y8, n8 = chains(8, 2048); xb, yb, nb_ = chains(12, 1024)
pot1 = mcmc.SMPotential(x1, y1, n1, num_mixtures=4); pot8 = mcmc.SMPotential(x8, y8, n8, num_mixtures=4)
rng = np.random.default_rng(5)
def zfor(pot):
    z = rng.normal(0, 0.2, (pot.B, pot.P)); z[:, 5:9] += np.log(1.0 / np.array([150.0, 67.0, 400.0, 31.0])); z[:, 9:13] += np.log(0.1 / np.array([150.0, 67.0, 400.0, 31.0])); return z
z1, z8 = zfor(pot1), zfor(pot8)
h = syn.cfg_hypers(2, yb[0].cpu())
wb, mub, vb = (t.to(dev).expand(12, *t.shape).contiguous() for t in (h["w"], h["mu"].reshape(4, 1), h["v"].reshape(4, 1)))
def plain(): 
    o = _hip.mll_value_grad(x1[0], y1[0], torch.zeros(2048, dtype=D, device=dev), n1[0], None, h["w"].to(dev), h["mu"].reshape(4, 1).to(dev), h["v"].reshape(4, 1).to(dev), 0, 0.0, True)
    return (float(o["mll"]), float(o["g_mu"].sum()))
def small():
    o = _hip.mll_value_grad(xb, yb, torch.zeros(12, 1024, dtype=D, device=dev), nb_, None, wb, mub, vb, 0, 0.0, True)
    return (float(o["mll"].sum()), float(o["g_mu"].sum()))
first, bad = {}, 0
t0 = time.time()
for rep in range(40):
    for name, f in (("pot1", lambda: tuple(np.concatenate([a.ravel() for a in pot1(z1)]))), ("plain", plain), ("pot8", lambda: tuple(np.concatenate([a.ravel() for a in pot8(z8)]))), ("small", small)):
        r = f()
        if name not in first: first[name] = r
        elif first[name] != r: bad += 1; print("POTENTIAL MISMATCH", rep, name)
torch.cuda.synchronize()
print(f"potential soak: 160 calls (one-chain tick, plain N=2048 evaluation, eight-chain tick, 12 x N=1024 batch, alternating on shared workspaces) in {time.time() - t0:.1f} s, {bad} mismatches")

# round 6: the one-launch path (k_small) interleaved with everything around it on shared cached workspaces -- single short light curves,
# a batch of them, a ragged batch of short ones, a device-resident fit advancing between the others, a potential on short light curves,
# prediction after a short evaluation -- every repeat bit for bit the first result
from pgmuvi_amd.batch import evaluate_ragged as _er
small_sizes = [1, 17, 64, 89, 100, 128]
sd = {}
for n in small_sizes:
    t, y, e = syn.cfg2(n_obs=n); hh = syn.cfg_hypers(2, y.double())
    sd[n] = (t.double().reshape(-1, 1).to(dev), y.double().to(dev), torch.full((n,), float(hh["mean"]), dtype=D, device=dev), (e.double() ** 2).to(dev),
             hh["w"].to(dev), hh["mu"].reshape(4, 1).to(dev), hh["v"].reshape(4, 1).to(dev))
Bs, ns_ = 37, 100
xs2, ys2, nz2 = chains(Bs, ns_)
hb = syn.cfg_hypers(2, ys2[0].cpu())
wb2, mub2, vb2 = (t.to(dev).expand(Bs, *t.shape).contiguous() for t in (hb["w"], hb["mu"].reshape(4, 1), hb["v"].reshape(4, 1)))
rag_curves = []
for i, n in enumerate([5, 128, 33, 97, 64, 120, 12, 77, 101, 128, 50, 88, 19, 127]):
    (t, y, e), per = syn.cfg3_lightcurve(i, n_obs=n)
    hh = syn.cfg_hypers(3, y.double(), lead_period=per)
    rag_curves.append(dict(x=t.double(), y=y.double(), noise=e.double() ** 2, mean=hh["mean"], w=hh["w"], mu=hh["mu"], v=hh["v"]))
rag_padded = pad_curves(rag_curves, device=dev)
xp1, yp1, np1 = chains(3, 96)
pot_s = mcmc.SMPotential(xp1, yp1, np1, num_mixtures=4)
zs = zfor(pot_s)
from pgmuvi_amd import gpytorch as g
from pgmuvi_amd.trainers import train_native
def fit_once():
    x, y, m, nz, w, mu, v = sd[89]
    lik = g.likelihoods.FixedNoiseGaussianLikelihood(nz)
    class M(g.models.ExactGP):
        def __init__(s):
            super().__init__(x.reshape(-1), y, lik); s.mean_module = g.means.ConstantMean(); s.covar_module = g.kernels.SpectralMixtureKernel(num_mixtures=4)
        def forward(s, xx): return g.distributions.MultivariateNormal(s.mean_module(xx), s.covar_module(xx))
    mm = M().double().to(dev)
    mm.initialize(**{"covar_module.mixture_weights": w, "covar_module.mixture_means": mu.reshape(4, 1, 1), "covar_module.mixture_scales": v.reshape(4, 1, 1)})
    r = train_native(model=mm, likelihood=lik, train_x=x.reshape(-1), train_y=y, maxiter=60, lr=0.01, optim="AdamW", stop=None, check_every=25)
    return tuple(float(v_) for v_ in r["loss"])
def one(n):
    x, y, m, nz, w, mu, v = sd[n]
    o = _hip.mll_value_grad(x, y, m, nz, None, w, mu, v, 0, 0.0, True)
    return (float(o["mll"]), float(o["g_w"].sum()), float(o["g_noise"].sum()), int(o["info"]))
def batch_small():
    o = _hip.mll_value_grad(xs2, ys2, torch.zeros(Bs, ns_, dtype=D, device=dev), nz2, None, wb2, mub2, vb2, 0, 0.0, True)
    return (float(o["mll"].sum()), float(o["g_mu"].sum()), int(o["info"].abs().max()))
def rag_small():
    p, lens = rag_padded
    o = _hip.mll_value_grad_ragged(p["x"], p["y"], p["mean"], p["noise"], None, lens, p["w"], p["mu"], p["v"], 0, 0.0, True)
    return (float(o["mll"].sum()), float(o["g_mu"].sum()), int(o["info"].abs().max()))
def predict_small():
    x, y, m, nz, w, mu, v = sd[89]
    o = _hip.mll_value_grad(x, y, m, nz, None, w, mu, v, 0, 0.0, True)
    xt = torch.linspace(float(x.min()), float(x.max()), 300, dtype=D, device=dev).reshape(-1, 1)
    pm, pv = _hip.predict(o["workspace"], xt, torch.zeros(300, dtype=D, device=dev))
    return (float(pm.sum()), float(pv.sum()))
first, bad, calls = {}, 0, 0
t0 = time.time()
jobs = [(f"n{n}", (lambda n=n: one(n))) for n in small_sizes] + [("batch37x100", batch_small), ("ragged14", rag_small), ("pot3x96", lambda: tuple(np.concatenate([a.ravel() for a in pot_s(zs)]))),
        ("predict89", predict_small), ("plain2048", plain), ("fit89", fit_once)]
for rep in range(25):
    for name, f in jobs:
        if name == "fit89" and rep % 5: continue
        r = f(); calls += 1
        if name not in first: first[name] = r
        elif first[name] != r: bad += 1; print("SMALL-PATH MISMATCH", rep, name)
torch.cuda.synchronize()
print(f"one-launch path soak: {calls} calls ({', '.join(n for n, _ in jobs)}; alternating on shared cached workspaces) in {time.time() - t0:.1f} s, {bad} mismatches")
