"""Random soak of the cached workspaces: a random sequence of calls -- single evaluations of 1 .. 1600 points, equal-length batches,
ragged batches, predictions, short device-resident fits and sampler-potential ticks of random shapes -- all through the binding's
workspace cache (a cached workspace serves every smaller shape: graph caches, work-item tables, early-pass tables, the one-launch
flag and the status stamps are shared between the shapes that meet on it).  Every evaluation is repeated on a FRESH workspace of
exactly its own shape: the value must be the same bits, the gradients equal to rounding (which workspace serves a call can change
how its gradient sums are split, never what is summed).

    python tools/lab/random_soak.py [calls] [seed]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pgmuvi_amd import _hip, mcmc  # noqa: E402
from pgmuvi_amd import gpytorch as g  # noqa: E402
from pgmuvi_amd.trainers import train_native  # noqa: E402

D = torch.float64
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(seed)
rng = np.random.default_rng(seed)
ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=gen))


def curve(n, d, q):
    x = torch.rand(n, d, generator=gen, dtype=D) * 900.0
    if d == 1:
        x = torch.sort(x[:, 0])[0].reshape(n, 1)
    else:
        x[:, 1] = torch.randint(1, 4, (n,), generator=gen).double() * 0.5
    return (x, torch.randn(n, generator=gen, dtype=D), 0.01 + 0.05 * torch.rand(n, generator=gen, dtype=D), 0.1 + torch.rand(q, generator=gen, dtype=D),
            0.005 + 0.3 * torch.rand(q, d, generator=gen, dtype=D), 0.001 + 0.02 * torch.rand(q, d, generator=gen, dtype=D))


def size():
    r = ri(0, 9)
    return ri(1, 128) if r < 3 else (ri(129, 520) if r < 6 else (ri(521, 1100) if r < 9 else ri(1101, 1600)))


def dev_of(a, r, th):
    va = torch.cat([a[k].cpu().reshape(-1) * t.reshape(-1) for k, t in zip(("g_w", "g_mu", "g_v"), th)])
    vr = torch.cat([r[k].cpu().reshape(-1) * t.reshape(-1) for k, t in zip(("g_w", "g_mu", "g_v"), th)])
    return float((va - vr).abs().max() / (vr.abs().max() + 1e-300))


bad, worst, kinds = 0, 0.0, {}
for c in range(calls):
    kind = ("single", "single", "single", "batch", "ragged", "predict", "fit", "potential")[ri(0, 7)]
    kinds[kind] = kinds.get(kind, 0) + 1
    d = 1 + (ri(0, 4) == 0); q = ri(1, 4); order = ri(0, 1) if d == 2 else 0
    if os.environ.get("SOAK_TRACE"):
        print(f"-> call {c}: {kind} q={q} d={d} order={order}", flush=True)
    if c < int(os.environ.get("SOAK_FIRST", "0")):             # (tools/lab/soak_bisect.py: the draws of the earlier calls, nothing on the GPU)
        if kind in ("single", "predict"):
            curve(size(), d if kind == "single" else 1, q)
            if kind == "predict":
                torch.rand(ri(1, 300), generator=gen, dtype=D)
        elif kind == "batch":
            B, n = ri(2, 24), size()
            [curve(n, d, q) for _ in range(B)]
        elif kind == "ragged":
            for n in [size() for _ in range(ri(3, 30))]:
                curve(n, d, q)
            ri(0, 1)
        elif kind == "fit":
            curve(size() if ri(0, 1) else ri(10, 128), 1, q); ri(3, 40)
        else:
            C, n = ri(1, 6), max(size(), 3)
            [curve(n, 1, q) for _ in range(C)]
            for _ in range(ri(1, 4)):
                rng.normal(0, 0.3, (C, 1 + 3 * q))
        continue
    if kind in ("single", "predict"):
        n = size()
        x, y, nz, w, mu, v = curve(n, d if kind == "single" else 1, q)
        dd = x.shape[1]
        a = (x.to(dev), y.to(dev), torch.full((n,), 0.1, dtype=D, device=dev), nz.to(dev), None, w.to(dev), mu.to(dev), v.to(dev), order if dd == 2 else 0, 0.0, True)
        one = _hip.mll_value_grad(*a)
        torch.cuda.synchronize()
        pred = None
        if kind == "predict":
            xs = torch.rand(ri(1, 300), generator=gen, dtype=D) * 900.0
            pred = _hip.predict(one["workspace"], xs.reshape(-1, 1).to(dev), torch.full((xs.numel(),), 0.1, dtype=D, device=dev))
            torch.cuda.synchronize()
        fresh = _hip.Workspace(dev, n, q, dd, 1)
        two = _hip.mll_value_grad(*a, workspace=fresh)
        torch.cuda.synchronize()
        ok = float(one["mll"]) == float(two["mll"]) and int(one["info"]) == 0
        worst = max(worst, dev_of(one, two, (w, mu, v)))
        if pred is not None:
            p2 = _hip.predict(fresh, xs.reshape(-1, 1).to(dev), torch.full((xs.numel(),), 0.1, dtype=D, device=dev))
            torch.cuda.synchronize()
            ok = ok and torch.equal(pred[0], p2[0]) and torch.equal(pred[1], p2[1])
        fresh.close()
        if not ok:
            bad += 1
            print(f"call {c} {kind} n={n} q={q} d={dd}: cached {float(one['mll'])!r} fresh {float(two['mll'])!r}")
    elif kind == "batch":
        B, n = ri(2, 24), size()
        cs = [curve(n, d, q) for _ in range(B)]
        st = lambda i: torch.stack([t[i] for t in cs]).to(dev)
        a = (st(0), st(1), torch.full((B, n), 0.1, dtype=D, device=dev), st(2), None, st(3), st(4), st(5), order, 0.0, True)
        one = _hip.mll_value_grad(*a)
        torch.cuda.synchronize()
        keep = one["mll"].clone()
        fresh = _hip.Workspace(dev, n, q, d, B)
        two = _hip.mll_value_grad(*a, workspace=fresh)
        torch.cuda.synchronize()
        if not (torch.equal(keep, two["mll"]) and int(two["info"].abs().max()) == 0):
            bad += 1
            print(f"call {c} batch B={B} n={n} q={q} d={d}: values differ between the cached and a fresh workspace")
        fresh.close()
    elif kind == "ragged":
        B = ri(3, 30)
        lengths = [size() for _ in range(B)]
        S = max(lengths)
        x = torch.zeros(B, S, d, dtype=D); y = torch.zeros(B, S, dtype=D); nz = torch.zeros(B, S, dtype=D)
        ws_, mus, vs = [], [], []
        for b, n in enumerate(lengths):
            xb, yb, nb_, w, mu, v = curve(n, d, q)
            x[b, :n], y[b, :n], nz[b, :n] = xb, yb, nb_
            ws_.append(w); mus.append(mu); vs.append(v)
        a = (x.to(dev), y.to(dev), torch.full((B, S), 0.1, dtype=D, device=dev), nz.to(dev), None, lengths, torch.stack(ws_).to(dev), torch.stack(mus).to(dev),
             torch.stack(vs).to(dev), order, 0.0, True)
        one = _hip.mll_value_grad_ragged(*a)
        torch.cuda.synchronize()
        keep = one["mll"].clone()
        b = ri(0, B - 1); n = lengths[b]
        fresh = _hip.Workspace(dev, n, q, d, 1)
        s = _hip.mll_value_grad(x[b, :n].to(dev), y[b, :n].to(dev), torch.full((n,), 0.1, dtype=D, device=dev), nz[b, :n].to(dev), None, ws_[b].to(dev), mus[b].to(dev),
                                vs[b].to(dev), order, 0.0, True, workspace=fresh)
        torch.cuda.synchronize()
        if not (float(keep[b]) == float(s["mll"]) and int(one["info"].abs().max()) == 0):
            bad += 1
            print(f"call {c} ragged B={B} member {b} n={n}: {float(keep[b])!r} in the set, {float(s['mll'])!r} alone on a fresh workspace")
        fresh.close()
    elif kind == "fit":
        n = size() if ri(0, 1) else ri(10, 128)
        x, y, nz, w, mu, v = curve(n, 1, q)
        xd, yd = x[:, 0].to(dev), y.to(dev)
        lik = g.likelihoods.FixedNoiseGaussianLikelihood(nz.to(dev))

        class Model(g.models.ExactGP):
            def __init__(self):
                super().__init__(xd, yd, lik)
                self.mean_module = g.means.ConstantMean(); self.covar_module = g.kernels.SpectralMixtureKernel(num_mixtures=q)

            def forward(self, xx):
                return g.distributions.MultivariateNormal(self.mean_module(xx), self.covar_module(xx))

        m = Model().double().to(dev)
        m.initialize(**{"covar_module.mixture_weights": w.to(dev), "covar_module.mixture_means": mu.reshape(q, 1, 1).to(dev), "covar_module.mixture_scales": v.reshape(q, 1, 1).to(dev)})
        iters = ri(3, 40)
        try:
            r = train_native(model=m, likelihood=lik, train_x=xd, train_y=yd, maxiter=iters, lr=0.01, optim="AdamW", stop=None)
            if not np.isfinite(np.array(r["loss"], dtype=float)).all():
                raise RuntimeError("non-finite loss in the log")
        except Exception as exc:
            bad += 1
            print(f"call {c} fit n={n} q={q} iters={iters}: {type(exc).__name__}: {exc}")
            torch.save({"x": x, "y": y, "nz": nz, "w": w, "mu": mu, "v": v, "iters": iters}, os.path.join(ROOT, "gpurun_out", f"soak_fit_case_{seed}_{c}.pt"))
            # the same start through a plain evaluation and through the host loop
            e = _hip.mll_value_grad(x.to(dev), yd, torch.zeros(n, dtype=D, device=dev), nz.to(dev), None, w.to(dev), mu.to(dev), v.to(dev), 0, 0.0, True)
            print(f"   plain evaluation at the start values: mll {float(e['mll'])!r} info {int(e['info'])}")
    else:
        C, n = ri(1, 6), max(size(), 3)                    # (the default prior of the mean takes its scale from the data's spread: a single point has none)
        cs = [curve(n, 1, q) for _ in range(C)]
        pot = mcmc.SMPotential(torch.stack([t[0] for t in cs]).to(dev), torch.stack([t[1] for t in cs]).to(dev), torch.stack([t[2] for t in cs]).to(dev), num_mixtures=q)
        for _ in range(ri(1, 4)):
            z = rng.normal(0, 0.3, (C, pot.P)); z[:, 1 + q:1 + 2 * q] += np.log(1 / 120.0); z[:, 1 + 2 * q:1 + 3 * q] += np.log(1 / 1200.0)
            U, G = pot(z)
            if not (np.isfinite(U).all() and np.isfinite(G).all()):
                bad += 1
                print(f"call {c} potential C={C} n={n} q={q}: non-finite")
        del pot
print(f"random soak: {calls} calls ({', '.join(f'{k} {v}' for k, v in sorted(kinds.items()))}) through the workspace cache, {bad} mismatches against fresh workspaces; "
      f"worst gradient deviation {worst:.2e}; {len(_hip.cached_workspaces())} workspaces cached at the end")
sys.exit(1 if bad else 0)
