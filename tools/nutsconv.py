#!/usr/bin/env python3
"""Config 5 run to convergence on one GPU: 8 NUTS chains x (N=2048, Q=4) under the default priors of
``Lightcurve.set_default_priors`` (/root/reference/pgmuvi/lightcurve.py:3235-3330; the sampler call the reference
describes at :5964-6003), every chain advanced by ONE batched HIP evaluation per tick.

    python tools/nutsconv.py            # WARMUP=400 SAMPLES=300 NOBS=2048 CHAINS=8 TARGET=0.95 (environment)

(Target acceptance 0.95, Stan's usual answer to divergent transitions: the likelihood of a 2048-point light curve is a comb of
narrow modes in the frequencies, and at the default 0.8 about 5 % of the transitions end on an energy error > 1000 where a
trajectory leaves its mode.)

Two runs: (a) each chain on a light curve of its own (seed 5000 + chain: BASELINE config 5 as SURVEY section 8d states it) --
diagnostics per chain, split-R-hat between the two halves of a chain; (b) all chains on light curve 5000 from dispersed
starts -- split-R-hat across chains.  Reports acceptance, divergent transitions, split-R-hat, effective sample size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pgmuvi_amd import mcmc, synthetic as syn

C = int(os.environ.get("CHAINS", 8)); n = int(os.environ.get("NOBS", 2048)); S = int(os.environ.get("SAMPLES", 300)); W = int(os.environ.get("WARMUP", 400))
DEPTH = int(os.environ.get("DEPTH", 7)); TARGET = float(os.environ.get("TARGET", 0.95))
dev = torch.device("cuda:0")
amp = np.array([1.0, 0.5, 0.3, 0.2])


def start(pers, jitter_seed=None):
    """hyper-parameters of the generating model, optionally dispersed by exp(0.05 N(0,1)) per chain"""
    c = len(pers)
    mu = np.stack([1 / np.array([p, 67.0, 400.0, 31.0]) for p in pers])
    w = np.tile(amp ** 2 / 2, (c, 1))
    if jitter_seed is not None:
        rng = np.random.default_rng(jitter_seed)
        mu = mu * np.exp(0.002 * rng.standard_normal(mu.shape)); w = w * np.exp(0.2 * rng.standard_normal(w.shape))
    return {"mean_module.mean_prior": np.zeros(c), "covar_module.mixture_weights_prior": w,
            "covar_module.mixture_means_prior": mu.reshape(c, 4, 1, 1), "covar_module.mixture_scales_prior": (0.1 * mu).reshape(c, 4, 1, 1)}


def report(tag, out, dt, per_chain):
    d = out["_diagnostics"]
    z = d["unconstrained"]                                   # (C, S, P)
    print(f"== {tag}: {C} chains x N={n}, {W} warm-up + {S} draws, max tree depth {DEPTH}, target acceptance {TARGET}: {dt:.1f} s")
    print(f"   acceptance {d['accept_prob'].mean():.3f} (per chain {np.round(d['accept_prob'].mean(axis=1), 2)})")
    print(f"   divergent transitions {d['divergent'].mean() * 100:.2f} % (per chain {np.round(d['divergent'].mean(axis=1) * 100, 1)})")
    print(f"   leapfrog steps per draw {d['n_leapfrog'].mean():.1f}; step sizes {np.round(d['step_size'], 4)}")
    if per_chain:
        rh = np.stack([mcmc.split_rhat(z[c:c + 1]) for c in range(C)])       # halves of one chain
        ess = np.stack([mcmc.effective_sample_size(z[c:c + 1]) for c in range(C)])
        print(f"   split-R-hat within each chain (two halves), worst parameter per chain: {np.round(rh.max(axis=1), 3)}")
        print(f"   effective sample size per chain (of {S}), worst / median parameter: {np.round(ess.min(axis=1), 0)} / {np.round(np.median(ess, axis=1), 0)}")
    else:
        rh = mcmc.split_rhat(z); ess = mcmc.effective_sample_size(z)
        print(f"   split-R-hat across the {C} chains per parameter: {np.round(rh, 3)}")
        print(f"   effective sample size per parameter (of {C * S}): {np.round(ess, 0)}")
    return d


xs, ys, ns, pers = [], [], [], []
for c in range(C):
    (t, y, e), per = syn.cfg3_lightcurve(5000 + c, n_obs=n)
    xs.append(t.double().reshape(n, 1)); ys.append(y.double()); ns.append(e.double() ** 2); pers.append(per)
x, y, nz = (torch.stack(a).to(dev) for a in (xs, ys, ns))
t0 = time.perf_counter()
out = mcmc.run_mcmc(x, y, nz, num_mixtures=4, num_samples=S, warmup_steps=W, seed=0, initial_values=start(pers), group_by_chain=True,
                    max_tree_depth=DEPTH, init_metric="curvature", target_accept_prob=TARGET)
d = report("(a) one light curve per chain", out, time.perf_counter() - t0, True)
f = out["covar_module.mixture_means_prior"].reshape(C, S, 4)
print("   posterior median leading period per chain:", np.round(np.median(1 / f[:, :, 0], axis=1), 2), " generating:", np.round(pers, 2))
x1, y1, n1 = x[:1].expand(C, -1, -1).contiguous(), y[:1].expand(C, -1).contiguous(), nz[:1].expand(C, -1).contiguous()
t0 = time.perf_counter()
out = mcmc.run_mcmc(x1, y1, n1, num_mixtures=4, num_samples=S, warmup_steps=W, seed=1, initial_values=start([pers[0]] * C, jitter_seed=11),
                    group_by_chain=True, max_tree_depth=DEPTH, init_metric="curvature", target_accept_prob=TARGET)
report("(b) all chains on light curve 5000, dispersed starts", out, time.perf_counter() - t0, False)
