"""Wider random checks of paths the test suite holds at a few fixed shapes (round 6, after a wider fuzz found a last-bit inconsistency):

  batches     equal-length batches of 2 .. 40 light curves x 129 .. 1500 points (fused sweep with windows, k_trsm64, panel sweep):
              every member's value is its single evaluation's, bit for bit; gradients to rounding (log-parameter metric)
  potential   pgm_pot_* (z -> theta, priors, Jacobians, chain rule on the device) against the same potential assembled on the host,
              random chains x points x mixtures x dimensions, fixed and learned noise, a few ticks each
  predict     posterior mean / variance after an evaluation, random sizes 1 .. 1500 (one launch, sixteenth tiles, fused sweep) and
              random test points, against the oracle's dense posterior

  fit         the device-resident optimiser loop (pgm_fit_*: one launch per iteration up to 128 points, fused fit launches and several
              iterations per graph replay beyond) against the reference-shaped host loop ``train``: random sizes 10 .. 1300, 1 .. 4
              mixtures, 1-D / 2-D, SGD / Adam / AdamW, fixed or learned noise, constant or linear mean, 5 .. 70 iterations

  generic     composed stationary kernels (the zoo of tests/test_gpu_parity.py: Matern / RBF / RQ / periodic / cosine / linear / constant
              under scale, product, sum; active dimensions in 2-D) at random sizes 5 .. 1600 and randomly perturbed parameters, fused
              path against the oracle's torch formulas with autograd

  ls          Lomb-Scargle periodograms: exact sums on random grids and the FFT approximation on regular ones, random light curves of
              5 .. 3000 points (batches of 1 .. 6), with and without error bars, fit_mean / center_data, against the numpy oracle

  dense       the dense entry points: pgm_sm_kernel_f64 (rectangular K(x1, x2), both dimension orders) against the oracle's matrix, and
              pgm_mll_dense_f64 / pgm_predict_dense_f64 (a caller-built matrix through the sweep, dmll/dA by autograd on the oracle)

    python tools/lab/fuzz_more.py [batches|potential|predict|fit|generic|ls|dense|all] [cases] [seed]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pgmuvi_amd import _hip, mcmc  # noqa: E402
from oracle import sm_mll_oracle as orc  # noqa: E402

D = torch.float64
what = sys.argv[1] if len(sys.argv) > 1 else "all"
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(seed)
ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=gen))
bad = 0


def curve(n, d, q):
    x = torch.rand(n, d, generator=gen, dtype=D) * 900.0
    if d == 1:
        x = torch.sort(x[:, 0])[0].reshape(n, 1)
    else:
        x[:, 1] = torch.randint(1, 4, (n,), generator=gen).double() * 0.5
    y = torch.randn(n, generator=gen, dtype=D)
    nz = 0.01 + 0.05 * torch.rand(n, generator=gen, dtype=D)
    w = 0.1 + torch.rand(q, generator=gen, dtype=D)
    mu = 0.005 + 0.3 * torch.rand(q, d, generator=gen, dtype=D)
    v = 0.001 + 0.02 * torch.rand(q, d, generator=gen, dtype=D)
    return x, y, nz, w, mu, v


def hyper_dev(a, r, w, mu, v):
    cat = lambda g: torch.cat([g["g_w"].cpu().reshape(-1) * w.reshape(-1), g["g_mu"].cpu().reshape(-1) * mu.reshape(-1), g["g_v"].cpu().reshape(-1) * v.reshape(-1)])
    va, vr = cat(a), cat(r)
    return float((va - vr).abs().max() / (vr.abs().max() + 1e-300))


if what in ("batches", "all"):
    worst = 0.0
    for c in range(cases):
        B = ri(2, 40) if c % 3 else ri(2, 12)
        n = ri(129, 1500) if c % 2 else ri(129, 700)
        d = 1 + (c % 5 == 1); q = ri(1, 4); order = ri(0, 1) if d == 2 else 0
        cs = [curve(n, d, q) for _ in range(B)]
        st = lambda i: torch.stack([t[i] for t in cs]).to(dev)
        out = _hip.mll_value_grad(st(0), st(1), torch.full((B, n), 0.1, dtype=D, device=dev), st(2), None, st(3), st(4), st(5), order, 0.0, True)
        torch.cuda.synchronize()
        res = {k: out[k].clone() for k in ("mll", "g_w", "g_mu", "g_v", "g_noise", "g_mean", "info")}
        assert int(res["info"].abs().max()) == 0
        for b in range(B):
            x, y, nz, w, mu, v = cs[b]
            s = _hip.mll_value_grad(x.to(dev), y.to(dev), torch.full((n,), 0.1, dtype=D, device=dev), nz.to(dev), None, w.to(dev), mu.to(dev), v.to(dev), order, 0.0, True)
            torch.cuda.synchronize()
            if float(s["mll"]) != float(res["mll"][b]):
                bad += 1
                print(f"batches: call {c} (B={B} n={n} q={q} d={d} order={order}) member {b}: value {float(res['mll'][b])!r} in the batch, {float(s['mll'])!r} alone")
            worst = max(worst, hyper_dev({k: res[k][b] for k in ("g_w", "g_mu", "g_v")}, s, w, mu, v))
            worst = max(worst, float((res["g_noise"][b] - s["g_noise"]).abs().max() / s["g_noise"].abs().max()))
        _hip.release_workspaces()
    print(f"batches: {cases} calls, {bad} members' values differ from their single evaluation in their bits; worst gradient deviation {worst:.2e}")

if what in ("potential", "all"):
    rng = np.random.default_rng(seed)
    worst_u, worst_g, nb = 0.0, 0.0, 0
    for c in range(cases):
        C = ri(1, 8); n = ri(20, 900) if c % 3 else ri(20, 128); Q = ri(1, 4); d = 1 + (c % 4 == 1); learn = c % 3 == 2
        cs = [curve(n, d, Q) for _ in range(C)]
        xx = torch.stack([t[0] for t in cs]); y = torch.stack([t[1] for t in cs]); nz = torch.stack([t[2] for t in cs])
        native = mcmc.SMPotential(xx.to(dev), y.to(dev), None if learn else nz.to(dev), num_mixtures=Q)
        host = mcmc.SMPotential(xx.to(dev), y.to(dev), None if learn else nz.to(dev), num_mixtures=Q)
        host._use_native = False
        assert native._use_native
        for tick in range(3):
            z = rng.normal(0, 0.3, (C, native.P))
            z[:, 1 + Q:1 + Q + Q * d] += np.log(1 / 120.0)
            z[:, 1 + Q + Q * d:1 + Q + 2 * Q * d] += np.log(1 / 1200.0)
            if learn:
                z[:, -1] = np.log(0.02)
            Un, Gn = native(z); Uh, Gh = host(z)
            ok = np.isfinite(Un).all() and np.allclose(Un, Uh, rtol=1e-12, atol=1e-9) and np.allclose(Gn, Gh, rtol=1e-9, atol=1e-9 * n)
            worst_u = max(worst_u, float(np.abs(Un - Uh).max())); worst_g = max(worst_g, float(np.abs(Gn - Gh).max() / (np.abs(Gh).max() + 1e-300)))
            if not ok:
                nb += 1
                print(f"potential: case {c} (C={C} n={n} Q={Q} d={d} learn={learn}) tick {tick}: max |dU| {np.abs(Un - Uh).max():.3e}, gradient {np.abs(Gn - Gh).max():.3e}")
        del native, host
        _hip.release_workspaces()
    bad += nb
    print(f"potential: {cases} cases x 3 ticks, {nb} outside tolerance; worst |dU| {worst_u:.2e}, worst gradient deviation {worst_g:.2e} (relative to the largest entry)")

if what in ("predict", "all"):
    worst_m, worst_v, nb = 0.0, 0.0, 0
    for c in range(cases):
        n = ri(1, 128) if c % 3 == 0 else (ri(129, 520) if c % 3 == 1 else ri(521, 1500))
        q = ri(1, 4); m = ri(1, 700)
        x, y, nz, w, mu, v = curve(n, 1, q)
        ws = _hip.Workspace(dev, n, q, 1, 1)
        out = _hip.mll_value_grad(x.to(dev), y.to(dev), torch.full((n,), 0.2, dtype=D, device=dev), nz.to(dev), None, w.to(dev), mu.to(dev), v.to(dev), 0, 0.0, True, workspace=ws)
        assert int(out["info"]) == 0
        xs = torch.rand(m, generator=gen, dtype=D) * 1000.0 - 50.0
        pm, pv = _hip.predict(ws, xs.reshape(-1, 1).to(dev), torch.full((m,), 0.2, dtype=D, device=dev))
        torch.cuda.synchronize()
        rm, rv = orc.posterior(x[:, 0], y, 0.2, nz, w, mu, v, xs, 0.2)
        dm, dv = float((pm.cpu() - rm).abs().max()), float((pv.cpu() - rv).abs().max())
        worst_m, worst_v = max(worst_m, dm), max(worst_v, dv)
        if not (dm < 1e-8 and dv < 1e-8):
            nb += 1
            print(f"predict: case {c} (n={n} q={q} m={m}): max |d mean| {dm:.3e}, max |d var| {dv:.3e}")
        ws.close()
    bad += nb
    print(f"predict: {cases} cases, {nb} outside 1e-8; worst |d mean| {worst_m:.2e}, worst |d var| {worst_v:.2e}")

if what in ("fit", "all"):
    from pgmuvi_amd import gpytorch as g
    from pgmuvi_amd.trainers import train, train_native
    worst, nb, diverged = 0.0, 0, 0
    for c in range(cases):
        n = ri(10, 128) if c % 3 == 0 else (ri(129, 520) if c % 3 == 1 else ri(521, 1300))
        d = 1 + (c % 4 == 1); Q = ri(1, 4); learn = c % 5 == 2; mean = "linear" if c % 7 == 3 else "constant"
        optim = ("SGD", "Adam", "AdamW")[c % 3] if mean == "constant" else ("Adam", "AdamW")[c % 2]; iters = ri(5, 70); lr = 1e-4 if optim == "SGD" else 0.01
        x, y, nz, w, mu, v = curve(n, d, Q)
        xd, yd, nzd = (x[:, 0] if d == 1 else x).to(dev), y.to(dev), nz.to(dev)

        def build():
            lik = g.likelihoods.GaussianLikelihood().double().to(dev) if learn else g.likelihoods.FixedNoiseGaussianLikelihood(nzd)

            class Model(g.models.ExactGP):
                def __init__(self):
                    super().__init__(xd, yd, lik)
                    self.mean_module = g.means.ConstantMean() if mean == "constant" else g.means.LinearMean(input_size=d)
                    self.covar_module = g.kernels.SpectralMixtureKernel(num_mixtures=Q, ard_num_dims=d)

                def forward(self, xx):
                    return g.distributions.MultivariateNormal(self.mean_module(xx), self.covar_module(xx))

            m = Model().double().to(dev)
            if mean == "linear":
                with torch.no_grad():
                    m.mean_module.weights.fill_(1e-4); m.mean_module.bias.fill_(0.05)
            m.initialize(**{"covar_module.mixture_weights": w.to(dev), "covar_module.mixture_means": mu.reshape(Q, 1, d).to(dev),
                            "covar_module.mixture_scales": v.reshape(Q, 1, d).to(dev)})
            if learn:
                lik.noise = torch.tensor(0.03, dtype=D, device=dev)
            return m, lik

        m1, l1 = build(); m2, l2 = build()
        try:
            r1 = train(model=m1, likelihood=l1, train_x=xd, train_y=yd, maxiter=iters, lr=lr, optim=optim, progress=False, stop=None)
            r2 = train_native(model=m2, likelihood=l2, train_x=xd, train_y=yd, maxiter=iters, lr=lr, optim=optim, check_every=max(4, iters // 2), stop=None)
        except Exception as exc:                                 # (a diverging fit ends in a matrix that cannot be factored)
            if "positive definite" not in str(exc) and "NaN" not in str(exc):
                raise
            diverged += 1
            _hip.release_workspaces()
            continue
        a, b = np.array(r1["loss"], dtype=float), np.array(r2["loss"], dtype=float)
        if len(a) == len(b) and np.abs(a).max() > 1e6:          # (SGD on a linear mean of times up to 900: the fit itself diverges, rounding differences with it)
            diverged += 1
            _hip.release_workspaces()
            continue
        dl = float((np.abs(a - b) / np.maximum(1.0, np.abs(a))).max()) if len(a) == len(b) else float("inf")
        dp = max(float((p1.detach() - p2.detach()).abs().max() / (p2.detach().abs().max() + 1e-12)) for (_, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()))
        worst = max(worst, dl)
        if not (dl < 1e-8 and dp < 1e-7):
            nb += 1
            print(f"fit: case {c} (n={n} d={d} Q={Q} learn={learn} mean={mean} {optim} x{iters}): max |d loss| {dl:.3e}, parameters {dp:.3e}")
        _hip.release_workspaces()
    bad += nb
    print(f"fit: {cases} cases ({diverged} left out: the optimisation itself diverged), {nb} outside tolerance; worst |d loss| / max(1, |loss|) along the trajectories {worst:.2e}")

if what in ("generic", "all"):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle_backend as ob
    import test_gpu_parity as tp
    from pgmuvi_amd.gpytorch.kernels import compile_program
    worst_v, worst_g, nb = 0.0, 0.0, 0
    zoo = [(d, name, k) for d in (1, 2) for name, k in tp._kernel_zoo(d).items()]
    for c in range(cases):
        d, name, kern = zoo[c % len(zoo)]
        prog = compile_program(kern, d)
        theta = prog.theta().detach() * torch.exp(0.3 * torch.randn(prog.theta().numel(), generator=gen, dtype=D))
        n = ri(5, 128) if c % 4 == 0 else (ri(129, 700) if c % 4 < 3 else ri(701, 1600))
        x = torch.rand(n, d, generator=gen, dtype=D) * torch.tensor([300.0, 2.0][:d], dtype=D)
        x = x[torch.argsort(x[:, 0])]
        y = torch.randn(n, generator=gen, dtype=D)
        nz = 0.02 + 0.05 * torch.rand(n, generator=gen, dtype=D)
        mean = torch.full((n,), 0.1, dtype=D)
        out = _hip.mll_kernel_value_grad(x.to(dev), y.to(dev), mean.to(dev), nz.to(dev), torch.tensor(0.01, dtype=D, device=dev), prog, theta.to(dev))
        ref = ob.mll_kernel_value_grad(x, y, mean, nz, torch.tensor(0.01, dtype=D), prog, theta)
        dv = abs(float(out["mll"]) - float(ref["mll"]))
        gt = out["g_theta"].cpu() * theta; rt = ref["g_theta"] * theta                 # (log-parameter metric, one vector)
        dg = float((gt - rt).abs().max() / (rt.abs().max() + 1e-300))
        dn = float((out["g_noise"].cpu() - ref["g_noise"]).abs().max() / (ref["g_noise"].abs().max() + 1e-300))
        worst_v, worst_g = max(worst_v, dv), max(worst_g, dg, dn)
        if not (int(out["info"]) == 0 and dv < 1e-9 and dg < 1e-7 and dn < 1e-7):
            nb += 1
            print(f"generic: case {c} ({name}, d={d}, n={n}): |d mll| {dv:.3e}, theta gradient {dg:.3e}, noise gradient {dn:.3e}, info {int(out['info'])}")
        _hip.release_workspaces()
    bad += nb
    print(f"generic: {cases} cases, {nb} outside tolerance; worst |d mll| {worst_v:.2e}, worst gradient deviation {worst_g:.2e}")

if what in ("ls", "all"):
    from oracle import ls_oracle as lso
    worst, nb = 0.0, 0
    for c in range(cases):
        B = ri(1, 6); n = ri(5, 200) if c % 3 == 0 else ri(201, 3000)
        t = torch.sort(torch.rand(B, n, generator=gen, dtype=D) * ri(50, 3000), dim=1)[0]
        y = torch.randn(B, n, generator=gen, dtype=D) + 0.7 * torch.sin(2 * np.pi * t / ri(5, 400))
        with_dy = c % 2 == 0
        dy = (0.05 + 0.3 * torch.rand(B, n, generator=gen, dtype=D)) if with_dy else None
        fit_mean = c % 5 != 4; center = True if not fit_mean else (c % 3 != 2)
        nf = ri(20, 1500)
        f0, df = 1e-4 + 1e-3 * float(torch.rand((), generator=gen)), (0.5 + float(torch.rand((), generator=gen))) / (float(t.max()) * 5)
        freq = torch.as_tensor(f0 + df * np.arange(nf), dtype=D)
        out = _hip.lomb_scargle(t.to(dev), y.to(dev), None if dy is None else dy.to(dev), freq.to(dev), fit_mean, center).cpu().numpy()
        fast = _hip.lomb_scargle_fast(t.to(dev), y.to(dev), None if dy is None else dy.to(dev), f0, df, nf, fit_mean, center).cpu().numpy()
        for b in range(B):
            tb, yb, db = t[b].numpy(), y[b].numpy(), None if dy is None else dy[b].numpy()
            ref = lso.power(tb, yb, db, freq.numpy(), fit_mean, center)
            reff = lso.power_fast(tb, yb, db, f0, df, nf, fit_mean, center)
            d1, d2 = float(np.abs(out[b] - ref).max()), float(np.abs(fast[b] - reff).max())
            worst = max(worst, d1, d2)
            if not (d1 < 1e-9 and d2 < 1e-9):
                nb += 1
                print(f"ls: case {c} (B={B} n={n} nf={nf} dy={with_dy} fit_mean={fit_mean} center={center}) curve {b}: exact sums {d1:.3e}, FFT form {d2:.3e}")
    bad += nb
    print(f"ls: {cases} cases, {nb} periodograms outside 1e-9; worst deviation {worst:.2e}")

if what in ("dense", "all"):
    worst_k, worst_v, worst_g, worst_p, nb = 0.0, 0.0, 0.0, 0.0, 0
    for c in range(cases):
        n = ri(1, 128) if c % 3 == 0 else ri(129, 900); m = ri(1, 400); d = 1 + (c % 2); q = ri(1, 4); order = ri(0, 1) if d == 2 else 0
        x, y, nz, w, mu, v = curve(n, d, q)
        x2 = torch.rand(m, d, generator=gen, dtype=D) * 900.0
        if d == 2: x2[:, 1] = torch.randint(1, 4, (m,), generator=gen).double() * 0.5
        Kd = _hip.sm_kernel_dense(x.to(dev), x2.to(dev), w.to(dev), mu.to(dev), v.to(dev), None, 0.0, order).cpu()
        Kr = orc.sm_kernel(x, x2, w, mu, v, order)
        dk = float((Kd - Kr).abs().max())
        # the dense back-end on A = K(x, x) + noise
        A = (orc.sm_kernel(x, x, w, mu, v, order) + torch.diag(nz)).requires_grad_(True)
        r = (y - 0.1)
        val = orc.mll_dense(A, y, 0.1, torch.zeros(n, dtype=D))
        val.backward()
        ws = _hip.Workspace(dev, n, 1, 1, 1)
        out = _hip.mll_dense(A.detach().to(dev), r.to(dev), 0.0, True, workspace=ws)
        torch.cuda.synchronize()
        dv = abs(float(out["mll"]) - float(val))
        ga = out["g_a"].cpu().reshape(n, n); gr = A.grad
        gsym = 0.5 * (gr + gr.T)                                 # (the library returns the symmetric dmll/dA)
        dg = float((ga - gsym).abs().max() / (gsym.abs().max() + 1e-300))
        pm, pv = _hip.predict_dense(ws, Kd.to(dev), torch.full((m,), float(w.sum()) if order == 1 or d == 1 else float(w.sum()) ** d, dtype=D, device=dev), torch.full((m,), 0.1, dtype=D, device=dev))
        kss = torch.full((m,), float(w.sum()) if order == 1 or d == 1 else float(w.sum()) ** d, dtype=D)
        rm, rv = orc.posterior_dense(A.detach() - torch.diag(nz), Kr, kss, y, 0.1, nz, 0.1)
        dp = max(float((pm.cpu() - rm).abs().max()), float((pv.cpu() - rv).abs().max()))
        worst_k, worst_v, worst_g, worst_p = max(worst_k, dk), max(worst_v, dv), max(worst_g, dg), max(worst_p, dp)
        if not (dk < 1e-12 and int(out["info"]) == 0 and dv < 1e-9 and dg < 1e-7 and dp < 1e-8):
            nb += 1
            print(f"dense: case {c} (n={n} m={m} d={d} q={q} order={order}): K {dk:.3e}, mll {dv:.3e}, dmll/dA {dg:.3e}, posterior {dp:.3e}, info {int(out['info'])}")
        ws.close()
    bad += nb
    print(f"dense: {cases} cases, {nb} outside tolerance; worst |dK| {worst_k:.2e}, |d mll| {worst_v:.2e}, dmll/dA {worst_g:.2e}, posterior {worst_p:.2e}")

sys.exit(1 if bad else 0)
