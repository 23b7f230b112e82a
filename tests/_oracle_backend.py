"""TEST-ONLY stand-in for ``pgmuvi_amd._hip.mll_value_grad`` built on the CPU oracle.

Used (via unittest.mock.patch) by the ``not gpu`` tests to exercise the host logic --
the GPyTorch-shaped surface, the autograd node, the trainer, the batch sharding and the
drop-in with the reference's own ``pgmuvi`` package -- in a container without a GPU.
The product never imports this; GPU tests compare the real HIP path with the oracle.
"""
import torch

from oracle import sm_mll_oracle as orc


def mll_value_grad(x, y, mean, noise, noise_scalar, w, mu, v, dim_order=0, jitter=0.0, need_grad=True, workspace=None):
    batched = y.dim() == 2
    B = y.shape[0] if batched else 1
    n = y.shape[-1]
    D = torch.float64
    xd = x.detach().to(D).reshape(B, n, -1)
    d = xd.shape[-1]
    q = w.shape[-1]
    yd = y.detach().to(D).reshape(B, n)
    md = mean.detach().to(D).expand(y.shape).reshape(B, n)
    nz = None if noise is None else noise.detach().to(D).expand(y.shape).reshape(B, n)
    ns = None if noise_scalar is None else torch.as_tensor(noise_scalar).detach().to(D).expand(B).reshape(B)
    wd, mud, vd = w.detach().to(D).reshape(B, q), mu.detach().to(D).reshape(B, q, d), v.detach().to(D).reshape(B, q, d)
    keys = ["mll", "info", "g_w", "g_mu", "g_v", "g_noise", "g_mean"]
    res = {k: [] for k in keys}
    for b in range(B):
        nv = torch.zeros(n, dtype=D) if nz is None else nz[b]
        if ns is not None:
            nv = nv + ns[b]
        try:
            val, g = orc.mll_value_grad_closed_form(xd[b], yd[b], md[b], nv, wd[b], mud[b], vd[b], dim_order, jitter)
            info = 0
        except Exception:                              # torch.linalg.cholesky failure = non-PD
            val = torch.tensor(float("nan"), dtype=D)
            g = dict(w=torch.zeros(q, dtype=D), mu=torch.zeros(q, d, dtype=D), v=torch.zeros(q, d, dtype=D),
                     noise=torch.zeros(n, dtype=D), mean=torch.zeros(n, dtype=D))
            info = 1
        res["mll"].append(val.reshape(())); res["info"].append(torch.tensor(info, dtype=torch.int32))
        res["g_w"].append(g["w"]); res["g_mu"].append(g["mu"].reshape(q, d)); res["g_v"].append(g["v"].reshape(q, d))
        res["g_noise"].append(g["noise"].reshape(n)); res["g_mean"].append(g["mean"])
    out = {k: torch.stack(vs) for k, vs in res.items()}
    if not batched:
        out = {k: t[0] for k, t in out.items()}
    out = {k: t.to(y.device) for k, t in out.items()}
    out["workspace"] = None
    return out


def mll_value_grad_ragged(x, y, mean, noise, noise_scalar, lengths, w, mu, v, dim_order=0, jitter=0.0, need_grad=True,
                          workspace=None, max_batch=None):
    """TEST-ONLY stand-in for ``pgmuvi_amd._hip.mll_value_grad_ragged``: every light curve on its own, by the oracle."""
    B, S = y.shape
    D = torch.float64
    q = w.shape[-1]
    d = x.reshape(B, S, -1).shape[-1]
    out = dict(mll=torch.zeros(B, dtype=D), info=torch.zeros(B, dtype=torch.int32), g_w=torch.zeros(B, q, dtype=D),
               g_mu=torch.zeros(B, q, d, dtype=D), g_v=torch.zeros(B, q, d, dtype=D), g_noise=torch.zeros(B, S, dtype=D),
               g_mean=torch.zeros(B, S, dtype=D))
    for b, n in enumerate(lengths):
        o = mll_value_grad(x.reshape(B, S, d)[b, :n], y[b, :n], mean.expand(B, S)[b, :n], None if noise is None else noise[b, :n],
                           None if noise_scalar is None else noise_scalar[b], w[b], mu[b], v[b], dim_order, jitter, need_grad)
        out["mll"][b] = o["mll"]; out["info"][b] = o["info"]
        out["g_w"][b] = o["g_w"]; out["g_mu"][b] = o["g_mu"]; out["g_v"][b] = o["g_v"]
        out["g_noise"][b, :n] = o["g_noise"]; out["g_mean"][b, :n] = o["g_mean"]
    out["workspace"] = None
    return out


class _Remember:
    """What the last stand-in evaluation was called with (for the predict stand-in)."""
    last = None


_orig_mll_value_grad = mll_value_grad


def mll_value_grad_remember(*a, **k):
    _Remember.last = (a, k)
    return _orig_mll_value_grad(*a, **k)


def predict(ws, x_test, mean_test):
    """TEST-ONLY stand-in for ``pgmuvi_amd._hip.predict`` (posterior by the oracle)."""
    (x, y, mean, noise, noise_scalar, w, mu, v, dim_order, *_), _k = _Remember.last
    D = torch.float64
    n = y.shape[-1]
    nv = torch.zeros(n, dtype=D) if noise is None else noise.detach().to(D).reshape(n)
    if noise_scalar is not None:
        nv = nv + torch.as_tensor(noise_scalar).detach().to(D)
    q = w.shape[-1]
    xd = x.detach().to(D).reshape(n, -1)
    d = xd.shape[-1]
    xs = x_test.detach().to(D).reshape(x_test.shape[0], -1)
    pm, pv = orc.posterior(xd, y.detach().to(D), mean.detach().to(D).expand(n), nv, w.detach().to(D).reshape(q),
                           mu.detach().to(D).reshape(q, d), v.detach().to(D).reshape(q, d), xs,
                           mean_test.detach().to(D).expand(xs.shape[0]), dim_order)
    return pm, pv


def lomb_scargle(t, y, dy, freq, fit_mean=True, center_data=True):
    """TEST-ONLY stand-in for ``pgmuvi_amd._hip.lomb_scargle`` (periodogram by the numpy oracle)."""
    import numpy as np
    from oracle import ls_oracle
    B = y.shape[0]
    f = freq.detach().cpu().numpy()
    out = np.stack([ls_oracle.power(t[b].detach().cpu().numpy(), y[b].detach().cpu().numpy(),
                                    None if dy is None else dy[b].detach().cpu().numpy(), f, fit_mean, center_data) for b in range(B)])
    return torch.as_tensor(out, dtype=torch.float64, device=y.device)


def lomb_scargle_fast(t, y, dy, f0, df, nf, fit_mean=True, center_data=True, oversampling=5):
    """TEST-ONLY stand-in for ``pgmuvi_amd._hip.lomb_scargle_fast`` (the FFT approximation, by the numpy oracle)."""
    import numpy as np
    from oracle import ls_oracle
    if y.dim() == 1:
        t, y = t.reshape(1, -1), y.reshape(1, -1)
        dy = None if dy is None else dy.reshape(1, -1)
    t = t.expand(y.shape)
    out = np.stack([ls_oracle.power_fast(t[b].detach().cpu().numpy(), y[b].detach().cpu().numpy(),
                                         None if dy is None else dy[b].detach().cpu().numpy(), f0, df, nf, fit_mean, center_data)
                    for b in range(y.shape[0])])
    return torch.as_tensor(out, dtype=torch.float64, device=y.device)


def lomb_scargle_fast_by_exact_sums(t, y, dy, f0, df, nf, fit_mean=True, center_data=True, oversampling=5):
    """TEST-ONLY: the same entry point answered with the exact sums (to show what the approximation changes)."""
    import numpy as np
    return lomb_scargle(t.reshape(1, -1) if y.dim() == 1 else t.expand(y.shape), y.reshape(1, -1) if y.dim() == 1 else y,
                        None if dy is None else (dy.reshape(1, -1) if y.dim() == 1 else dy),
                        torch.as_tensor(f0 + df * np.arange(nf), dtype=torch.float64), fit_mean, center_data)


def mll_kernel_value_grad(x, y, mean, noise, noise_scalar, program, theta, jitter=0.0, need_grad=True, workspace=None):
    """TEST-ONLY stand-in for ``pgmuvi_amd._hip.mll_kernel_value_grad``: the program's matrix by the oracle's torch
    formulas, value by a torch Cholesky, every gradient by autograd."""
    D = torch.float64
    n = y.shape[-1]
    leaves = [(k, m, p) for (k, m, _), p in zip(program.leaves, program.leaf_par)]
    terms = [(lv, sc) for (lv, _), sc in zip(program.terms, program.term_scales)]
    th = theta.detach().to(D).reshape(-1).clone().requires_grad_(True)
    mu = mean.detach().to(D).expand(n).clone().requires_grad_(True)
    nv = (torch.zeros(n, dtype=D) if noise is None else noise.detach().to(D).expand(n).clone())
    if noise_scalar is not None:
        nv = nv + torch.as_tensor(noise_scalar).detach().to(D)
    nv = nv.clone().requires_grad_(True)
    try:
        with torch.enable_grad():
            K = orc.kernel_program_matrix(leaves, terms, th, x.detach().to(D).reshape(n, -1))
            val = orc.mll_dense(K, y.detach().to(D), mu, nv, jitter)
            val.backward()
        out = dict(mll=val.detach(), g_theta=th.grad, g_noise=nv.grad, g_mean=mu.grad, info=torch.tensor(0, dtype=torch.int32))
    except torch.linalg.LinAlgError:
        z = torch.zeros
        out = dict(mll=torch.tensor(float("nan"), dtype=D), g_theta=z(th.numel(), dtype=D), g_noise=z(n, dtype=D), g_mean=z(n, dtype=D),
                   info=torch.tensor(1, dtype=torch.int32))
    out = {k: t.to(y.device) for k, t in out.items()}
    out["workspace"] = None
    return out


def mll_dense(A, r, jitter=0.0, need_grad=True, workspace=None):
    """TEST-ONLY stand-in for ``pgmuvi_amd._hip.mll_dense``: value by a torch Cholesky, gradients by autograd."""
    batched = A.dim() == 3
    Ab = A.detach().to(torch.float64).reshape(-1, A.shape[-1], A.shape[-1])
    rb = r.detach().to(torch.float64).reshape(Ab.shape[0], -1)
    n = Ab.shape[-1]
    vals, gas, grs, infos = [], [], [], []
    for b in range(Ab.shape[0]):
        a = Ab[b].clone().requires_grad_(True)
        rr = rb[b].clone().requires_grad_(True)
        try:
            with torch.enable_grad():                     # (called from inside an autograd.Function.forward)
                L = torch.linalg.cholesky(a + jitter * torch.eye(n, dtype=torch.float64))
                z = torch.linalg.solve_triangular(L, rr.reshape(n, 1), upper=False)
                val = -0.5 * ((z * z).sum() + 2.0 * torch.log(torch.diagonal(L)).sum() + n * orc.LOG_2PI) / n
                val.backward()
            ga = 0.5 * (a.grad + a.grad.T)
            vals.append(val.detach()); gas.append(ga); grs.append(rr.grad); infos.append(0)
        except torch.linalg.LinAlgError:              # non-PD
            vals.append(torch.tensor(float("nan"), dtype=torch.float64)); gas.append(torch.zeros(n, n, dtype=torch.float64))
            grs.append(torch.zeros(n, dtype=torch.float64)); infos.append(1)
    out = dict(mll=torch.stack(vals), g_a=torch.stack(gas), g_r=torch.stack(grs), info=torch.tensor(infos, dtype=torch.int32))
    if not batched:
        out = {k: t[0] for k, t in out.items()}
    out = {k: t.to(A.device) for k, t in out.items()}
    out["workspace"] = ("dense", A.detach().clone(), r.detach().clone())
    return out


def predict_dense(ws, k_star, k_ss, mean_test):
    """TEST-ONLY stand-in for ``pgmuvi_amd._hip.predict_dense``."""
    _tag, A, r = ws
    D = torch.float64
    L = torch.linalg.cholesky(A.to(D))
    B = torch.linalg.solve_triangular(L, k_star.detach().to(D), upper=False)
    z = torch.linalg.solve_triangular(L, r.to(D).reshape(-1, 1), upper=False)
    m = k_star.shape[1]
    return mean_test.detach().to(D).expand(m) + (B.T @ z).reshape(-1), k_ss.detach().to(D).expand(m) - (B * B).sum(0)
