"""NUTS / HMC over the spectral-mixture hyper-parameters.

What the reference intends (``/root/reference/pgmuvi/lightcurve.py:5882-6075``: ``Lightcurve.mcmc`` --
currently disabled there with ``NotImplementedError`` -- on the priors of
``set_default_priors``, ``lightcurve.py:3235-3330``) is a pyro NUTS run whose model samples the
*constrained* hyper-parameters from their priors and conditions on the exact-GP marginal likelihood.
pyro is not installable here, so the sampler is in this file; the expensive part, one value+gradient of
the marginal likelihood per leapfrog step, is the same fused HIP evaluation ``fit()`` uses.

MI355X-first shape of the computation: every chain is a Python *coroutine* that yields the position at
which it needs ``(U, dU/dz)`` and is resumed with the answer; the driver advances all chains of the rank
in lock step and answers all requests of a tick with ONE batched C-ABI call (batch on ``gridDim.z``), so
B chains cost about one launch sequence per leapfrog step instead of B (the batched sweep also runs at
twice the MFMA efficiency of a single factorisation).  Chains never wait for each other's trees: a chain
that finishes a tree simply starts its next one at the following tick.

Potential, in the unconstrained coordinates pyro would use (``biject_to(prior.support)``: identity for the
Normal site, ``exp`` for the LogNormal sites, Jacobian included)::

    U(z) = -[ N * mll(theta(z)) + sum_sites log p(theta_s) + sum_positive z_s ]

with ``N * mll`` the total log marginal likelihood.  For a LogNormal(m, s) site ``log p(theta) + z`` is the
Normal(m, s) log-density of ``z = log theta``.

The sampler is the multinomial NUTS of Betancourt (2017) with the generalised U-turn criterion including
the between-subtree checks, dual-averaging step size (Nesterov/Hoffman-Gelman, target 0.8) and a windowed
diagonal metric, i.e. the published Stan/pyro algorithm; site names follow pyro's
(``"<module>.<prior name>"``, e.g. ``covar_module.mixture_means_prior``, which is what the reference's
post-processing reads, ``lightcurve.py:6046-6060``).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _hip

_LOG_2PI = math.log(2.0 * math.pi)


# --------------------------------------------------------------------------------------------
# potential of the SM exact GP with the reference's default priors
# --------------------------------------------------------------------------------------------
@dataclass
class Site:
    """One prior site: ``kind`` 'real' (Normal prior on the value) or 'positive' (LogNormal prior;
    sampled as z = log value).  ``loc``/``scale`` broadcast against (B, size)."""
    name: str
    kind: str
    shape: Tuple[int, ...]
    loc: np.ndarray
    scale: np.ndarray

    @property
    def size(self) -> int:
        return int(np.prod(self.shape)) if len(self.shape) else 1


class SMPotential:
    """``U(z)`` and its gradient for B chains at once (each chain may have its own light curve).

    x (B,N,d)|(B,N), y (B,N), noise (B,N) fixed per-point variances or None (then a noise variance is
    learned, ``likelihood.noise_covar.noise_prior``).  Per-chain parameter vector
    ``z = [c, log w (Q), log mu (Q*d), log v (Q*d) (, log sigma^2)]``.
    """

    def __init__(self, x, y, noise=None, num_mixtures=4, dim_order=0, priors: Optional[Dict[str, Tuple[float, float]]] = None,
                 compute: Optional[Callable] = None):
        if y.dim() == 1:
            x, y = x.unsqueeze(0), y.unsqueeze(0)
            noise = None if noise is None else noise.unsqueeze(0)
        self.B, self.N = y.shape
        self.x = x.reshape(self.B, self.N, -1).to(torch.float64)
        self.d = self.x.shape[-1]
        self.y = y.to(torch.float64)
        self.noise = None if noise is None else noise.to(torch.float64).expand(self.B, self.N)
        self.Q = int(num_mixtures)
        self.dim_order = dim_order
        self.device = y.device
        self._compute = compute or _hip.mll_value_grad
        # (on the GPU, without a stand-in evaluation: transforms, priors, Jacobians and the chain rule run on the device too --
        #  ``pgm_pot_*``, one graph replay per tick; made on the first call)
        self._native = None
        self._use_native = compute is None and self.device.type == "cuda"
        Q, d, B = self.Q, self.d, self.B
        ymean = self.y.mean(-1).cpu().numpy().reshape(B, 1)
        ystd = self.y.std(-1).cpu().numpy().reshape(B, 1)
        pri = dict(priors or {})
        zero, one = np.zeros((1, 1)), np.ones((1, 1))

        def ln(key):                                     # LogNormal(0, 1) unless overridden (lightcurve.py:3293-3322)
            m, s = pri.get(key, (0.0, 1.0))
            return np.asarray(m, dtype=float) + zero, np.asarray(s, dtype=float) * one

        cm, cs = pri.get("mean", (ymean, ystd / 10.0))     # Normal(mean y, std y / 10) (lightcurve.py:3280-3283)
        self.sites: List[Site] = [
            Site("mean_module.mean_prior", "real", (), np.asarray(cm, dtype=float).reshape(-1, 1), np.asarray(cs, dtype=float).reshape(-1, 1)),
            Site("covar_module.mixture_weights_prior", "positive", (Q,), *ln("mixture_weights")),
            Site("covar_module.mixture_means_prior", "positive", (Q, 1, d), *ln("mixture_means")),
            Site("covar_module.mixture_scales_prior", "positive", (Q, 1, d), *ln("mixture_scales")),
        ]
        if self.noise is None:                           # LogNormal(log s, s), s = 1e-4 std(y) (lightcurve.py:3268-3276)
            s = 1e-4 * ystd
            nm, nsd = pri.get("noise", (np.log(s), s))
            self.sites.append(Site("likelihood.noise_covar.noise_prior", "positive", (1,),
                                   np.asarray(nm, dtype=float).reshape(-1, 1), np.asarray(nsd, dtype=float).reshape(-1, 1)))
        self.P = sum(s.size for s in self.sites)
        loc = np.concatenate([np.broadcast_to(s.loc, (B, s.size)) for s in self.sites], axis=1)
        scale = np.concatenate([np.broadcast_to(s.scale, (B, s.size)) for s in self.sites], axis=1)
        self._loc = torch.as_tensor(loc, dtype=torch.float64, device=self.device)
        self._scale = torch.as_tensor(scale, dtype=torch.float64, device=self.device)
        self._loc_h, self._scale_h = np.array(loc, dtype=float), np.array(scale, dtype=float)
        self.evaluations = 0

    # ---- coordinates ---------------------------------------------------------------------
    def slices(self):
        o, out = 0, {}
        for s in self.sites:
            out[s.name] = slice(o, o + s.size)
            o += s.size
        return out

    def constrain(self, z: np.ndarray) -> Dict[str, np.ndarray]:
        """Unconstrained vectors (..., P) -> dict of constrained site values (..., *site.shape)."""
        out = {}
        for s in self.sites:
            part = z[..., self.slices()[s.name]]
            val = part if s.kind == "real" else np.exp(part)
            out[s.name] = val.reshape(z.shape[:-1] + s.shape)
        return out

    def unconstrain(self, values: Dict[str, np.ndarray]) -> np.ndarray:
        parts = []
        for s in self.sites:
            v = np.asarray(values[s.name], dtype=float).reshape(self.B, s.size)
            parts.append(v if s.kind == "real" else np.log(v))
        return np.concatenate(parts, axis=1)

    def prior_draw(self, rng: np.random.Generator) -> np.ndarray:
        """z drawn from the priors (what pyro's ``init_to_sample`` would do)."""
        return self._loc.cpu().numpy() + self._scale.cpu().numpy() * rng.standard_normal((self.B, self.P))

    # ---- the evaluation --------------------------------------------------------------------
    def __call__(self, z: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        """(B,P) host array -> (U (B,), dU/dz (B,P)); one batched device evaluation, one synchronisation.

        The transforms, the chain rule back to z and the prior terms are O(B P) and run on the host in numpy; the
        device sees one upload (the constrained parameters, packed), the fused evaluation, one packing of its outputs
        and one download -- a tick is latency-bound, and every small device op it does not issue is ~10 us saved."""
        B, Q, d, N = self.B, self.Q, self.d, self.N
        z = np.ascontiguousarray(z, dtype=np.float64)
        if self._use_native:
            if self._native is None:
                self._native = _hip.NativePotential(self.x, self.y, self.noise, Q, self.dim_order, self._loc_h, self._scale_h)
            U, g, _info = self._native(z)
            self.evaluations += B
            return U, g
        theta = z.copy()
        theta[:, 1:] = np.exp(z[:, 1:])                                 # every site but the mean is sampled as a logarithm
        tt = torch.from_numpy(theta).to(self.device)
        c = tt[:, 0]
        w = tt[:, 1:1 + Q]
        mu = tt[:, 1 + Q:1 + Q + Q * d].reshape(B, Q, d)
        v = tt[:, 1 + Q + Q * d:1 + Q + 2 * Q * d].reshape(B, Q, d)
        ns = tt[:, 1 + Q + 2 * Q * d] if self.noise is None else None
        out = self._compute(self.x, self.y, c[:, None].expand(B, N), self.noise, ns, w, mu, v, self.dim_order, 0.0, True)
        self.evaluations += B
        parts = [out["mll"].reshape(B, 1), out["info"].reshape(B, 1).to(torch.float64), out["g_mean"].sum(-1, keepdim=True),
                 out["g_w"].reshape(B, Q), out["g_mu"].reshape(B, Q * d), out["g_v"].reshape(B, Q * d)]
        if ns is not None:
            parts.append(out["g_noise"].sum(-1, keepdim=True))
        host = torch.cat(parts, dim=1).cpu().numpy()                     # the tick's one synchronisation
        mll, info, gtheta = host[:, 0], host[:, 1], host[:, 2:]
        gl = gtheta.copy()
        gl[:, 1:] *= theta[:, 1:]                                       # d/dz = theta d/dtheta for the log sites
        t = (z - self._loc_h) / self._scale_h
        lp = (-0.5 * t * t - np.log(self._scale_h) - 0.5 * _LOG_2PI).sum(-1)
        U = -(N * mll + lp)
        g = -(N * gl - t / self._scale_h)
        bad = (info != 0) | ~np.isfinite(U)
        U = np.where(bad, np.inf, U)
        g = np.where(bad[:, None], 0.0, g)
        return U, g


# --------------------------------------------------------------------------------------------
# sampler coroutines: ``(U, grad) = yield z``
# --------------------------------------------------------------------------------------------
def _leapfrog(z, r, g, eps, minv):
    r = r - 0.5 * eps * g
    z = z + eps * minv * r
    U, g = yield z
    r = r - 0.5 * eps * g
    return z, r, U, g


def _no_turn(ps_a, ps_b, rho):
    return float(np.dot(ps_a, rho)) > 0.0 and float(np.dot(ps_b, rho)) > 0.0


@dataclass
class _Tree:
    z_end: np.ndarray
    r_end: np.ndarray
    g_end: np.ndarray
    r_begin: np.ndarray
    rho: np.ndarray
    z_prop: np.ndarray
    U_prop: float
    g_prop: np.ndarray
    log_w: float
    ok: bool
    divergent: bool
    sum_acc: float
    n: int


def _build_tree(z, r, g, depth, eps, H0, minv, rng, max_dh=1000.0):
    """Extends the trajectory by 2**depth leapfrog steps of (signed) size eps from the edge (z, r, g)."""
    if depth == 0:
        z1, r1, U1, g1 = yield from _leapfrog(z, r, g, eps, minv)
        h = U1 + 0.5 * float(np.dot(minv * r1, r1))
        if not np.isfinite(h):
            h = float("inf")
        dH = H0 - h
        div = -dH > max_dh
        acc = 1.0 if dH > 0 else math.exp(dH) if dH > -700 else 0.0
        return _Tree(z1, r1, g1, r1, r1.copy(), z1, U1, g1, dH if np.isfinite(dH) else -float("inf"), not div, div, acc, 1)
    a = yield from _build_tree(z, r, g, depth - 1, eps, H0, minv, rng, max_dh)
    if not a.ok:
        return a
    b = yield from _build_tree(a.z_end, a.r_end, a.g_end, depth - 1, eps, H0, minv, rng, max_dh)
    n, acc = a.n + b.n, a.sum_acc + b.sum_acc
    if not b.ok:
        b.n, b.sum_acc, b.r_begin = n, acc, a.r_begin
        return b
    log_w = float(np.logaddexp(a.log_w, b.log_w))
    take_b = b.log_w > log_w or rng.random() < math.exp(b.log_w - log_w)
    zp, Up, gp = (b.z_prop, b.U_prop, b.g_prop) if take_b else (a.z_prop, a.U_prop, a.g_prop)
    rho = a.rho + b.rho
    ok = _no_turn(minv * a.r_begin, minv * b.r_end, rho)
    ok = ok and _no_turn(minv * a.r_begin, minv * b.r_begin, a.rho + b.r_begin)
    ok = ok and _no_turn(minv * a.r_end, minv * b.r_end, b.rho + a.r_end)
    return _Tree(b.z_end, b.r_end, b.g_end, a.r_begin, rho, zp, Up, gp, log_w, ok, False, acc, n)


def _nuts_transition(z, U, g, eps, minv, rng, max_depth):
    """One NUTS transition from (z, U, g); returns the new state and the transition's statistics."""
    r0 = rng.standard_normal(z.shape[0]) / np.sqrt(minv)
    H0 = U + 0.5 * float(np.dot(minv * r0, r0))
    zb, rb, gb = z, r0, g                      # backward-most edge of the trajectory
    zf, rf, gf = z, r0, g                      # forward-most edge
    zs, Us, gs = z, U, g                       # current multinomial sample
    rho = r0.copy()
    log_w, depth, n_leap, sum_acc, divergent = 0.0, 0, 0, 0.0, False
    while depth < max_depth:
        fwd = rng.random() < 0.5
        rb_old, rf_old, rho_old = rb, rf, rho
        if fwd:
            sub = yield from _build_tree(zf, rf, gf, depth, eps, H0, minv, rng)
            zf, rf, gf = sub.z_end, sub.r_end, sub.g_end
        else:
            sub = yield from _build_tree(zb, rb, gb, depth, -eps, H0, minv, rng)
            zb, rb, gb = sub.z_end, sub.r_end, sub.g_end
        n_leap += sub.n
        sum_acc += sub.sum_acc
        if not sub.ok:
            divergent = sub.divergent
            break
        depth += 1
        if sub.log_w > log_w or rng.random() < math.exp(sub.log_w - log_w):
            zs, Us, gs = sub.z_prop, sub.U_prop, sub.g_prop
        log_w = float(np.logaddexp(log_w, sub.log_w))
        rho = rho_old + sub.rho
        # generalised criterion over the merged trajectory, and between the old trajectory and the new subtree
        ok = _no_turn(minv * rb, minv * rf, rho)
        if fwd:          # [rb_old .. rf_old] + [sub.r_begin .. sub.r_end]
            ok = ok and _no_turn(minv * rb_old, minv * sub.r_begin, rho_old + sub.r_begin)
            ok = ok and _no_turn(minv * rf_old, minv * sub.r_end, sub.rho + rf_old)
        else:            # [sub.r_end .. sub.r_begin] + [rb_old .. rf_old]
            ok = ok and _no_turn(minv * sub.r_end, minv * rb_old, sub.rho + rb_old)
            ok = ok and _no_turn(minv * sub.r_begin, minv * rf_old, rho_old + sub.r_begin)
        if not ok:
            break
    stats = dict(depth=depth, n_leapfrog=n_leap, accept_prob=sum_acc / max(n_leap, 1), divergent=divergent, energy=H0)
    return zs, Us, gs, stats


def _hmc_transition(z, U, g, eps, minv, rng, num_steps):
    """Plain HMC: ``num_steps`` leapfrog steps, Metropolis accept."""
    r = rng.standard_normal(z.shape[0]) / np.sqrt(minv)
    H0 = U + 0.5 * float(np.dot(minv * r, r))
    z1, r1, U1, g1 = z, r, U, g
    for _ in range(num_steps):
        z1, r1, U1, g1 = yield from _leapfrog(z1, r1, g1, eps, minv)
        if not np.isfinite(U1):
            break
    h = U1 + 0.5 * float(np.dot(minv * r1, r1))
    dH = H0 - h if np.isfinite(h) else -float("inf")
    acc = 1.0 if dH > 0 else (math.exp(dH) if dH > -700 else 0.0)
    stats = dict(depth=0, n_leapfrog=num_steps, accept_prob=acc, divergent=(-dH > 1000.0), energy=H0)
    if rng.random() < acc:
        return z1, U1, g1, stats
    return z, U, g, stats


def _reasonable_step_size(z, U, g, eps, minv, rng):
    """Doubles/halves eps until the one-step acceptance probability crosses 0.8 (Hoffman & Gelman, alg. 4)."""
    r = rng.standard_normal(z.shape[0]) / np.sqrt(minv)
    H0 = U + 0.5 * float(np.dot(minv * r, r))

    def delta(e):
        z1, r1, U1, _ = yield from _leapfrog(z, r, g, e, minv)
        h = U1 + 0.5 * float(np.dot(minv * r1, r1))
        return H0 - h if np.isfinite(h) else -float("inf")

    dH = yield from delta(eps)
    direction = 1 if dH > math.log(0.8) else -1
    for _ in range(50):
        eps = eps * 2.0 if direction == 1 else eps * 0.5
        dH = yield from delta(eps)
        if (direction == 1 and not dH > math.log(0.8)) or (direction == -1 and dH > math.log(0.8)):
            break
        if eps > 1e7 or eps < 1e-12:
            break
    return eps


class _DualAveraging:
    def __init__(self, eps, delta=0.8, gamma=0.05, t0=10.0, kappa=0.75):
        self.delta, self.gamma, self.t0, self.kappa = delta, gamma, t0, kappa
        self.restart(eps)

    def restart(self, eps):
        self.mu = math.log(10.0 * eps)
        self.t, self.hbar, self.log_eps_bar, self.log_eps = 0, 0.0, 0.0, math.log(eps)

    def update(self, accept_prob):
        self.t += 1
        a = min(1.0, accept_prob) if np.isfinite(accept_prob) else 0.0
        w = 1.0 / (self.t + self.t0)
        self.hbar = (1 - w) * self.hbar + w * (self.delta - a)
        self.log_eps = self.mu - math.sqrt(self.t) / self.gamma * self.hbar
        eta = self.t ** (-self.kappa)
        self.log_eps_bar = eta * self.log_eps + (1 - eta) * self.log_eps_bar
        return math.exp(self.log_eps)

    def final(self):
        return math.exp(self.log_eps_bar)


def _adaptation_windows(warmup):
    """Ends (exclusive) of the metric-adaptation windows inside the warm-up (Stan's schedule: a fast initial
    buffer, doubling slow windows, a fast terminal buffer)."""
    if warmup < 20:
        return []
    init, term, base = 75, 50, 25
    if init + base + term > warmup:
        init, term = int(0.15 * warmup), int(0.1 * warmup)
        base = warmup - init - term
    ends, start, size = [], init, base
    while start < warmup - term:
        end = start + size
        if end + 2 * size > warmup - term:
            end = warmup - term
        ends.append(end)
        start, size = end, size * 2
    return ends


def _curvature_metric(z, U, g):
    """Diagonal inverse metric from secant curvatures at the starting point, ``1 / (d^2 U / dz_i^2)``: P extra
    evaluations.  The GP posterior of a long light curve is stiff (a log-frequency is known to ~1e-5, a log-weight to
    ~0.3): with the identity metric pyro/Stan start from, the first adaptation windows are spent at a step size set by
    the stiffest coordinate.  The probe length shrinks until the potential rises by less than ~1 (inside the bulk)."""
    P = z.shape[0]
    minv = np.ones(P)
    for i in range(P):
        delta, h = 1e-2, None
        for _ in range(24):
            zi = z.copy()
            zi[i] += delta
            Ui, gi = yield zi
            if np.isfinite(Ui) and abs(Ui - U) < 1.0:
                h = (gi[i] - g[i]) / delta
                break
            delta *= 0.25
        if h is not None and h > 1e-6:
            minv[i] = min(max(1.0 / h, 1e-14), 1e4)
    return minv


def _chain(z, eps0, num_samples, warmup, rng, sampler="NUTS", max_depth=10, target_accept=0.8, adapt_metric=True,
           num_steps=None, trajectory_length=None, init_metric="identity"):
    """Coroutine for one chain: yields positions, is sent ``(U, grad)``; returns its samples and statistics."""
    P = z.shape[0]
    U, g = yield z
    tries = 0
    while not np.isfinite(U) and tries < 100:            # like pyro: re-draw an initial point with finite potential
        z = rng.uniform(-2.0, 2.0, P)
        U, g = yield z
        tries += 1
    if not np.isfinite(U):
        raise RuntimeError("no initial point with a finite potential found (non-PD covariance at every try)")
    if init_metric not in ("identity", "curvature"):
        raise ValueError("init_metric must be 'identity' or 'curvature'")
    minv = (yield from _curvature_metric(z, U, g)) if init_metric == "curvature" else np.ones(P)
    eps = yield from _reasonable_step_size(z, U, g, eps0, minv, rng)
    da = _DualAveraging(eps, delta=target_accept)
    ends = _adaptation_windows(warmup) if adapt_metric else []
    win_start = (75 if 75 + 25 + 50 <= warmup else int(0.15 * warmup)) if ends else None
    buf: List[np.ndarray] = []
    samples = np.empty((num_samples, P))
    stats = {k: np.zeros(num_samples, dtype=t) for k, t in (("accept_prob", float), ("n_leapfrog", int), ("depth", int),
                                                            ("divergent", bool), ("potential_energy", float), ("energy", float))}
    for it in range(warmup + num_samples):
        if sampler == "NUTS":
            z, U, g, st = yield from _nuts_transition(z, U, g, eps, minv, rng, max_depth)
        else:
            L = num_steps if num_steps is not None else max(1, int(round((trajectory_length or 2 * math.pi) / eps)))
            z, U, g, st = yield from _hmc_transition(z, U, g, eps, minv, rng, min(L, 1024))
        if it < warmup:
            eps = da.update(st["accept_prob"])
            if ends and win_start <= it < ends[-1]:
                buf.append(z.copy())
            if ends and it + 1 in ends:
                arr = np.asarray(buf)
                n = arr.shape[0]
                var = arr.var(axis=0, ddof=1) if n > 1 else np.ones(P)
                minv = (n / (n + 5.0)) * var + 1e-3 * (5.0 / (n + 5.0))
                buf = []
                eps = yield from _reasonable_step_size(z, U, g, eps, minv, rng)
                da.restart(eps)
            if it + 1 == warmup:
                eps = da.final() if da.t > 0 else eps
        else:
            k = it - warmup
            samples[k] = z
            stats["accept_prob"][k], stats["n_leapfrog"][k], stats["depth"][k] = st["accept_prob"], st["n_leapfrog"], st["depth"]
            stats["divergent"][k], stats["potential_energy"][k], stats["energy"][k] = st["divergent"], U, st["energy"]
    return dict(samples=samples, stats=stats, step_size=eps, inverse_mass=minv)


# --------------------------------------------------------------------------------------------
# driver: all chains of this rank in lock step, one batched evaluation per tick
# --------------------------------------------------------------------------------------------
def sample(potential: Callable[[np.ndarray], Tuple[np.ndarray, np.ndarray]], z0: np.ndarray, num_samples=500, warmup_steps=100,
           sampler="NUTS", seed=0, step_size=0.1, max_tree_depth=10, target_accept_prob=0.8, adapt_mass_matrix=True,
           num_steps=None, trajectory_length=None, init_metric="identity", chain_ids: Optional[Sequence[int]] = None,
           progress: Optional[Callable] = None):
    """Runs ``B = z0.shape[0]`` chains on ``potential`` ((B,P) -> (U (B,), grad (B,P))).

    Chain b draws from ``np.random.default_rng([seed, chain_ids[b]])``, so a chain's stream does not depend on
    which rank runs it or on how many chains share the batch.  Returns ``dict(samples (B,S,P), stats, step_size (B,),
    inverse_mass (B,P), ticks)``.
    """
    if sampler not in ("NUTS", "HMC"):
        raise ValueError("sampler must be one of 'NUTS' or 'HMC'")
    z0 = np.array(z0, dtype=float)
    B, P = z0.shape
    ids = list(range(B)) if chain_ids is None else list(chain_ids)
    gens = [_chain(z0[b].copy(), step_size, num_samples, warmup_steps, np.random.default_rng([seed, ids[b]]), sampler,
                   max_tree_depth, target_accept_prob, adapt_mass_matrix, num_steps, trajectory_length, init_metric)
            for b in range(B)]
    req = np.stack([next(gen) for gen in gens])
    done: List[Optional[dict]] = [None] * B
    ticks = 0
    while any(d is None for d in done):
        U, G = potential(req)
        ticks += 1
        for b in range(B):
            if done[b] is not None:
                continue
            try:
                req[b] = gens[b].send((float(U[b]), np.array(G[b], dtype=float)))
            except StopIteration as fin:
                done[b] = fin.value
        if progress is not None:
            progress(ticks, sum(d is not None for d in done))
    keys = done[0]["stats"].keys()
    return dict(samples=np.stack([d["samples"] for d in done]), stats={k: np.stack([d["stats"][k] for d in done]) for k in keys},
                step_size=np.array([d["step_size"] for d in done]), inverse_mass=np.stack([d["inverse_mass"] for d in done]), ticks=ticks)


def split_rhat(x: np.ndarray) -> np.ndarray:
    """Split-chain potential scale reduction of draws (chains, samples, ...) (Gelman et al., BDA3 11.4)."""
    c, s = x.shape[:2]
    h = s // 2
    parts = np.concatenate([x[:, :h], x[:, s - h:]], axis=0)
    m = parts.mean(axis=1)
    w = parts.var(axis=1, ddof=1).mean(axis=0)
    b = h * m.var(axis=0, ddof=1)
    return np.sqrt(((h - 1) / h * w + b / h) / w)


def effective_sample_size(x: np.ndarray) -> np.ndarray:
    """Effective number of independent draws in (chains, samples, ...) -- Geyer's initial positive sequence on the
    chain-averaged autocorrelations, as pyro's / Stan's diagnostics estimate it (BDA3 11.5)."""
    x = np.asarray(x, dtype=float)
    c, s = x.shape[:2]
    flat = x.reshape(c, s, -1)
    out = np.empty(flat.shape[2])
    for k in range(flat.shape[2]):
        v = flat[:, :, k] - flat[:, :, k].mean(axis=1, keepdims=True)
        n = 1 << int(np.ceil(np.log2(2 * s)))
        f = np.fft.rfft(v, n=n, axis=1)
        acov = np.fft.irfft(f * np.conj(f), n=n, axis=1)[:, :s] / s          # biased autocovariances per chain
        w = acov[:, 0].mean() * s / (s - 1.0)
        var_plus = w * (s - 1.0) / s
        if c > 1:
            var_plus += flat[:, :, k].mean(axis=1).var(ddof=1)
        if not (var_plus > 0):
            out[k] = float(c * s)
            continue
        rho = 1.0 - (w - acov.mean(axis=0)) / var_plus
        tau, t = -1.0, 0
        while t + 1 < s:
            pair = rho[t] + rho[t + 1]
            if pair < 0:
                break
            tau += 2.0 * pair
            t += 2
        out[k] = c * s / max(tau, 1.0 / np.log10(max(c * s, 10)))
    return out.reshape(x.shape[2:])


def run_mcmc(x, y, noise=None, num_mixtures=4, sampler="NUTS", num_samples=500, warmup_steps=100, num_chains=1, seed=0,
             initial_values: Optional[Dict[str, np.ndarray]] = None, priors=None, dim_order=0, group=None, compute=None,
             group_by_chain=False, **sampler_kwargs):
    """The run ``Lightcurve.mcmc`` describes (``lightcurve.py:5882-6003``: sampler 'NUTS'|'HMC', ``num_samples=500``,
    ``warmup_steps=100``, ``num_chains``): posterior samples of the SM hyper-parameters under the default priors.

    ``y`` (N,) -> ``num_chains`` chains on that light curve; ``y`` (C,N) -> one chain per row (config 5: each chain its own
    light curve).  With ``torch.distributed`` initialised the chains are block-partitioned over the ranks
    (``batch.shard_bounds``), each rank advances its chains with batched evaluations on its own GPU, and one
    ``all_gather`` of the draws closes the run (the only collective).  Returns pyro-style ``{site name: draws}`` with
    chains flattened unless ``group_by_chain``, plus ``"_diagnostics"``.
    """
    import torch.distributed as dist
    from .batch import gather_logliks, shard_bounds
    if y.dim() == 1:
        C = int(num_chains)
        xs = x.reshape(1, y.shape[0], -1).expand(C, -1, -1)
        ys = y.unsqueeze(0).expand(C, -1)
        nz = None if noise is None else noise.reshape(1, -1).expand(C, -1)
    else:
        C = y.shape[0]
        xs, ys, nz = x.reshape(C, y.shape[1], -1), y, noise
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank(group) if world > 1 else 0
    lo, hi = shard_bounds(C, rank, world)
    if hi > lo:
        pot = SMPotential(xs[lo:hi].contiguous(), ys[lo:hi].contiguous(), None if nz is None else nz[lo:hi].contiguous(),
                          num_mixtures, dim_order, priors, compute)
        if initial_values is not None:      # {site: value of site.shape (all chains) or (C, *site.shape)}
            vals = {}
            for st in pot.sites:
                v = np.asarray(initial_values[st.name], dtype=float)
                v = np.broadcast_to(v.reshape((1,) + st.shape), (C,) + st.shape) if v.size == st.size else v.reshape((C,) + st.shape)
                vals[st.name] = v[lo:hi]
            z0 = pot.unconstrain(vals)
        else:                                # pyro's default: uniform(-2, 2) in the unconstrained space
            z0 = np.stack([np.random.default_rng([seed, 7919, c]).uniform(-2.0, 2.0, pot.P) for c in range(lo, hi)])
            z0[:, 0] += np.broadcast_to(pot.sites[0].loc, (hi - lo, 1))[:, 0]
        res = sample(pot, z0, num_samples, warmup_steps, sampler, seed, chain_ids=range(lo, hi), **sampler_kwargs)
        P = pot.P
        local = np.concatenate([res["samples"].reshape(hi - lo, -1), res["stats"]["accept_prob"], res["stats"]["n_leapfrog"].astype(float),
                                res["stats"]["divergent"].astype(float), res["stats"]["potential_energy"], res["step_size"][:, None]], axis=1)
    else:
        raise RuntimeError(f"rank {rank} has no chain: use at most num_chains={C} ranks")
    sites, slices = pot.sites, pot.slices()
    if world > 1:
        dev = ys.device
        local = gather_logliks(torch.as_tensor(local, dtype=torch.float64, device=dev), C, group).cpu().numpy()
    S = num_samples
    draws = local[:, :S * P].reshape(C, S, P)
    o = S * P
    diag = dict(accept_prob=local[:, o:o + S], n_leapfrog=local[:, o + S:o + 2 * S], divergent=local[:, o + 2 * S:o + 3 * S] > 0.5,
                potential_energy=local[:, o + 3 * S:o + 4 * S], step_size=local[:, o + 4 * S])
    out = {}
    for s in sites:
        part = draws[..., slices[s.name]]
        val = (part if s.kind == "real" else np.exp(part)).reshape((C, S) + s.shape)
        out[s.name] = val if group_by_chain else val.reshape((C * S,) + s.shape)
    if C > 1 and S >= 4:
        diag["split_rhat"] = split_rhat(draws)
    diag["unconstrained"] = draws
    out["_diagnostics"] = diag
    return out
