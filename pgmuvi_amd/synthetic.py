"""Synthetic light curves for the hot-path workloads.

Independent implementation of the *recipe* behind the reference generators
``make_simple_sinusoid_1d`` / ``make_multi_sinusoid_1d`` /
``make_chromatic_sinusoid_2d`` (``/root/reference/pgmuvi/synthetic.py:308, 385,
503``): a seeded ``numpy.random.default_rng`` draws the observation times first
and the noise second, values are rounded to float32 exactly where the reference
builds its tensors (``synthetic.py:379-381, 674-680``).  The reference returns a
``Lightcurve``; here the raw ``(t, y, yerr)`` tensors are returned because only
they feed the hot path.  ``tests/golden/*.npz`` (made by importing the
reference's helpers, see ``tests/golden/make_golden.py``) pin this bit-for-bit.
"""
from __future__ import annotations

import math
from typing import Sequence, Optional, Tuple

import numpy as np
import torch

TSPAN_FACTOR = 2.3  # reference default span = 2.3 longest periods (synthetic.py:77)

CFG2_COMPONENTS = (
    dict(period=150.0, amplitude=1.0, phase=0.0),
    dict(period=67.0, amplitude=0.5, phase=math.pi / 3),
    dict(period=400.0, amplitude=0.3, phase=2 * math.pi / 3),
    dict(period=31.0, amplitude=0.2, phase=1.0),
)


def _times(rng, n, t_min, t_span, irregular):
    if irregular:
        return np.sort(rng.uniform(t_min, t_min + t_span, n))
    return np.linspace(t_min, t_min + t_span, n)


def _noisy(rng, signal, noise_level, noise_type):
    """Returns (y, yerr).  'poisson' = shot-noise-like: sigma grows as the square
    root of the (shifted, positive) flux and equals noise_level at mean flux."""
    if noise_type is None or noise_level <= 0:
        return signal.copy(), None
    n = signal.shape[0]
    if noise_type == "gaussian":
        return signal + rng.standard_normal(n) * noise_level, np.full(n, noise_level)
    if noise_type != "poisson":
        raise ValueError(f"Unknown noise_type '{noise_type}'")
    floor = float(np.abs(signal).max()) * 0.01 + 1e-10
    positive = signal - float(signal.min()) + floor
    sigma = noise_level * np.sqrt(positive / float(positive.mean()))
    return signal + rng.standard_normal(n) * sigma, sigma


def _to32(a):
    return None if a is None else torch.as_tensor(a, dtype=torch.float32)


def multi_sinusoid_1d(
    n_obs: int,
    components: Sequence[dict] = CFG2_COMPONENTS,
    noise_level: float = 0.1,
    noise_type: Optional[str] = "poisson",
    t_min: float = 0.0,
    t_span: Optional[float] = None,
    irregular: bool = True,
    seed: Optional[int] = None,
) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor]]:
    rng = np.random.default_rng(seed)
    if t_span is None:
        t_span = TSPAN_FACTOR * max(c["period"] for c in components)
    t = _times(rng, n_obs, t_min, t_span, irregular)
    y = np.zeros(n_obs)
    for c in components:
        y = y + c["amplitude"] * np.sin(2 * math.pi * t / c["period"] + c["phase"])
    y, yerr = _noisy(rng, y, noise_level, noise_type)
    return _to32(t), _to32(y), _to32(yerr)


def simple_sinusoid_1d(n_obs, period=150.0, amplitude=1.0, phase=0.0, noise_level=0.1,
                       noise_type="poisson", t_min=0.0, t_span=None, irregular=True, seed=None):
    rng = np.random.default_rng(seed)
    if t_span is None:
        t_span = TSPAN_FACTOR * period
    t = _times(rng, n_obs, t_min, t_span, irregular)
    y = amplitude * np.sin(2 * math.pi * t / period + phase)
    y, yerr = _noisy(rng, y, noise_level, noise_type)
    return _to32(t), _to32(y), _to32(yerr)


def chromatic_sinusoid_2d(n_per_band, period, wavelengths, amplitude=1.0, amplitude_slope=0.0,
                          wl_ref=0.0, phase=0.0, noise_level=0.1, noise_type="poisson",
                          t_min=0.0, t_span=None, irregular=True, seed=None):
    """Linear amplitude law, no phase law (the cfg-4 shape)."""
    if t_span is None:
        t_span = TSPAN_FACTOR * period
    rng = np.random.default_rng(seed)
    wl = np.asarray(wavelengths, dtype=float)
    amps = amplitude * (1.0 + amplitude_slope * (wl - wl_ref))
    ts, ws, ys, es = [], [], [], []
    for lam, amp in zip(wl, amps):
        tb = _times(rng, n_per_band, t_min, t_span, irregular)
        yb, eb = _noisy(rng, amp * np.sin(2 * math.pi * tb / period + phase), noise_level, noise_type)
        ts.append(tb); ws.append(np.full(n_per_band, lam)); ys.append(yb)
        if eb is not None:
            es.append(eb)
    x = torch.tensor(np.column_stack([np.concatenate(ts), np.concatenate(ws)]), dtype=torch.float32)
    y = _to32(np.concatenate(ys))
    yerr = _to32(np.concatenate(es)) if es else None
    return x, y, yerr


# ---- the five BASELINE.json configurations ----------
def cfg1(n_obs=256, seed=1):
    return simple_sinusoid_1d(n_obs, period=150.0, amplitude=1.0, noise_level=0.1, seed=seed)


def cfg2(n_obs=4096, seed=2, t_span=3450.0):
    return multi_sinusoid_1d(n_obs, CFG2_COMPONENTS, noise_level=0.1, t_span=t_span, seed=seed)


def cfg3_lightcurve(i: int, n_obs=2048, base_seed=1000):
    """i-th light curve of the cfg-3 batch: leading period ~ U(30,300) from default_rng(3)."""
    periods = np.random.default_rng(3).uniform(30.0, 300.0, size=max(i + 1, 512))
    comps = [dict(c) for c in CFG2_COMPONENTS]
    comps[0]["period"] = float(periods[i])
    return multi_sinusoid_1d(n_obs, comps, noise_level=0.1, t_span=3450.0, seed=base_seed + i), float(periods[i])


def cfg4(n_per_band=1024, seed=42):
    return chromatic_sinusoid_2d(n_per_band, period=12.5, wavelengths=np.linspace(0.45, 2.2, 8),
                                 amplitude=1.0, amplitude_slope=2.5, wl_ref=0.45,
                                 noise_level=0.15, t_span=100.0, seed=seed)


def cfg_hypers(cfg: int, y: torch.Tensor, dtype=torch.float64, lead_period: float = 150.0):
    """Hyper-parameters at which section 8d evaluates the MLL."""
    if cfg == 1:
        w = torch.tensor([1.0], dtype=dtype)
        mu = torch.tensor([[[1.0 / 150.0]]], dtype=dtype)
        v = torch.tensor([[[1.0 / 1500.0]]], dtype=dtype)
    elif cfg in (2, 3, 5):
        amp = torch.tensor([1.0, 0.5, 0.3, 0.2], dtype=dtype)
        w = amp ** 2 / 2.0
        mu = (1.0 / torch.tensor([lead_period, 67.0, 400.0, 31.0], dtype=dtype)).reshape(4, 1, 1)
        v = mu / 10.0
    elif cfg == 4:
        w = torch.full((3,), 1.0 / 3.0, dtype=dtype)
        mu = torch.tensor([[1 / 12.5, 0.5], [2 / 12.5, 0.5], [1 / 25.0, 0.5]], dtype=dtype).reshape(3, 1, 2)
        v = torch.tensor([[0.01, 0.3]] * 3, dtype=dtype).reshape(3, 1, 2)
    else:
        raise ValueError(cfg)
    return dict(w=w, mu=mu, v=v, mean=y.to(dtype).mean())


def perturbed_hypers(h: dict, k: int, seed: int = 20, scale: float = 0.1):
    """k-th random perturbation theta * exp(scale * N(0,1)) (section 8d, cfg 2)."""
    rng = np.random.default_rng(seed)
    out = None
    for _ in range(k + 1):
        out = {}
        for name in ("w", "mu", "v"):
            z = torch.as_tensor(rng.standard_normal(tuple(h[name].shape)), dtype=h[name].dtype)
            out[name] = h[name] * torch.exp(scale * z)
        out["mean"] = h["mean"]
    return out
