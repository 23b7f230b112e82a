#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz.  RUN IN THE BUILD CONTAINER ONLY.

Two kinds of data are written:

1. ``inputs_*.npz`` -- light curves produced by *importing the reference's own
   helpers* ``pgmuvi.synthetic._rng / _make_times / _apply_noise /
   _linear_amplitude`` from ``/root/reference`` (they import without gpytorch;
   the public generators do not, they end in ``from pgmuvi.lightcurve import
   Lightcurve``).  These are golden INPUT vectors: ``pgmuvi_amd.synthetic`` must
   reproduce them bit-for-bit from ``(seed, n, ...)``.
2. ``expect_*.npz`` -- outputs of ``oracle/sm_mll_oracle.py`` on those inputs at
   the SURVEY.md section-8d hyper-parameters.  These are oracle outputs, NOT
   GPyTorch outputs (gpytorch cannot be installed here): parity is UNPINNED at
   the gpytorch boundary and these files only pin the oracle against drift.

No reference source text is stored; the .npz files hold numbers only.
"""
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from pgmuvi import synthetic as ref  # noqa: E402  (reference helpers)
from oracle import sm_mll_oracle as orc  # noqa: E402
from pgmuvi_amd import synthetic as mine  # noqa: E402

COMPONENTS = [
    {"period": 150.0, "amplitude": 1.0, "phase": 0.0},
    {"period": 67.0, "amplitude": 0.5, "phase": math.pi / 3},
    {"period": 400.0, "amplitude": 0.3, "phase": 2 * math.pi / 3},
    {"period": 31.0, "amplitude": 0.2, "phase": 1.0},
]


def f32(a):
    return torch.as_tensor(a, dtype=torch.float32).numpy()


def ref_simple(n_obs, period, amplitude, noise_level, seed, t_span=None):
    """Body of make_simple_sinusoid_1d (synthetic.py:372-381) on the reference helpers."""
    if t_span is None:
        t_span = ref._DEFAULT_TSPAN_FACTOR * period
    rng = ref._rng(seed)
    t = ref._make_times(n_obs, 0.0, t_span, True, rng)
    y = amplitude * np.sin(2 * math.pi * t / period + 0.0)
    y, e = ref._apply_noise(y, noise_level, "poisson", rng)
    return f32(t), f32(y), f32(e)


def ref_multi(n_obs, comps, noise_level, seed, t_span):
    """Body of make_multi_sinusoid_1d (synthetic.py:484-500) on the reference helpers."""
    rng = ref._rng(seed)
    t = ref._make_times(n_obs, 0.0, t_span, True, rng)
    y = np.zeros(n_obs)
    for c in comps:
        y = y + c["amplitude"] * np.sin(2 * math.pi * t / c["period"] + c["phase"])
    y, e = ref._apply_noise(y, noise_level, "poisson", rng)
    return f32(t), f32(y), f32(e)


def ref_chromatic(n_per_band, period, wavelengths, slope, wl_ref, noise_level, t_span, seed):
    """Body of make_chromatic_sinusoid_2d (synthetic.py:627-683), linear amplitude law."""
    rng = ref._rng(seed)
    counts = ref._resolve_n_per_band(n_per_band, len(wavelengths), rng)
    wl = np.asarray(wavelengths, dtype=float)
    amps = ref._linear_amplitude(wl, 1.0, slope, wl_ref)
    ts, ws, ys, es = [], [], [], []
    for lam, n, amp in zip(wl, counts, amps):
        tb = ref._make_times(n, 0.0, t_span, True, rng)
        yb, eb = ref._apply_noise(amp * np.sin(2 * math.pi * tb / period + 0.0), noise_level, "poisson", rng)
        ts.append(tb); ws.append(np.full(n, lam)); ys.append(yb); es.append(eb)
    x = torch.tensor(np.column_stack([np.concatenate(ts), np.concatenate(ws)]), dtype=torch.float32).numpy()
    return x, f32(np.concatenate(ys)), f32(np.concatenate(es))


def expect(cfg, x, y, e, dim_order=0, thetas=1):
    xd, yd, nd = (torch.as_tensor(a, dtype=torch.float64) for a in (x, y, e))
    nd = nd ** 2
    h0 = mine.cfg_hypers(cfg, yd)
    out = {}
    for k in range(thetas):
        h = h0 if k == 0 else mine.perturbed_hypers(h0, k - 1)
        Q = h["w"].shape[0]
        mu, v = h["mu"].reshape(Q, -1), h["v"].reshape(Q, -1)
        val, g = orc.mll_value_grad_closed_form(xd, yd, h["mean"], nd, h["w"], mu, v, dim_order)
        val2, g2 = orc.mll_value_grad_autograd(xd, yd, h["mean"], nd, h["w"], mu, v, dim_order)
        assert abs(float(val - val2)) < 1e-12
        for name in ("w", "mu", "v", "noise", "mean"):
            err = (g[name] - g2[name]).abs().max() / (g2[name].abs().max() + 1e-300)
            assert err < 1e-8, (name, float(err))
        out[f"mll_{k}"] = val.numpy()
        for name in ("w", "mu", "v", "noise", "mean"):
            out[f"g_{name}_{k}"] = g[name].numpy()
        out[f"w_{k}"], out[f"mu_{k}"], out[f"v_{k}"] = h["w"].numpy(), mu.numpy(), v.numpy()
        out[f"meanc_{k}"] = h["mean"].numpy()
    return out


def main():
    torch.set_num_threads(8)
    # ---- cfg 1: N=256, Q=1
    x, y, e = ref_simple(256, 150.0, 1.0, 0.1, seed=1)
    np.savez(os.path.join(HERE, "inputs_cfg1.npz"), x=x, y=y, yerr=e)
    np.savez(os.path.join(HERE, "expect_cfg1.npz"), **expect(1, x, y, e))
    # ---- cfg 2 recipe at N = 64, 200 (ragged vs the 128 block), 512 and the full 4096
    for n in (64, 200, 512, 4096):
        x, y, e = ref_multi(n, COMPONENTS, 0.1, seed=2, t_span=3450.0)
        np.savez(os.path.join(HERE, f"inputs_cfg2_n{n}.npz"), x=x, y=y, yerr=e)
        if n <= 512:
            np.savez(os.path.join(HERE, f"expect_cfg2_n{n}.npz"), **expect(2, x, y, e, thetas=3))
        else:
            np.savez(os.path.join(HERE, f"expect_cfg2_n{n}.npz"), **expect(2, x, y, e, thetas=1))
    # ---- cfg 4 recipe (2-D), 8 bands x 32 = 256 points, both dimension orders
    wl = np.linspace(0.45, 2.2, 8)
    x, y, e = ref_chromatic(32, 12.5, wl, 2.5, 0.45, 0.15, 100.0, seed=42)
    np.savez(os.path.join(HERE, "inputs_cfg4_n256.npz"), x=x, y=y, yerr=e)
    np.savez(os.path.join(HERE, "expect_cfg4_n256_order0.npz"), **expect(4, x, y, e, 0))
    np.savez(os.path.join(HERE, "expect_cfg4_n256_order1.npz"), **expect(4, x, y, e, 1))
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
