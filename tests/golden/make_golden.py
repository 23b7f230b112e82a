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
   GPyTorch outputs (gpytorch cannot be installed here).  The oracle itself is
   pinned by the reference's recorded notebook outputs (DESIGN.md section 5,
   ``make_notebook_pin.py``): formula, division by N, noise, constraints and the
   training loop at N=89 / N=225, to the printed digits -- not round-off-level
   agreement with GPyTorch at these sizes.  These files hold the oracle (and
   through it the HIP path) to fixed numbers at every BASELINE configuration.

``python make_golden.py`` writes the small fixtures (a minute);
``python make_golden.py --fullsize`` adds the BASELINE sizes the dense oracle
needs a quarter of an hour and tens of GB for: config 4 (N=8192, d=2, Q=3, both
dimension orders, value + every gradient, through the row-blocked closed form)
and config 3 (512 light curves x N=2048: every value; every gradient for
members 0, 255 and 511).  Config 3's 12.6 MB of inputs are not stored: the
fixture carries the SHA-256 of the arrays made with the reference's helpers,
which ``pgmuvi_amd.batch.make_shard`` must reproduce.
``python make_golden.py --beyond64``: config 2's recipe at N = 8320 and N = 16384
(65 and 128 block rows: the contract for N > 8192, include/pgmuvi_hip.h).

No reference source text is stored; the .npz files hold numbers only.
"""
import hashlib
import math
import os
import sys
import time

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


def fullsize():
    """Configs 3 and 4 at BASELINE.json's stated sizes ('Next round' 1)."""
    torch.set_num_threads(8)
    # ---- cfg 4: 8 bands x 1024 = 8192 points, d=2, Q=3
    wl = np.linspace(0.45, 2.2, 8)
    x, y, e = ref_chromatic(1024, 12.5, wl, 2.5, 0.45, 0.15, 100.0, seed=42)
    mx, my, me = mine.cfg4()
    assert np.array_equal(x, mx.numpy()) and np.array_equal(y, my.numpy()) and np.array_equal(e, me.numpy())
    np.savez_compressed(os.path.join(HERE, "inputs_cfg4_n8192.npz"), x=x, y=y, yerr=e)
    xd, yd, nd = (torch.as_tensor(a, dtype=torch.float64) for a in (x, y, e))
    nd = nd ** 2
    h = mine.cfg_hypers(4, yd)
    mu, v = h["mu"].reshape(3, 2), h["v"].reshape(3, 2)
    for order in (0, 1):
        t0 = time.time()
        val, g = orc.mll_value_grad_closed_form_blocked(xd, yd, h["mean"], nd, h["w"], mu, v, order, rows=256)
        # independent of the closed form: central differences of the oracle's plain value along one random direction
        rng = np.random.default_rng(44 + order)
        dirs = {k: torch.as_tensor(rng.standard_normal(tuple(t.shape))) * t for k, t in (("w", h["w"]), ("mu", mu), ("v", v))}
        eps = 1e-6
        vp = orc.mll(xd, yd, h["mean"], nd, h["w"] + eps * dirs["w"], mu + eps * dirs["mu"], v + eps * dirs["v"], order)
        vm = orc.mll(xd, yd, h["mean"], nd, h["w"] - eps * dirs["w"], mu - eps * dirs["mu"], v - eps * dirs["v"], order)
        fd = float(vp - vm) / (2 * eps)
        an = float(sum((g[k].reshape(dirs[k].shape) * dirs[k]).sum() for k in dirs))
        v0 = orc.mll(xd, yd, h["mean"], nd, h["w"], mu, v, order)
        assert abs(float(v0 - val)) < 1e-12, float(v0 - val)
        assert abs(fd - an) < 1e-6 * max(1.0, abs(an)), (fd, an)
        out = {"mll_0": val.numpy(), "w_0": h["w"].numpy(), "mu_0": mu.numpy(), "v_0": v.numpy(), "meanc_0": h["mean"].numpy(),
               "fd_check": np.array([fd, an])}
        for name in ("w", "mu", "v", "noise", "mean"):
            out[f"g_{name}_0"] = g[name].numpy()
        np.savez_compressed(os.path.join(HERE, f"expect_cfg4_n8192_order{order}.npz"), **out)
        print(f"cfg4 N=8192 order {order}: mll {float(val):.15g}  directional fd {fd:.9g} vs {an:.9g}  ({time.time() - t0:.0f} s)", flush=True)
    # ---- cfg 3: 512 light curves x N=2048
    B, n = 512, 2048
    periods = ref._rng(3).uniform(30.0, 300.0, size=B)
    sha = hashlib.sha256()
    vals = np.zeros(B)
    grads = {}
    t0 = time.time()
    for i in range(B):
        comps = [dict(c) for c in COMPONENTS]
        comps[0]["period"] = float(periods[i])
        x, y, e = ref_multi(n, comps, 0.1, seed=1000 + i, t_span=3450.0)
        for a in (x, y, e):
            sha.update(np.ascontiguousarray(a).tobytes())
        (mt, my, me), per = mine.cfg3_lightcurve(i, n_obs=n)
        assert per == float(periods[i]) and np.array_equal(x, mt.numpy()) and np.array_equal(y, my.numpy()) and np.array_equal(e, me.numpy())
        xd, yd, nd = (torch.as_tensor(a, dtype=torch.float64) for a in (x, y, e))
        h = mine.cfg_hypers(3, yd, lead_period=per)
        mu, v = h["mu"].reshape(4, 1), h["v"].reshape(4, 1)
        if i in (0, 255, 511):
            val, g = orc.mll_value_grad_closed_form(xd, yd, h["mean"], nd ** 2, h["w"], mu, v)
            for name in ("w", "mu", "v", "noise", "mean"):
                grads[f"g_{name}_{i}"] = g[name].numpy()
        else:
            val = orc.mll(xd, yd, h["mean"], nd ** 2, h["w"], mu, v)
        vals[i] = float(val)
        if i % 32 == 31:
            print(f"cfg3 {i + 1}/{B}  ({time.time() - t0:.0f} s)", flush=True)
    np.savez_compressed(os.path.join(HERE, "expect_cfg3_b512_n2048.npz"), mll=vals, lead_period=periods,
                        inputs_sha256=np.frombuffer(sha.digest(), dtype=np.uint8), **grads)
    print("cfg3 512 x 2048: sum of log-likelihoods", repr(float(vals.sum())), "inputs sha256", sha.hexdigest())


def beyond_64_block_rows():
    """One light curve of more than 64 block rows of 128 (N > 8192: beyond the fused sweep's 64-bit plans, the panel sweep's
    ground): config 2's recipe at N = 8320 (65 block rows) and N = 16384 (128 block rows, the largest size the library's
    contract admits: include/pgmuvi_hip.h, PGM_MAX_N).  Value + every gradient through the row-blocked closed form; the
    inputs are not stored, their SHA-256 (reference helpers' arrays) is."""
    torch.set_num_threads(8)
    for n in (8320, 16384):
        t0 = time.time()
        x, y, e = ref_multi(n, COMPONENTS, 0.1, seed=2, t_span=3450.0)
        mt, my, me = mine.cfg2(n_obs=n)
        assert np.array_equal(x, mt.numpy()) and np.array_equal(y, my.numpy()) and np.array_equal(e, me.numpy())
        sha = hashlib.sha256()
        for a in (x, y, e):
            sha.update(np.ascontiguousarray(a).tobytes())
        xd, yd, nd = (torch.as_tensor(a, dtype=torch.float64) for a in (x, y, e))
        h = mine.cfg_hypers(2, yd)
        mu, v = h["mu"].reshape(4, 1), h["v"].reshape(4, 1)
        val, g = orc.mll_value_grad_closed_form_blocked(xd, yd, h["mean"], nd ** 2, h["w"], mu, v, 0, rows=256)
        if n <= 9000:                                            # (the dense form of N=16384 does not fit this container's memory)
            v0 = orc.mll(xd, yd, h["mean"], nd ** 2, h["w"], mu, v)
            assert abs(float(v0 - val)) < 1e-12
        out = {"mll_0": val.numpy(), "w_0": h["w"].numpy(), "mu_0": mu.numpy(), "v_0": v.numpy(), "meanc_0": h["mean"].numpy(),
               "inputs_sha256": np.frombuffer(sha.digest(), dtype=np.uint8)}
        for name in ("w", "mu", "v", "noise", "mean"):
            out[f"g_{name}_0"] = g[name].numpy()
        np.savez_compressed(os.path.join(HERE, f"expect_cfg2_n{n}.npz"), **out)
        print(f"cfg2 recipe N={n}: mll {float(val):.15g}  ({time.time() - t0:.0f} s)", flush=True)


def main():
    if "--fullsize" in sys.argv:
        return fullsize()
    if "--beyond64" in sys.argv:
        return beyond_64_block_rows()
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
