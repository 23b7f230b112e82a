"""CPU oracle for the Lomb-Scargle seeding step (TEST INFRASTRUCTURE, NOT PRODUCT CODE).

The reference seeds the spectral-mixture frequencies of ``Lightcurve.fit()`` from a Lomb-Scargle
periodogram (``/root/reference/pgmuvi/lightcurve.py:4214-4611`` ``fit_LS``; used by ``fit`` at
``:5516-5541``).  The arithmetic lives in the third-party package **astropy**
(``astropy.timeseries.LombScargle``; unpinned in ``/root/reference/pyproject.toml``, not installed here,
no network), called as

    LS = LombScargle(t, y, yerr)                       # fit_mean=True, center_data=True, 'standard'
    freq = LS.autofrequency(nyquist_factor=5)          # samples_per_peak=5
    power = LS.power(freq, assume_regular_frequency=True)
    LS.false_alarm_probability(power.max(), method='davies');  LS.false_alarm_probability(p, method='single')

This file restates the *published* algorithm: the floating-mean ("generalised") periodogram of
Zechmeister & Kuerster (2009) in the tau-free form, the frequency heuristic of VanderPlas (2018, sec. 7.1)
and the false-alarm formulas of Baluev (2008) as astropy documents them.  astropy's default
``method='auto'`` evaluates the same periodogram with an FFT-based approximation (Press & Rybicki) whose
result differs from the exact sums by ~1e-3 relative; the exact sums are the reference point here.

``power_fast`` restates that approximation (extirpolation onto a regular grid + one inverse FFT per trigonometric
sum) because the reference's recorded outputs come from it: on pgmuvi's grids (> 200 regular frequencies) astropy's
``method='auto'`` resolves to it, and it differs from the exact sums by up to 1e-2 in the power at the high-frequency
end -- enough to swap the order of two near-equal peaks.

**Pinned by the recorded outputs the reference holds** (``tests/test_dropin_reference.py``, all through the
reference's own ``fit_LS`` on the astropy-shaped shim): the comparison notebook's initial frequencies (0.0067,
0.0154); the Lomb-Scargle notebook's five 1-D peaks to the 6 printed digits with their significance flags
(``docs/source/notebooks/PGMUVI_Lomb_Scargle.ipynb`` cells 10/12/34: exact sums give the same five peaks with the
4th and 5th swapped -- their powers are 0.3913 and 0.3910 --, ``power_fast`` gives the recorded order; the flags come
from the reference's own phase-scramble bootstrap over the periodogram, because a band selected from a 2-D light curve
stays on the multiband code path -- the analytic 'davies' / 'baluev' / 'single' formulas below remain unverified); the
``use_best_band_init=True`` periodogram's peak period / height / prominence (149.170715 / 0.992789 / 0.859738, cell 20).
The *default multiband* numbers of that cell (height 0.909449, prominence 0.579050, area fraction 0.016485) and the eight
peaks of the two-period cell are reproduced too (round 6) -- by weighting the per-band periodograms with the sum of their
own squared powers, which is what astropy's 'fast' multiband method does (``multiband_fast``); the published chi^2 weights
give 0.984977 / 0.824881.  Independent known-answer check available here: the
periodogram is 1 - chi^2(f)/chi^2_ref of an explicit weighted least-squares fit (``numpy.linalg.lstsq``).

Only ``tests/`` may import this module.
"""
from __future__ import annotations

import math

import numpy as np
from scipy.special import gammaln


def autofrequency(t, samples_per_peak=5, nyquist_factor=5, minimum_frequency=None, maximum_frequency=None):
    """Regular grid f0 + df * arange(Nf): df = 1 / (samples_per_peak * baseline), f0 = df / 2,
    f_max = nyquist_factor * (N / 2 / baseline)."""
    t = np.asarray(t, dtype=float)
    baseline = t.max() - t.min()
    n = t.size
    df = 1.0 / baseline / samples_per_peak
    f0 = 0.5 * df if minimum_frequency is None else minimum_frequency
    if maximum_frequency is None:
        maximum_frequency = nyquist_factor * (0.5 * n / baseline)
    nf = 1 + int(np.round((maximum_frequency - f0) / df))
    return f0 + df * np.arange(nf)


def power(t, y, dy, freq, fit_mean=True, center_data=True):
    """Standard-normalised generalised Lomb-Scargle power (exact sums, fp64)."""
    t = np.asarray(t, dtype=float); y = np.asarray(y, dtype=float)
    w = np.ones_like(t) if dy is None else np.asarray(dy, dtype=float) ** -2.0
    w = w / w.sum()
    if center_data or fit_mean:
        y = y - np.dot(w, y)
    freq = np.asarray(freq, dtype=float)
    out = np.empty(freq.shape)
    yy = np.dot(w, y * y)
    for lo in range(0, freq.size, 2048):
        f = freq[lo:lo + 2048]
        arg = 2.0 * math.pi * f[:, None] * t[None, :]
        c, s = np.cos(arg), np.sin(arg)
        yc, ys = c @ (w * y), s @ (w * y)
        cc, cs = (c * c) @ w, (c * s) @ w
        ss = 1.0 - cc
        if fit_mean:
            C, S = c @ w, s @ w
            cc, ss, cs = cc - C * C, ss - S * S, cs - C * S      # (Y = 0 after centring)
        d = cc * ss - cs * cs
        out[lo:lo + 2048] = (ss * yc * yc + cc * ys * ys - 2.0 * cs * yc * ys) / (yy * d)
    return out


def _spread(x, h, n, m=4):
    """Extirpolation (Press & Rybicki 1989): each sample h_k at the fractional position x_k is spread onto the ``m``
    nearest grid points with Lagrange weights, so that sum_j grid_j g(j) == sum_k h_k g(x_k) for polynomials g of
    degree < m."""
    grid = np.zeros(n, dtype=h.dtype)
    on = x % 1 == 0
    np.add.at(grid, x[on].astype(int), h[on])
    x, h = x[~on], h[~on]
    lo = np.clip((x - m // 2).astype(int), 0, n - m)
    for j in range(m):
        wgt = np.ones_like(x)
        for i in range(m):
            if i != j:
                wgt = wgt * (x - lo - i) / (j - i)
        np.add.at(grid, lo + j, h * wgt)
    return grid


def _trig_sums_fft(t, h, f0, df, nf, factor=1, oversampling=5, m=4):
    """S_k = sum_i h_i sin(2 pi f_k t_i), C_k likewise, f_k = factor (f0 + k df), by extirpolation of h onto a grid of
    ``2**ceil(log2(oversampling nf))`` points and one inverse FFT (astropy's defaults: oversampling 5, 4-point spread)."""
    f0, df = f0 * factor, df * factor
    nfft = 1 << int(nf * oversampling - 1).bit_length()
    t0 = t.min()
    h = h.astype(complex)
    if f0 > 0:
        h = h * np.exp(2j * np.pi * f0 * (t - t0))
    grid = _spread(((t - t0) * nfft * df) % nfft, h, nfft, m)
    z = np.fft.ifft(grid)[:nf] * nfft
    if t0 != 0:
        z = z * np.exp(2j * np.pi * t0 * (f0 + df * np.arange(nf)))
    return z.imag, z.real


def power_fast(t, y, dy, f0, df, nf, fit_mean=True, center_data=True):
    """The FFT approximation astropy's ``method='auto'`` takes on a regular grid of more than 200 frequencies
    (``LombScargle.power(freq, assume_regular_frequency=True)``, ``/root/reference/pgmuvi/lightcurve.py:4514``): the same
    floating-mean periodogram in its tau form, with the three pairs of trigonometric sums taken from extirpolated FFTs."""
    t = np.asarray(t, dtype=float); y = np.asarray(y, dtype=float)
    w = np.ones_like(t) if dy is None else np.asarray(dy, dtype=float) ** -2.0
    w = w / w.sum()
    if center_data or fit_mean:
        y = y - np.dot(w, y)
    sh, ch = _trig_sums_fft(t, w * y, f0, df, nf)
    s2, c2 = _trig_sums_fft(t, w, f0, df, nf, factor=2)
    if fit_mean:
        s, c = _trig_sums_fft(t, w, f0, df, nf)
        tan2 = (s2 - 2.0 * s * c) / (c2 - (c * c - s * s))
    else:
        tan2 = s2 / c2
    c2w = 1.0 / np.sqrt(1.0 + tan2 * tan2)
    s2w = tan2 * c2w
    cw = np.sqrt(0.5) * np.sqrt(1.0 + c2w)
    sw = np.sqrt(0.5) * np.sign(s2w) * np.sqrt(1.0 - c2w)
    yc, ys = ch * cw + sh * sw, sh * cw - ch * sw
    cc = 0.5 * (1.0 + c2 * c2w + s2 * s2w)
    ss = 0.5 * (1.0 - c2 * c2w - s2 * s2w)
    if fit_mean:
        cc = cc - (c * cw + s * sw) ** 2
        ss = ss - (s * cw - c * sw) ** 2
    return (yc * yc / cc + ys * ys / ss) / np.dot(w, y * y)


def power_auto(t, y, dy, freq, fit_mean=True, center_data=True):
    """astropy's ``method='auto'`` rule for the single-term periodogram: the FFT approximation when the grid is regular
    and longer than 200 frequencies, the exact sums otherwise."""
    freq = np.asarray(freq, dtype=float)
    if freq.ndim == 1 and freq.size > 200:
        d = np.diff(freq)
        if np.allclose(d, d[0]) and d[0] > 0:
            return power_fast(t, y, dy, freq[0], d[0], freq.size, fit_mean, center_data)
    return power(t, y, dy, freq, fit_mean, center_data)


def multiband_fast(t, y, bands, dy, freq, fit_mean=True, center_data=True, sb_auto=False):
    """Multiband periodogram, "fast" form, AS ASTROPY COMPUTES IT (what ``pgmuvi/multiband_ls_significance.py:51-106, 202`` asks
    for: ``LombScargleMultiband(t, y, bands, dy=dy).power(freq, method='fast')``): one standard-normalised floating-mean
    periodogram per band, combined with the weights

        weight[b] = sum_f P_b(f)^2 / sum_b' sum_f P_b'(f)^2

    -- the sum of the band's own SQUARED POWERS over the frequency grid.  The published method (VanderPlas & Ivezic 2015;
    gatspy's ``LombScargleMultibandFast``) weights by each band's reference chi^2; astropy's port keeps that line's shape but
    applies it to what its per-band call returns, which is the power array (``mbfast_impl.lombscargle_mbfast``: "Total score
    is the sum of powers weighted by chi2-normalization").  Settled by the reference's recorded outputs, not by reading astropy
    (not installed here): with these weights the Lomb-Scargle notebook's default multiband cell reproduces to the printed
    digit -- height 0.909449 (0.9094498 here), prominence 0.579050 (0.579052), area fraction 0.016485 (0.016482), all eight
    peak frequencies of the two-period cell in the recorded order -- while the chi^2 weights give 0.984977 / 0.824881
    (``docs/source/notebooks/PGMUVI_Lomb_Scargle.ipynb:893-903``; ``multiband_chi2_weighted`` below keeps that form)."""
    t = np.asarray(t, dtype=float); y = np.asarray(y, dtype=float); bands = np.asarray(bands)
    powers = []
    for b in np.unique(bands):
        m = bands == b
        dyb = None if dy is None else np.asarray(dy, dtype=float)[m]
        powers.append((power_auto if sb_auto else power)(t[m], y[m], dyb, freq, fit_mean, center_data))
    powers = np.asarray(powers)
    wgt = np.sum(powers ** 2, axis=1)
    return np.dot(wgt / wgt.sum(), powers)


def multiband_chi2_weighted(t, y, bands, dy, freq, fit_mean=True, center_data=True, sb_auto=False):
    """The published form (per-band powers weighted by each band's reference chi^2 about its weighted mean) -- NOT what the
    reference's recorded numbers contain (see ``multiband_fast``); kept so that the test of the recorded cell can show both."""
    t = np.asarray(t, dtype=float); y = np.asarray(y, dtype=float); bands = np.asarray(bands)
    chi2_0, powers = [], []
    for b in np.unique(bands):
        m = bands == b
        dyb = None if dy is None else np.asarray(dy, dtype=float)[m]
        w = np.ones(int(m.sum())) if dyb is None else dyb ** -2.0
        chi2_0.append(np.sum(w * (y[m] - np.dot(w, y[m]) / w.sum()) ** 2))
        powers.append((power_auto if sb_auto else power)(t[m], y[m], dyb, freq, fit_mean, center_data))
    chi2_0 = np.asarray(chi2_0)
    return np.dot(chi2_0 / chi2_0.sum(), np.asarray(powers))


def power_by_least_squares(t, y, dy, f):
    """Independent definition: 1 - chi2(f) / chi2_ref with chi2(f) from an explicit weighted fit of
    (1, cos, sin) and chi2_ref from the weighted mean alone."""
    t = np.asarray(t, dtype=float); y = np.asarray(y, dtype=float)
    sw = np.ones_like(t) if dy is None else 1.0 / np.asarray(dy, dtype=float)
    X0 = np.ones((t.size, 1))
    X1 = np.stack([np.ones_like(t), np.cos(2 * math.pi * f * t), np.sin(2 * math.pi * f * t)], axis=1)
    chi = []
    for X in (X0, X1):
        beta, *_ = np.linalg.lstsq(X * sw[:, None], y * sw, rcond=None)
        r = (y - X @ beta) * sw
        chi.append(float(r @ r))
    return 1.0 - chi[1] / chi[0]


# ---- false-alarm probabilities, 'standard' normalisation (Baluev 2008; unverified against astropy itself)
def _gamma(n):
    return math.sqrt(2.0 / n) * math.exp(gammaln(n / 2.0) - gammaln((n - 1) / 2.0))


def fap_single(z, n, dK=3):
    return (1.0 - np.asarray(z, dtype=float)) ** (0.5 * (n - dK))


def tau_davies(z, fmax, t, dy, dH=1, dK=3):
    t = np.asarray(t, dtype=float)
    n = t.size
    w = np.ones_like(t) if dy is None else np.asarray(dy, dtype=float) ** -2.0
    w = w / w.sum()
    dt = np.dot(w, t * t) - np.dot(w, t) ** 2
    teff = math.sqrt(4.0 * math.pi * dt)
    W = fmax * teff
    z = np.asarray(z, dtype=float)
    nh, nk = n - dH, n - dK
    return _gamma(nh) * W * (1.0 - z) ** (0.5 * (nk - 1)) * np.sqrt(0.5 * nh * z)


def false_alarm_probability(z, fmax, t, dy, method="baluev"):
    n = np.asarray(t).size
    fs = fap_single(z, n)
    if method == "single":
        return fs
    tau = tau_davies(z, fmax, t, dy)
    if method == "davies":
        return fs + tau
    if method == "baluev":
        return 1.0 - (1.0 - fs) * np.exp(-tau)
    if method == "naive":
        T = np.max(t) - np.min(t)
        return 1.0 - (1.0 - fs) ** (fmax * T)
    raise ValueError(method)
