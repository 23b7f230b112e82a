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

**Parity pinned only by one recorded reference output**: with this oracle behind the astropy-shaped shim,
the reference's own ``fit()`` reproduces the initial frequencies the comparison notebook printed
(``tests/test_dropin_reference.py``); the FAP formulas are restated from memory of astropy's
``_statistics.py`` and labelled unverified.  Independent known-answer check available here: the
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


def multiband_fast(t, y, bands, dy, freq, fit_mean=True, center_data=True):
    """Multiband periodogram, "fast" form (VanderPlas & Ivezic 2015; what ``pgmuvi/multiband_ls_significance.py:51-106`` asks
    astropy for): per-band standard-normalised powers weighted by each band's reference chi^2 about its weighted mean."""
    t = np.asarray(t, dtype=float); y = np.asarray(y, dtype=float); bands = np.asarray(bands)
    chi2_0, powers = [], []
    for b in np.unique(bands):
        m = bands == b
        dyb = None if dy is None else np.asarray(dy, dtype=float)[m]
        w = np.ones(int(m.sum())) if dyb is None else dyb ** -2.0
        chi2_0.append(np.sum(w * (y[m] - np.dot(w, y[m]) / w.sum()) ** 2))
        powers.append(power(t[m], y[m], dyb, freq, fit_mean, center_data))
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
