"""Lomb-Scargle seeding of the spectral-mixture frequencies on the MI355X.

``Lightcurve.fit()`` seeds ``mixture_means`` from the peaks of a Lomb-Scargle periodogram
(``/root/reference/pgmuvi/lightcurve.py:5516-5541`` -> ``fit_LS``, ``:4214-4611``), which the reference
computes with ``astropy.timeseries.LombScargle``::

    LS = LombScargle(t, y, yerr); freq = LS.autofrequency(nyquist_factor=5)
    power = LS.power(freq, assume_regular_frequency=True)
    LS.false_alarm_probability(power.max(), method='davies'); LS.false_alarm_probability(power[peaks], method='single')

This module provides that class surface on the HIP kernel ``pgm_lomb_scargle_f64`` (exact floating-mean
periodogram sums, one thread per frequency, batched over light curves) and, for many-light-curve work,
``periodogram_batched`` / ``seed_frequencies``.  ``install_as_astropy()`` registers a minimal
``astropy.timeseries`` so that the reference's own ``fit_LS`` runs unmodified where astropy is absent.
There is no CPU path: the periodogram runs on the GPU or raises.

The periodogram is the published generalised Lomb-Scargle (Zechmeister & Kuerster 2009).  ``power(method=...)`` follows
astropy's rule: ``'auto'`` (the default, what pgmuvi calls) takes the FFT approximation of Press & Rybicki on a regular
grid of more than 200 frequencies -- ``pgm_lomb_scargle_fast_f64``: samples spread onto an oversampled grid, inverse FFTs,
the tau form -- and the exact sums otherwise; ``'fast'`` / ``'slow'`` force one or the other.  The approximation is what
the reference's recorded outputs contain: it differs from the exact sums by up to 1e-2 in the power at the high-frequency
end and reorders near-equal peaks.  The false-alarm formulas (Baluev 2008) are restated from
astropy's published implementation -- unverified against an installed astropy.
"""
from __future__ import annotations

import math
import sys
import types
from typing import Optional

import numpy as np
import torch

from . import _hip


def _np(a):
    if a is None:
        return None
    if torch.is_tensor(a):
        return a.detach().cpu().numpy().astype(np.float64)
    return np.asarray(a, dtype=np.float64)


def _compute_device() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("pgmuvi_amd.lombscargle evaluates the periodogram with its HIP kernel on an MI355X; "
                           "there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _regular_grid(f: np.ndarray):
    """(f0, df) when ``f`` is a regular increasing grid (astropy's ``_is_regular``), else None."""
    if f.ndim != 1 or f.size < 2:
        return None
    d = np.diff(f)
    if not (d[0] > 0 and np.allclose(d, d[0])):
        return None
    return float(f[0]), float(d[0])


def periodogram_batched(t: torch.Tensor, y: torch.Tensor, dy: Optional[torch.Tensor], freq: torch.Tensor,
                        fit_mean=True, center_data=True, method="slow", assume_regular_frequency=False) -> torch.Tensor:
    """Standard-normalised floating-mean Lomb-Scargle power, (B, Nf), for B light curves (B, N) on the GPU and
    one frequency grid (Nf,) shared by the batch.  ``method``: 'slow' exact sums (default here), 'fast' the FFT
    approximation (regular grid), 'auto' astropy's choice between the two."""
    if method not in ("auto", "fast", "slow", "cython", "scipy"):
        raise NotImplementedError(f"Lomb-Scargle method {method!r} is not implemented (auto, fast, slow)")
    if method in ("auto", "fast"):
        f = freq.detach().cpu().numpy().astype(np.float64).reshape(-1)
        if method == "fast" or f.size > 200:
            reg = (float(f[0]), float(f[1] - f[0])) if (assume_regular_frequency and f.size > 1) else _regular_grid(f)
            if reg is not None and reg[0] >= 0.0:
                return _hip.lomb_scargle_fast(t, y, dy, reg[0], reg[1], f.size, fit_mean, center_data)
            if method == "fast":
                raise ValueError("method='fast' needs a regular frequency grid with f0 >= 0")
    return _hip.lomb_scargle(t, y, dy, freq, fit_mean, center_data)


class LombScargle:
    """``astropy.timeseries.LombScargle`` as pgmuvi uses it (single term, 'standard' normalisation)."""

    def __init__(self, t, y, dy=None, fit_mean=True, center_data=True, nterms=1, normalization="standard"):
        if nterms != 1:
            raise NotImplementedError("only the single-term periodogram pgmuvi uses is implemented")
        if normalization != "standard":
            raise NotImplementedError("only normalization='standard' (astropy's default, what pgmuvi uses) is implemented")
        self.t, self.y, self.dy = _np(t).reshape(-1), _np(y).reshape(-1), None if dy is None else _np(dy).reshape(-1)
        if self.t.shape != self.y.shape or (self.dy is not None and self.dy.shape != self.t.shape):
            raise ValueError("t, y, dy must have the same shape")
        self.fit_mean, self.center_data, self.normalization = fit_mean, center_data, normalization

    # VanderPlas (2018) sec. 7.1 heuristic, astropy's defaults
    def autofrequency(self, samples_per_peak=5, nyquist_factor=5, minimum_frequency=None, maximum_frequency=None,
                      return_freq_limits=False):
        baseline = self.t.max() - self.t.min()
        n = self.t.size
        df = 1.0 / baseline / samples_per_peak
        f0 = 0.5 * df if minimum_frequency is None else float(minimum_frequency)
        if maximum_frequency is None:
            maximum_frequency = nyquist_factor * (0.5 * n / baseline)
        nf = 1 + int(np.round((maximum_frequency - f0) / df))
        if return_freq_limits:
            return f0, f0 + df * (nf - 1)
        return f0 + df * np.arange(nf)

    def power(self, frequency, normalization=None, method="auto", assume_regular_frequency=False, method_kwds=None,
              fit_mean=None, center_data=None):
        if normalization not in (None, "standard"):
            raise NotImplementedError("only normalization='standard'")
        dev = _compute_device()
        D = torch.float64
        f = np.asarray(_np(frequency), dtype=np.float64)
        shape = f.shape
        tt = torch.as_tensor(self.t, dtype=D, device=dev).reshape(1, -1)
        yy = torch.as_tensor(self.y, dtype=D, device=dev).reshape(1, -1)
        dd = None if self.dy is None else torch.as_tensor(self.dy, dtype=D, device=dev).reshape(1, -1)
        ff = torch.as_tensor(f.reshape(-1), dtype=D, device=dev)
        p = periodogram_batched(tt, yy, dd, ff, self.fit_mean if fit_mean is None else fit_mean,
                                self.center_data if center_data is None else center_data, method, assume_regular_frequency)
        return p[0].cpu().numpy().reshape(shape)

    # ---- false-alarm probabilities ('standard' normalisation; Baluev 2008 as implemented by astropy)
    def _fmax(self, samples_per_peak, nyquist_factor, minimum_frequency, maximum_frequency):
        if maximum_frequency is not None:
            return float(maximum_frequency)
        return self.autofrequency(samples_per_peak, nyquist_factor, minimum_frequency, None, return_freq_limits=True)[1]

    def false_alarm_probability(self, power, method="baluev", samples_per_peak=5, nyquist_factor=5,
                                minimum_frequency=None, maximum_frequency=None, method_kwds=None):
        z = np.asarray(_np(power), dtype=np.float64)
        n = self.t.size
        fs = (1.0 - z) ** (0.5 * (n - 3))
        if method == "single":
            return fs
        fmax = self._fmax(samples_per_peak, nyquist_factor, minimum_frequency, maximum_frequency)
        w = np.ones_like(self.t) if self.dy is None else self.dy ** -2.0
        w = w / w.sum()
        dt = np.dot(w, self.t ** 2) - np.dot(w, self.t) ** 2
        W = fmax * math.sqrt(4.0 * math.pi * dt)
        nh, nk = n - 1, n - 3
        gam = math.sqrt(2.0 / nh) * math.exp(math.lgamma(nh / 2.0) - math.lgamma((nh - 1) / 2.0))
        tau = gam * W * (1.0 - z) ** (0.5 * (nk - 1)) * np.sqrt(0.5 * nh * z)
        if method == "davies":
            return fs + tau
        if method == "baluev":
            return 1.0 - (1.0 - fs) * np.exp(-tau)
        if method == "naive":
            return 1.0 - (1.0 - fs) ** (fmax * (self.t.max() - self.t.min()))
        raise NotImplementedError(f"false_alarm_probability method {method!r} (bootstrap needs resampling runs; use "
                                  "'davies', 'baluev', 'naive' or 'single')")


class LombScargleMultiband:
    """``astropy.timeseries.LombScargleMultiband`` as pgmuvi uses it (``pgmuvi/multiband_ls_significance.py:51-106``, always with
    ``method='fast'``): one floating-mean Lomb-Scargle periodogram per band (the HIP kernel, one call per band), combined
    with the weights astropy's 'fast' multiband method uses -- ``sum_f P_b(f)**2`` of each band's own periodogram over the
    frequency grid, normalised over the bands (NOT the reference chi^2 of the published method, VanderPlas & Ivezic 2015).
    Pinned by the reference's recorded outputs: the Lomb-Scargle notebook's default multiband cell (height 0.909449,
    prominence 0.579050) and the eight peaks of its two-period cell reproduce to the printed digits through the reference's
    own ``fit_LS`` (``tests/test_dropin_reference.py``; ``docs/source/notebooks/PGMUVI_Lomb_Scargle.ipynb:893-903``).  The
    'flexible' method (a regularised multi-term model fit) is not implemented."""

    def __init__(self, t, y, bands, dy=None, normalization="standard", nterms_base=1, nterms_band=1, reg_base=None,
                 reg_band=1e-6, regularize_by_trace=True, center_data=True, fit_mean=True):
        if normalization != "standard":
            raise NotImplementedError("only normalization='standard' (astropy's default, what pgmuvi uses) is implemented")
        self.t, self.y = _np(t).reshape(-1).astype(np.float64), _np(y).reshape(-1).astype(np.float64)
        self.bands = np.asarray(_np(bands)).reshape(-1)
        self.dy = None if dy is None else _np(dy).reshape(-1).astype(np.float64)
        if not (self.t.shape == self.y.shape == self.bands.shape) or (self.dy is not None and self.dy.shape != self.t.shape):
            raise ValueError("t, y, bands, dy must have the same shape")
        self.normalization, self.center_data, self.fit_mean = normalization, center_data, fit_mean
        self.nterms_base, self.nterms_band = nterms_base, nterms_band

    def autofrequency(self, samples_per_peak=5, nyquist_factor=5, minimum_frequency=None, maximum_frequency=None,
                      return_freq_limits=False):
        return LombScargle(self.t, self.y).autofrequency(samples_per_peak, nyquist_factor, minimum_frequency, maximum_frequency,
                                                         return_freq_limits)

    def power(self, frequency, method="flexible", sb_method="auto", normalization="standard"):
        if method != "fast":
            raise NotImplementedError("LombScargleMultiband: only method='fast' (what pgmuvi uses) is implemented")
        if normalization not in (None, "standard"):
            raise NotImplementedError("only normalization='standard'")
        f = np.asarray(_np(frequency), dtype=np.float64)
        powers = []
        for band in np.unique(self.bands):
            m = self.bands == band
            dyb = None if self.dy is None else self.dy[m]
            powers.append(LombScargle(self.t[m], self.y[m], dyb, fit_mean=self.fit_mean, center_data=self.center_data).power(f.reshape(-1), method=sb_method))
        powers = np.asarray(powers)
        wgt = np.sum(powers ** 2, axis=1)                        # (astropy's weights: each band's summed squared power)
        return np.dot(wgt / wgt.sum(), powers).reshape(f.shape)

    def autopower(self, method="flexible", sb_method="auto", normalization="standard", samples_per_peak=5, nyquist_factor=5,
                  minimum_frequency=None, maximum_frequency=None):
        f = self.autofrequency(samples_per_peak, nyquist_factor, minimum_frequency, maximum_frequency)
        return f, self.power(f, method=method, sb_method=sb_method, normalization=normalization)


def seed_frequencies(t: torch.Tensor, y: torch.Tensor, dy: Optional[torch.Tensor], num_peaks=1, nyquist_factor=5,
                     samples_per_peak=5):
    """Batched seeding for many light curves (B, N) that share one sampling span: the ``num_peaks`` highest
    periodogram peaks of each (peaks at least ``nyquist_factor`` grid points apart, like
    ``find_peaks(power, distance=Nyquist_factor)`` in ``fit_LS``).  Returns (freqs (B, num_peaks), powers, grid)."""
    from scipy.signal import find_peaks
    B, n = y.shape
    tmin, tmax = float(t.min()), float(t.max())
    baseline = tmax - tmin
    df = 1.0 / baseline / samples_per_peak
    f0 = 0.5 * df
    nf = 1 + int(np.round((nyquist_factor * 0.5 * n / baseline - f0) / df))
    grid = f0 + df * torch.arange(nf, dtype=torch.float64, device=y.device)
    P = periodogram_batched(t, y, dy, grid).cpu().numpy()
    g = grid.cpu().numpy()
    freqs = np.full((B, num_peaks), np.nan)
    pows = np.full((B, num_peaks), np.nan)
    for b in range(B):
        pk, _ = find_peaks(P[b], distance=nyquist_factor)
        pk = pk[np.argsort(P[b][pk])][::-1][:num_peaks]
        freqs[b, :len(pk)] = g[pk]
        pows[b, :len(pk)] = P[b][pk]
    return freqs, pows, g


def install_as_astropy(force=False):
    """Registers a minimal ``astropy.timeseries`` (LombScargle on the HIP kernel, LombScargleMultiband as an
    importable placeholder) so that ``pgmuvi``'s ``fit_LS`` / default ``fit()`` seeding run where astropy is not
    installed.  Does nothing if a real astropy is importable, unless ``force``."""
    if not force:
        try:
            import astropy.timeseries  # noqa: F401
            return False
        except Exception:
            pass
    pkg = types.ModuleType("astropy")
    pkg.__path__ = []
    ts = types.ModuleType("astropy.timeseries")
    ts.LombScargle = LombScargle
    ts.LombScargleMultiband = LombScargleMultiband
    pkg.timeseries = ts
    sys.modules["astropy"] = pkg
    sys.modules["astropy.timeseries"] = ts
    return True
