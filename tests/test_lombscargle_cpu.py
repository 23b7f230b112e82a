"""CPU tests of the Lomb-Scargle seeding layer: the numpy oracle against an independent
least-squares definition, the astropy-shaped class surface (periodogram by the oracle stand-in: no GPU here), the
false-alarm formulas' basic properties."""
import math
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from unittest import mock  # noqa: E402

from oracle import ls_oracle as lso  # noqa: E402
from pgmuvi_amd import _hip, lombscargle as L, synthetic as syn  # noqa: E402
import _oracle_backend as ob  # noqa: E402


def _data(n=200, seed=3):
    t, y, e = syn.cfg2(n_obs=n)
    return t.double().numpy(), y.double().numpy(), e.double().numpy()


def test_oracle_equals_the_least_squares_definition():
    t, y, e = _data()
    for dy in (e, None):
        for f in (1 / 400.0, 1 / 150.0, 1 / 67.0, 0.05, 0.31):
            assert abs(lso.power(t, y, dy, np.array([f]))[0] - lso.power_by_least_squares(t, y, dy, f)) < 1e-10


def test_pure_sinusoid_and_grid():
    rng = np.random.default_rng(0)
    t = np.sort(rng.uniform(0, 1000, 300))
    y = 3.0 + 2.0 * np.sin(2 * math.pi * t / 37.0 + 0.4)
    f = lso.autofrequency(t)
    base = t.max() - t.min()
    assert abs(f[0] - 0.1 / base) < 1e-15 and abs((f[1] - f[0]) - 0.2 / base) < 1e-15
    assert abs(f[-1] - 5 * 0.5 * 300 / base) <= 0.2 / base
    p = lso.power(t, y, None, f)
    assert p.max() <= 1.0 + 1e-12 and abs(f[np.argmax(p)] - 1 / 37.0) < 0.2 / base
    assert p.max() > 0.98                       # (the grid does not hit the frequency exactly)
    assert lso.power(t, y, None, np.array([1 / 37.0]))[0] > 1 - 1e-12     # noiseless: the sinusoid + floating mean explain everything


def test_class_surface_matches_oracle_and_astropy_conventions():
    t, y, e = _data(150)
    with mock.patch.object(_hip, "lomb_scargle", ob.lomb_scargle), mock.patch.object(_hip, "lomb_scargle_fast", ob.lomb_scargle_fast), mock.patch.object(L, "_compute_device", lambda: torch.device("cpu")):
        ls = L.LombScargle(torch.as_tensor(t), torch.as_tensor(y), torch.as_tensor(e))
        f = ls.autofrequency(nyquist_factor=5)
        assert np.allclose(f, lso.autofrequency(t))
        p = ls.power(f, assume_regular_frequency=True)                  # astropy's 'auto': the FFT approximation on this grid
        assert isinstance(p, np.ndarray) and p.shape == f.shape and f.size > 200
        assert np.allclose(p, lso.power_fast(t, y, e, f[0], f[1] - f[0], f.size), rtol=0, atol=1e-13)
        assert np.array_equal(p, ls.power(f)) and np.array_equal(p, ls.power(f, method="fast"))
        ps = ls.power(f, method="slow")
        assert np.allclose(ps, lso.power(t, y, e, f), rtol=0, atol=1e-13)
        assert 1e-6 < np.abs(ps - p).max() < 0.05 and np.abs(ps - p)[:100].max() < 1e-3      # an approximation, best at low frequencies
        assert np.allclose(ls.power(f[:150]), lso.power(t, y, e, f[:150]), rtol=0, atol=1e-13)   # short grid: exact sums
        irregular = np.sort(np.concatenate([f[:300], f[300:] * 1.0003]))
        assert np.allclose(ls.power(irregular), lso.power(t, y, e, irregular), rtol=0, atol=1e-13)
        with pytest.raises(ValueError):
            ls.power(irregular, method="fast")
        p = ps
        fap_max = ls.false_alarm_probability(p.max(), method="davies")
        fap_single = ls.false_alarm_probability(p, method="single")
        assert np.allclose(fap_single, lso.fap_single(p, t.size))
        assert abs(fap_max - lso.false_alarm_probability(p.max(), f[-1], t, e, "davies")) < 1e-12 * max(1.0, fap_max)
        bal = ls.false_alarm_probability(np.array([0.05, 0.3, 0.6]), method="baluev")
        assert np.all(np.diff(bal) < 0) and np.all((bal >= 0) & (bal <= 1))      # higher peak, smaller FAP
        with pytest.raises(NotImplementedError):
            ls.false_alarm_probability(0.5, method="bootstrap")
    with pytest.raises(NotImplementedError):
        L.LombScargle(t, y, e, nterms=2)
    with pytest.raises(NotImplementedError):
        L.LombScargleMultiband(t, y, np.zeros_like(t)).power(np.array([0.01]))       # astropy's default method is 'flexible'
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        L.LombScargle(t, y, e).power(np.array([0.01]))          # no GPU here, and no silent CPU path


def test_batched_seeding_finds_the_injected_periods():
    ts, ys, es, per = [], [], [], []
    for i in range(3):
        (t, y, e), p = syn.cfg3_lightcurve(i, n_obs=256)
        ts.append(t.double()); ys.append(y.double()); es.append(e.double()); per.append(p)
    T, Y, E = torch.stack(ts), torch.stack(ys), torch.stack(es)
    with mock.patch.object(_hip, "lomb_scargle", ob.lomb_scargle), mock.patch.object(_hip, "lomb_scargle_fast", ob.lomb_scargle_fast):
        freqs, pows, grid = L.seed_frequencies(T, Y, E, num_peaks=2)
    assert freqs.shape == (3, 2) and grid.ndim == 1
    for b in range(3):        # the leading period (amplitude 1) is the highest peak, to within one grid step
        assert abs(freqs[b, 0] - 1.0 / per[b]) < 1.5 * (grid[1] - grid[0]), (freqs[b], per[b])


def test_multiband_fast_periodogram_vs_oracle():
    """``LombScargleMultiband(...).power(f, method='fast')`` as ``pgmuvi/multiband_ls_significance.py`` uses it: per-band powers
    (here through the oracle stand-in of the HIP kernel) combined with astropy's weights (each band's summed squared power:
    ``oracle/ls_oracle.py::multiband_fast``); a common period in every band is the highest peak."""
    rng = np.random.default_rng(3)
    period = 37.0
    ts, ys, bs, es = [], [], [], []
    for b, (nb, amp) in enumerate(((60, 1.0), (45, 0.6), (80, 0.3))):
        t = np.sort(rng.uniform(0, 400, nb))
        e = 0.05 + 0.05 * rng.random(nb)
        ts.append(t); bs.append(np.full(nb, b)); es.append(e)
        ys.append(2.0 * b + amp * np.sin(2 * np.pi * t / period + 0.3 * b) + e * rng.standard_normal(nb))
    t, y, bands, e = map(np.concatenate, (ts, ys, bs, es))
    with mock.patch.object(_hip, "lomb_scargle", ob.lomb_scargle), mock.patch.object(_hip, "lomb_scargle_fast", ob.lomb_scargle_fast), mock.patch.object(L, "_compute_device", lambda: torch.device("cpu")):
        for dy in (e, None):
            mb = L.LombScargleMultiband(t, y, bands, dy=dy)
            f = mb.autofrequency(nyquist_factor=2)
            assert np.array_equal(f, lso.autofrequency(t, nyquist_factor=2))
            p = mb.power(f, method="fast", sb_method="slow")
            assert p.shape == f.shape and np.allclose(p, lso.multiband_fast(t, y, bands, dy, f), rtol=0, atol=1e-13)
            pa = mb.power(f, method="fast")                            # per-band periodograms by astropy's 'auto' rule
            assert np.allclose(pa, lso.multiband_fast(t, y, bands, dy, f, sb_auto=True), rtol=0, atol=1e-13)
            p = pa
            assert abs(1.0 / f[np.argmax(p)] - period) < 0.5
            f2, p2 = mb.autopower(method="fast", nyquist_factor=2)
            assert np.array_equal(f2, f) and np.array_equal(p2, p)
