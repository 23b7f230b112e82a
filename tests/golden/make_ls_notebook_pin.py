#!/usr/bin/env python3
"""Regenerates tests/golden/ls_notebook_pin.npz.  RUN IN THE BUILD CONTAINER ONLY.

The reference holds recorded Lomb-Scargle numbers in the cell outputs of
``docs/source/notebooks/PGMUVI_Lomb_Scargle.ipynb`` (``fit_LS``: ``pgmuvi/lightcurve.py:4214-4611``):

  cells 10 / 12 / 34   ``lc1d.fit_LS(num_peaks=5)`` on band 0 (38 points) of the seeded one-period three-band light curve:
                       five peak frequencies to 6 digits with their significance flags; grid length 475
  cell 20              peak period / height / prominence / area fraction of the multiband periodogram, default and
                       ``use_best_band_init=True``
  cell 34              ``fit_LS(num_peaks=8)`` on the two-period light curve merged with a densely sampled fourth band

This script rebuilds those light curves by importing the reference's own ``pgmuvi.synthetic`` (numbers only are
stored: t, y, dy, band) next to the recorded values typed in from the cell outputs, so that the GPU box, which has no
reference, can run the HIP periodogram on the same data.
"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

RECORDED = dict(
    nb1d_n_points=38,
    nb1d_grid_length=475,
    nb1d_peak_freqs=[0.006704, 0.039931, 0.061499, 0.248038, 0.069660],
    nb1d_peak_significant=[True, False, False, False, False],
    # cell 20: peak_period, peak_height, peak_prominence, area_fraction
    nbmb_default=[149.170715, 0.909449, 0.579050, 0.016485],
    nbmb_best_band=[149.170715, 0.992789, 0.859738, 0.020845],
    nbmb2_band_counts=[38, 35, 33, 171],
    nbmb2_peak_freqs=[0.006704, 1.908814, 1.112528, 1.148086, 1.134096, 1.381843, 0.787834, 0.014865],
    nbmb2_peak_significant=[True, False, False, False, False, False, False, False],
)

TWO_PERIODS = [{"period": 150.0, "amplitude_fraction": 1.0, "phase": 0.0},
               {"period": 66.0, "amplitude_fraction": 0.3, "phase": np.pi / 2 * 0.85}]


def _label_bands(lc):
    wl = np.asarray(lc.xdata[:, 1], dtype=float)
    names = {w: str(i) for i, w in enumerate(np.unique(wl))}
    lc.band = np.array([f"band {names[w]}" for w in wl], dtype=object)
    return lc


def build_one_period():
    """Notebook cell 6."""
    from pgmuvi import synthetic
    return _label_bands(synthetic.make_chromatic_sinusoid_2d(period=150, t_span=150 * 2.3, n_per_band=(25, 40),
                                                             wavelengths=[0.8, 1.2, 2.2], amplitude_law="extinction", seed=0))


def build_two_periods_with_dense_band():
    """Notebook cells 23-25."""
    from pgmuvi import synthetic
    lc = _label_bands(synthetic.make_multi_sinusoid_chromatic_2d(components=TWO_PERIODS, t_span=150 * 2.3, n_per_band=(25, 40),
                                                                 wavelengths=[0.8, 1.2, 2.2], amplitude_law="extinction",
                                                                 noise_level=0.05, seed=0))
    hs = synthetic.make_multi_sinusoid_chromatic_2d(components=TWO_PERIODS, t_span=150 * 2.3, n_per_band=(100, 250),
                                                    wavelengths=[1.02], amplitude_law="extinction", noise_level=0.05, seed=1)
    hs.band = np.repeat("band 3", len(hs.xdata))
    return lc.merge(hs)


def main():
    import pgmuvi_amd
    from pgmuvi_amd import lombscargle
    pgmuvi_amd.install_as_gpytorch()
    lombscargle.install_as_astropy(force=True)
    warnings.simplefilter("ignore")
    out = dict(RECORDED)
    for tag, lc in (("one", build_one_period()), ("two", build_two_periods_with_dense_band())):
        out[tag + "_t"] = lc.xdata[:, 0].numpy().astype(np.float64)
        out[tag + "_wavelength"] = lc.xdata[:, 1].numpy().astype(np.float64)
        out[tag + "_y"] = lc.ydata.numpy().astype(np.float64)
        out[tag + "_dy"] = lc.yerr.numpy().astype(np.float64)
    np.savez(os.path.join(HERE, "ls_notebook_pin.npz"), **out)
    print({k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if k.startswith(("one", "two"))})


if __name__ == "__main__":
    main()
