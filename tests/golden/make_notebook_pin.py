#!/usr/bin/env python3
"""Regenerates tests/golden/notebook_pin_1d.npz.  RUN IN THE BUILD CONTAINER ONLY.

The one place the reference holds a NUMBER for the hot path is a recorded cell output of
``docs/source/notebooks/PGMUVI_comparison_with_other_codes.ipynb`` (the "pgmuvi -- 1D
spectral mixture GP" cell): after ``lc_1d.fit(model='1D', num_mixtures=2,
training_iter=1000, miniter=50, lr=0.05)`` on a seeded synthetic light curve (89 points)
the notebook printed

    initial  mean_module.constant 0.028102993965148926, mixture_weights [0.4779, 0.4779],
             mixture_means [0.0067, 0.0154], mixture_scales [0.0053, 0.0037]
    final    loss: -1.562, fitted frequencies: [0.00665436 0.0151593]

(the initial frequencies come from a Lomb-Scargle seeding that needs astropy, absent here, so
they are re-entered from the 4-digit printout through ``fit(guess=...)``).

This script rebuilds that light curve by *importing the reference* (against the
``pgmuvi_amd.gpytorch`` shim, evaluation by the CPU oracle), runs the same fit with the
reference's own ``Lightcurve.fit``/``trainers.train`` and stores

  * the recorded notebook numbers (data, typed in from the cell output),
  * the light curve as the GP sees it (transformed x, y, noise variance),
  * the hyper-parameters and loss this run ends at,

so that the GPU box (no reference there) can re-evaluate the HIP path at the end point and
compare with both the oracle value and the notebook's recorded loss.  Numbers only; no reference
source text is stored.
"""
import os
import sys
import warnings
from unittest import mock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, "/root/reference")

NOTEBOOK = dict(                      # recorded cell output (see module docstring)
    nb_n_points=89,
    nb_init_constant=0.028102993965148926,
    nb_init_weights=[0.4779, 0.4779],
    nb_init_means=[0.0067, 0.0154],
    nb_init_scales=[0.0053, 0.0037],
    nb_final_loss=-1.562,
    nb_final_freqs=[0.00665436, 0.0151593],
)


def build_lightcurve():
    """The notebook's dataset cell: 2-D chromatic two-period light curve, best-sampled band."""
    from pgmuvi.lightcurve import Lightcurve as LC
    from pgmuvi.synthetic import make_multi_sinusoid_chromatic_2d
    cfg = dict(components=[{"period": 150.0, "amplitude_fraction": 1.0, "phase": 0.0},
                           {"period": 66.0, "amplitude_fraction": 0.3, "phase": np.pi / 2 * 0.85}],
               t_span=150 * 2.3, n_per_band=(25, 100), wavelengths=[0.8, 1.2, 2.2],
               amplitude_law="extinction", noise_level=0.05, seed=0)
    lc_2d = make_multi_sinusoid_chromatic_2d(**cfg).double()
    waves, counts = np.unique(lc_2d.xdata[:, 1], return_counts=True)
    sel = lc_2d.xdata[:, 1] == waves[np.argmax(counts)]
    return LC(lc_2d.xdata[sel][:, 0], lc_2d.ydata[sel], yerr=lc_2d.yerr[sel]).double()


def run_fit_ls_seeded(lc, backend, ls_backend, max_iter=1000):
    """The notebook's fit cell exactly as written (no ``guess``): the initial frequencies come from the reference's
    own ``fit_LS`` through the astropy-shaped shim (``pgmuvi_amd.lombscargle.install_as_astropy``)."""
    import torch
    from pgmuvi_amd import _hip, lombscargle
    with mock.patch.object(_hip, "mll_value_grad", backend), mock.patch.object(_hip, "lomb_scargle", ls_backend), \
            mock.patch.object(lombscargle, "_compute_device", lambda: torch.device("cpu")):
        return lc.fit(model="1D", num_mixtures=2, training_iter=max_iter, miniter=50, lr=0.05)


def run_fit(lc, backend):
    """The notebook's fit cell, initial frequencies/scales re-entered from the printout."""
    from pgmuvi_amd import _hip
    guess = {"covar_module.mixture_means": torch.tensor(NOTEBOOK["nb_init_means"], dtype=torch.float64),
             "covar_module.mixture_scales": torch.tensor(NOTEBOOK["nb_init_scales"], dtype=torch.float64)}
    with mock.patch.object(_hip, "mll_value_grad", backend):
        return lc.fit(model="1D", num_mixtures=2, training_iter=1000, miniter=50, lr=0.05, guess=guess)


def main():
    import pgmuvi_amd
    import _oracle_backend as ob
    pgmuvi_amd.install_as_gpytorch()
    warnings.simplefilter("ignore")
    torch.manual_seed(0)
    lc = build_lightcurve()
    res = run_fit(lc, ob.mll_value_grad)
    cov = lc.model.covar_module
    out = dict(NOTEBOOK)
    out.update(
        x=lc._xdata_transformed.detach().double().numpy(),
        y=lc._ydata_transformed.detach().double().numpy(),
        noise=lc.likelihood.noise.detach().double().numpy(),
        n_iter=len(res["loss"]),
        first_loss=float(res["loss"][0]),
        final_loss=float(res["loss"][-1]),
        final_w=cov.mixture_weights.detach().double().numpy().reshape(-1),
        final_mu=cov.mixture_means.detach().double().numpy().reshape(-1),
        final_v=cov.mixture_scales.detach().double().numpy().reshape(-1),
        final_c=float(lc.model.mean_module.constant.detach()),
    )
    np.savez(os.path.join(HERE, "notebook_pin_1d.npz"), **out)
    print("loss", out["first_loss"], "->", out["final_loss"], "after", out["n_iter"], "iterations; notebook", NOTEBOOK["nb_final_loss"])
    print("freqs", out["final_mu"], "; notebook", NOTEBOOK["nb_final_freqs"])


if __name__ == "__main__":
    main()
