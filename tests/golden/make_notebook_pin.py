#!/usr/bin/env python3
"""Regenerates tests/golden/notebook_pin_1d.npz and notebook_pin_2d.npz.  RUN IN THE BUILD CONTAINER ONLY.

The reference holds recorded NUMBERS for the hot path in the cell outputs of
``docs/source/notebooks/PGMUVI_comparison_with_other_codes.ipynb``.  The "pgmuvi -- 2D spectral
mixture GP" cell is described at ``NOTEBOOK_2D`` below; the "pgmuvi -- 1D
spectral mixture GP" cell: after ``lc_1d.fit(model='1D', num_mixtures=2,
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


# The "pgmuvi -- 2D spectral mixture GP" cell (notebook lines 1463-1552): ``lc_2d.fit(model='2D', num_mixtures=2,
# training_iter=1000, miniter=50, lr=0.05)`` on the 225-point (89 + 73 + 63), three-band light curve printed
#     initial  mean_module.constant 0.028102993965148926, mixture_weights [0.6931, 0.6931],
#              mixture_means 9.4067 (all four), mixture_scales 0.6931 (all four)     <- no random draw anywhere
#     progress bar stopped at 348/1000: "Average change in loss over the last 30 iterations was 9.44400166417925e-06"
#     final    loss: 0.904, fitted time frequencies: [13.842627 13.842627]
# The start is deterministic, so stop iteration, loss and frequencies pin the 2-D kernel form: GPyTorch's
# prod_d sum_q (dim_order 0) or sum_q prod_d (dim_order 1) give different trajectories.
NOTEBOOK_2D = dict(
    nb_n_points=225,
    nb_band_counts=[89, 73, 63],
    nb_init_constant=0.028102993965148926,
    nb_init_weights=[0.6931, 0.6931],
    nb_init_means=[9.4067, 9.4067, 9.4067, 9.4067],
    nb_init_scales=[0.6931, 0.6931, 0.6931, 0.6931],
    nb_progress_bar_stop=348,
    nb_stopval=9.44400166417925e-06,
    nb_final_loss=0.904,
    nb_final_time_freqs=[13.842627, 13.842627],
)


def build_lightcurve_2d():
    """The notebook's dataset cell: 2-D chromatic two-period light curve, three bands, seed 0."""
    from pgmuvi.synthetic import make_multi_sinusoid_chromatic_2d
    cfg = dict(components=[{"period": 150.0, "amplitude_fraction": 1.0, "phase": 0.0},
                           {"period": 66.0, "amplitude_fraction": 0.3, "phase": np.pi / 2 * 0.85}],
               t_span=150 * 2.3, n_per_band=(25, 100), wavelengths=[0.8, 1.2, 2.2],
               amplitude_law="extinction", noise_level=0.05, seed=0)
    return make_multi_sinusoid_chromatic_2d(**cfg).double()


def with_dim_order(backend, order):
    """The evaluation stand-in with the 2-D kernel form forced (argument 9 of ``_hip.mll_value_grad``)."""
    def forced(*a, **k):
        a = list(a)
        if len(a) > 8:
            a[8] = order
        else:
            k["dim_order"] = order
        return backend(*a, **k)
    return forced


def run_fit_2d(lc, backend, order=0):
    """The notebook's 2-D fit cell exactly as written."""
    from pgmuvi_amd import _hip
    with mock.patch.object(_hip, "mll_value_grad", with_dim_order(backend, order)):
        return lc.fit(model="2D", num_mixtures=2, training_iter=1000, miniter=50, lr=0.05)


def build_lightcurve():
    """The notebook's dataset cell: 2-D chromatic two-period light curve, best-sampled band."""
    from pgmuvi.lightcurve import Lightcurve as LC
    lc_2d = build_lightcurve_2d()
    waves, counts = np.unique(lc_2d.xdata[:, 1], return_counts=True)
    sel = lc_2d.xdata[:, 1] == waves[np.argmax(counts)]
    return LC(lc_2d.xdata[sel][:, 0], lc_2d.ydata[sel], yerr=lc_2d.yerr[sel]).double()


def run_fit_ls_seeded(lc, backend, ls_backend, max_iter=1000):
    """The notebook's fit cell exactly as written (no ``guess``): the initial frequencies come from the reference's
    own ``fit_LS`` through the astropy-shaped shim (``pgmuvi_amd.lombscargle.install_as_astropy``)."""
    import torch
    from pgmuvi_amd import _hip, lombscargle
    import _oracle_backend as ob
    with mock.patch.object(_hip, "mll_value_grad", backend), mock.patch.object(_hip, "lomb_scargle", ls_backend), \
            mock.patch.object(_hip, "lomb_scargle_fast", ob.lomb_scargle_fast), \
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


def main_2d():
    """notebook_pin_2d.npz: the 2-D cell re-run with the reference's own ``Lightcurve.fit`` for both kernel forms, plus
    everything the GPU box needs to repeat the fit without the reference (data as the GP sees it, the constraints
    ``set_default_constraints`` registered, parameter dtype)."""
    import pgmuvi_amd
    import _oracle_backend as ob
    pgmuvi_amd.install_as_gpytorch()
    warnings.simplefilter("ignore")
    out = dict(NOTEBOOK_2D)
    for order in (0, 1):
        torch.manual_seed(0)
        lc = build_lightcurve_2d()
        res = run_fit_2d(lc, ob.mll_value_grad, order)
        cov = lc.model.covar_module
        tag = f"order{order}_"
        out.update({
            tag + "n_losses": len(res["loss"]),
            tag + "loss": np.asarray([float(v) for v in res["loss"]]),
            tag + "final_w": cov.mixture_weights.detach().double().numpy().reshape(-1),
            tag + "final_mu": cov.mixture_means.detach().double().numpy().reshape(2, 2),
            tag + "final_v": cov.mixture_scales.detach().double().numpy().reshape(2, 2),
            tag + "final_c": float(lc.model.mean_module.constant.detach()),
        })
        print(f"dim_order {order}: {len(res['loss'])} losses (loop index {len(res['loss']) - 1} at the break), final loss "
              f"{float(res['loss'][-1]):.6f}, time frequencies {cov.mixture_means.detach().numpy()[:, 0, 0]}")
    cons = dict(lc.model.named_constraints())
    par = dict(lc.model.named_parameters())
    out.update(
        x=lc._xdata_transformed.detach().double().numpy(),
        y=lc._ydata_transformed.detach().double().numpy(),
        noise=lc.likelihood.noise.detach().double().numpy(),
        param_dtype=str(par["covar_module.raw_mixture_means"].dtype),
        constant_bounds=np.asarray([float(cons["mean_module.raw_constant_constraint"].lower_bound),
                                    float(cons["mean_module.raw_constant_constraint"].upper_bound)]),
        means_bounds=np.asarray([float(cons["covar_module.raw_mixture_means_constraint"].lower_bound),
                                 float(cons["covar_module.raw_mixture_means_constraint"].upper_bound)]),
    )
    np.savez(os.path.join(HERE, "notebook_pin_2d.npz"), **out)
    print("notebook: stop", NOTEBOOK_2D["nb_progress_bar_stop"], "loss", NOTEBOOK_2D["nb_final_loss"], "freqs", NOTEBOOK_2D["nb_final_time_freqs"])


if __name__ == "__main__":
    main()
    main_2d()
