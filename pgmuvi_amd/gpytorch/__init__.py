"""GPyTorch-shaped operator surface of the MI355X exact-GP hot path.

This package reproduces, name for name, the part of GPyTorch that pgmuvi's
``gps.py`` / ``trainers.py`` / ``lightcurve.py`` call, with
the arithmetic of ``mll(model(x), y)`` + ``backward()`` executed by hand-written HIP
kernels.  ``pgmuvi_amd.install_as_gpytorch()`` registers it as ``gpytorch`` in
``sys.modules`` so pgmuvi's own code imports it unchanged.
"""
from . import constraints, distributions, kernels, likelihoods, means, mlls, models, priors, settings, utils, variational
from .module import Module
from .mlls import ExactMarginalLogLikelihood
from .distributions import MultivariateNormal

__version__ = "pgmuvi_amd-0.1"
__all__ = ["Module", "constraints", "distributions", "kernels", "likelihoods", "means", "mlls", "models",
           "priors", "settings", "utils", "variational", "ExactMarginalLogLikelihood", "MultivariateNormal"]
