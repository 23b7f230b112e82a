"""pgmuvi_amd -- MI355X-native exact-GP hot path of ICSM/pgmuvi.

Spectral-mixture kernel assembly + Cholesky marginal log-likelihood + gradient as
hand-written HIP kernels behind a C ABI (``include/pgmuvi_hip.h``), exposed through
the GPyTorch operator surface pgmuvi uses (``pgmuvi_amd.gpytorch``).
"""
import sys

from . import _hip, gpytorch, synthetic  # noqa: F401
from .mll_function import sm_exact_mll  # noqa: F401
from .trainers import train  # noqa: F401

__all__ = ["gpytorch", "install_as_gpytorch", "sm_exact_mll", "train", "synthetic"]


def install_as_gpytorch(force: bool = False):
    """Make ``import gpytorch`` resolve to ``pgmuvi_amd.gpytorch`` (drop-in for pgmuvi).

    Refuses to shadow a real GPyTorch that is already imported unless ``force``."""
    existing = sys.modules.get("gpytorch")
    if existing is not None and existing is not gpytorch and not force:
        raise RuntimeError("a different 'gpytorch' is already imported; pass force=True to replace it")
    names = ["constraints", "distributions", "kernels", "likelihoods", "means", "mlls", "models", "priors",
             "settings", "utils", "variational", "module", "lazy"]
    sys.modules["gpytorch"] = gpytorch
    for n in names:
        sys.modules[f"gpytorch.{n}"] = getattr(gpytorch, n, None) or __import__(f"pgmuvi_amd.gpytorch.{n}", fromlist=[n])
    sys.modules["gpytorch.likelihoods.likelihood"] = gpytorch.likelihoods.likelihood
    sys.modules["gpytorch.mlls.marginal_log_likelihood"] = gpytorch.mlls.marginal_log_likelihood
    sys.modules["gpytorch.utils.errors"] = gpytorch.utils.errors
    return gpytorch
