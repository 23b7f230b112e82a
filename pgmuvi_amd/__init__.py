"""pgmuvi_amd -- MI355X-native exact-GP hot path of ICSM/pgmuvi.

Spectral-mixture kernel assembly + Cholesky marginal log-likelihood + gradient as
hand-written HIP kernels behind a C ABI (``include/pgmuvi_hip.h``), exposed through
the GPyTorch operator surface pgmuvi uses (``pgmuvi_amd.gpytorch``).
"""
import sys

from . import _hip, gpytorch, synthetic  # noqa: F401
from .mll_function import sm_exact_mll  # noqa: F401
from .trainers import train  # noqa: F401

__all__ = ["gpytorch", "install_as_gpytorch", "install_native_trainer", "sm_exact_mll", "train", "synthetic"]


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


def install_native_trainer(fallback=None):
    """Route ``Lightcurve.fit()``'s optimiser loop (``pgmuvi/lightcurve.py:5870-5879`` calls the ``train`` it imported at
    ``:32``) to the device-resident loop ``pgmuvi_amd.trainers.train_native``; models outside its scope (priors, other
    kernels, data on the CPU) fall back to ``fallback`` -- by default the loop pgmuvi came with.  Call after
    ``import pgmuvi.lightcurve``.  Returns the replaced function."""
    import pgmuvi.lightcurve as lc_mod
    from . import trainers
    original = fallback or lc_mod.train

    def train(lightcurve=None, *args, **kwargs):
        dev_ok = lightcurve is not None and getattr(lightcurve, "_xdata_transformed", None) is not None and \
            lightcurve._xdata_transformed.is_cuda
        if dev_ok and not args:
            try:
                known = {k: v for k, v in kwargs.items() if k in ("maxiter", "miniter", "stop", "lr", "lossfn", "optim", "eps", "stopavg")}
                return trainers.train_native(lightcurve, **known)
            except NotImplementedError:
                pass
        return original(lightcurve, *args, **kwargs)

    train.__wrapped__ = original
    lc_mod.train = train
    return original
