"""TEST-ONLY pytest plugin: lets the reference's OWN test files (``/root/reference/tests``) run against the shim in this
container (no GPU): registers ``pgmuvi_amd.gpytorch`` as ``gpytorch`` and the Lomb-Scargle shim as ``astropy.timeseries``,
and replaces the HIP entry points by the oracle stand-ins of ``tests/_oracle_backend.py`` for the whole session.
Loaded with ``-p _reference_suite_plugin`` by ``tests/test_dropin_reference.py``; the product never imports it."""
import warnings

import pgmuvi_amd
from pgmuvi_amd import _hip, lombscargle

import _oracle_backend as ob

pgmuvi_amd.install_as_gpytorch()
try:
    lombscargle.install_as_astropy()
except Exception:                                              # a real astropy is importable: leave it alone
    pass
import torch

for _name in ("predict", "lomb_scargle", "lomb_scargle_fast", "mll_dense", "mll_kernel_value_grad", "predict_dense"):
    setattr(_hip, _name, getattr(ob, _name))
_hip.mll_value_grad = ob.mll_value_grad_remember              # (the predict stand-in needs the last evaluation's inputs)
_hip.require_gpu = lambda *a, **k: None
lombscargle._compute_device = lambda: torch.device("cpu")
warnings.simplefilter("ignore")
