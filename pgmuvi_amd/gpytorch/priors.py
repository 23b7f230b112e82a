"""``gpytorch.priors`` used by pgmuvi (``pgmuvi/lightcurve.py:35, 3280-3322``;
subclassed in ``pgmuvi/priors.py:137-460``): a Prior is a torch Distribution that is
also a Module so it can be registered; ``log_prob`` sums are added to the MLL
before the division by N (ExactMarginalLogLikelihood._add_other_terms)."""
from __future__ import annotations

import torch
from torch.distributions import HalfCauchy, LogNormal, Normal, Uniform, Gamma
from torch.nn import Module as TModule

from .module import Module


class Prior(torch.distributions.Distribution, Module):
    def transform(self, x):
        return self._transform(x) if getattr(self, "_transform", None) is not None else x

    def log_prob(self, x):
        return super().log_prob(self.transform(x))


def _bufferize(module, names):
    # keep distribution parameters as buffers so .cuda()/.double() move them
    for n in names:
        val = getattr(module, n)
        try:
            delattr(module, n)
        except AttributeError:
            pass
        module.register_buffer(n, val.clone() if torch.is_tensor(val) else torch.as_tensor(val))


class NormalPrior(Prior, Normal):
    def __init__(self, loc, scale, validate_args=False, transform=None):
        TModule.__init__(self)
        Normal.__init__(self, loc=torch.as_tensor(loc, dtype=torch.get_default_dtype()) if not torch.is_tensor(loc) else loc,
                        scale=torch.as_tensor(scale, dtype=torch.get_default_dtype()) if not torch.is_tensor(scale) else scale,
                        validate_args=validate_args)
        Module.__init__(self)
        _bufferize(self, ("loc", "scale"))
        self._transform = transform

    def expand(self, batch_shape):
        return NormalPrior(self.loc.expand(batch_shape), self.scale.expand(batch_shape))


class LogNormalPrior(Prior, LogNormal):
    def __init__(self, loc, scale, validate_args=None, transform=None):
        TModule.__init__(self)
        LogNormal.__init__(self, loc=loc, scale=scale, validate_args=validate_args)
        Module.__init__(self)
        self._transform = transform

    def expand(self, batch_shape):
        return LogNormalPrior(self.loc.expand(batch_shape), self.scale.expand(batch_shape))


class UniformPrior(Prior, Uniform):
    def __init__(self, a, b, validate_args=None, transform=None):
        TModule.__init__(self)
        Uniform.__init__(self, a, b, validate_args=validate_args)
        Module.__init__(self)
        self._transform = transform

    def expand(self, batch_shape):
        return UniformPrior(self.low.expand(batch_shape), self.high.expand(batch_shape))


class HalfCauchyPrior(Prior, HalfCauchy):
    def __init__(self, scale, validate_args=None, transform=None):
        TModule.__init__(self)
        HalfCauchy.__init__(self, scale=scale, validate_args=validate_args)
        Module.__init__(self)
        self._transform = transform


class GammaPrior(Prior, Gamma):
    def __init__(self, concentration, rate, validate_args=False, transform=None):
        TModule.__init__(self)
        Gamma.__init__(self, concentration=concentration, rate=rate, validate_args=validate_args)
        Module.__init__(self)
        self._transform = transform
