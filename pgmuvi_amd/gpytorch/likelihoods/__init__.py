"""``gpytorch.likelihoods`` (row A3): the noise the likelihood adds to the diagonal.

* ``FixedNoiseGaussianLikelihood(noise=yerr**2)`` -- heteroscedastic vector, pgmuvi's
  default whenever ``yerr`` is given (``pgmuvi/lightcurve.py:2778-2789``; optional
  ``learn_additional_noise`` at 2790-2798);
* ``GaussianLikelihood()`` -- learned scalar ``raw_noise`` with ``GreaterThan(1e-4)``
  (``lightcurve.py:2805-2807``; Interval from 3817-3829).
"""
from __future__ import annotations

import warnings

import torch

from .. import settings
from ..constraints import GreaterThan
from ..distributions import MultivariateNormal
from ..module import Module
from ..utils.errors import NumericalWarning
from . import likelihood
from .likelihood import Likelihood


class _HomoskedasticNoiseBase(Module):
    def __init__(self, noise_prior=None, noise_constraint=None, batch_shape=torch.Size(), num_tasks=1):
        super().__init__()
        if noise_constraint is None:
            noise_constraint = GreaterThan(1e-4)
        self.register_parameter("raw_noise", torch.nn.Parameter(torch.zeros(*batch_shape, num_tasks)))
        if noise_prior is not None:
            self.register_prior("noise_prior", noise_prior, lambda m: m.noise, lambda m, v: m._set_noise(v))
        self.register_constraint("raw_noise", noise_constraint)

    @property
    def noise(self):
        return self.raw_noise_constraint.transform(self.raw_noise)

    @noise.setter
    def noise(self, value):
        self._set_noise(value)

    def _set_noise(self, value):
        if not torch.is_tensor(value):
            value = torch.as_tensor(value).to(self.raw_noise)
        self.initialize(raw_noise=self.raw_noise_constraint.inverse_transform(value.to(self.raw_noise)))


class HomoskedasticNoise(_HomoskedasticNoiseBase):
    pass


class FixedGaussianNoise(Module):
    def __init__(self, noise):
        super().__init__()
        min_noise = settings.min_fixed_noise.value(noise.dtype)
        if noise.lt(min_noise).any():
            warnings.warn(
                "Very small noise values detected. This will likely lead to numerical instabilities. "
                f"Rounding small noise values up to {min_noise}.", NumericalWarning)
            noise = noise.clamp_min(min_noise)
        self.noise = noise

    def _apply(self, fn, *a, **k):
        self.noise = fn(self.noise)
        return super()._apply(fn, *a, **k)


class _GaussianLikelihoodBase(Likelihood):
    def _add_noise(self, covar, n, training):
        raise NotImplementedError

    def marginal(self, function_dist, *params, **kwargs):
        covar = function_dist.lazy_covariance_matrix
        return MultivariateNormal(function_dist.mean, self._add_noise(covar, function_dist.mean.shape[-1], **kwargs))


class GaussianLikelihood(_GaussianLikelihoodBase):
    def __init__(self, noise_prior=None, noise_constraint=None, batch_shape=torch.Size(), **kwargs):
        super().__init__()
        self.noise_covar = HomoskedasticNoise(noise_prior=noise_prior, noise_constraint=noise_constraint,
                                              batch_shape=batch_shape)

    @property
    def noise(self):
        return self.noise_covar.noise

    @noise.setter
    def noise(self, value):
        self.noise_covar.initialize(noise=value)

    @property
    def raw_noise(self):
        return self.noise_covar.raw_noise

    @raw_noise.setter
    def raw_noise(self, value):
        self.noise_covar.initialize(raw_noise=value)

    def _add_noise(self, covar, n, **kwargs):
        return covar.add_noise(noise_scalar=self.noise.reshape(()))


class FixedNoiseGaussianLikelihood(_GaussianLikelihoodBase):
    def __init__(self, noise, learn_additional_noise=False, batch_shape=torch.Size(), **kwargs):
        super().__init__()
        self.noise_covar = FixedGaussianNoise(noise=noise)
        self.second_noise_covar = None
        if learn_additional_noise:
            self.second_noise_covar = HomoskedasticNoise(
                noise_prior=kwargs.get("noise_prior"), noise_constraint=kwargs.get("noise_constraint"),
                batch_shape=batch_shape)

    @property
    def noise(self):
        return self.noise_covar.noise + self.second_noise

    @noise.setter
    def noise(self, value):
        self.noise_covar.noise = value

    @property
    def second_noise(self):
        return 0 if self.second_noise_covar is None else self.second_noise_covar.noise

    @second_noise.setter
    def second_noise(self, value):
        if self.second_noise_covar is None:
            raise RuntimeError(
                "Attempting to set secondary learned noise for FixedNoiseGaussianLikelihood, "
                "but learn_additional_noise must have been False!")
        self.second_noise_covar.initialize(noise=value)

    def _add_noise(self, covar, n, noise=None, **kwargs):
        fixed = self.noise_covar.noise if noise is None else noise
        ns = None if self.second_noise_covar is None else self.second_noise_covar.noise.reshape(())
        if fixed.shape[-1] != n:
            # GPyTorch: evaluating at other inputs without passing `noise=` is a no-op
            warnings.warn(
                "You have passed data through a FixedNoiseGaussianLikelihood that did not match the size "
                "of the fixed noise, *and* you did not specify noise. This is treated as a no-op.", NumericalWarning)
            return covar.add_noise(noise_scalar=ns) if ns is not None else covar
        return covar.add_noise(noise_vec=fixed, noise_scalar=ns)
