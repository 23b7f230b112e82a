"""``gpytorch.distributions.MultivariateNormal`` as pgmuvi uses it
(``pgmuvi/gps.py:20, 220``): a mean plus a *lazy* covariance.  ``log_prob`` of a
training-mode distribution whose covariance is the spectral-mixture kernel (+ noise)
dispatches to the fused HIP evaluation; eval-mode predictive distributions carry
their pointwise variances."""
from __future__ import annotations

import math

import torch

from .lazy import DenseCovariance, DiagCovariance, LazySMCovariance


class Distribution:
    pass


class MultivariateNormal(Distribution):
    def __init__(self, mean, covariance_matrix, validate_args=False):
        self.loc = mean
        self._covar = covariance_matrix

    # --- shape / moments ---------------------------------------------------
    @property
    def mean(self):
        return self.loc

    @property
    def event_shape(self):
        return self.loc.shape[-1:]

    @property
    def batch_shape(self):
        return self.loc.shape[:-1]

    @property
    def lazy_covariance_matrix(self):
        return self._covar

    @property
    def covariance_matrix(self):
        c = self._covar
        return c.to_dense() if hasattr(c, "to_dense") else c

    @property
    def variance(self):
        c = self._covar
        if hasattr(c, "diagonal_values"):
            return c.diagonal_values()
        return torch.diagonal(c, dim1=-2, dim2=-1)

    @property
    def stddev(self):
        return self.variance.clamp_min(0.0).sqrt()

    def confidence_region(self):
        std2 = self.stddev.mul(2)
        return self.mean.sub(std2), self.mean.add(std2)

    def rsample(self, sample_shape=torch.Size()):
        # independent marginals (pointwise variances) -- the joint posterior covariance
        # over test points is not materialised on this path
        eps = torch.randn(*sample_shape, *self.loc.shape, dtype=self.loc.dtype, device=self.loc.device)
        return self.loc + eps * self.stddev

    def sample(self, sample_shape=torch.Size()):
        with torch.no_grad():
            return self.rsample(sample_shape)

    # --- the hot path -------------------------------------------------------
    def log_prob(self, value):
        c = self._covar
        if isinstance(c, LazySMCovariance) and c.is_square:
            from ..mll_function import sm_exact_mll
            k = c.kernel
            n = value.shape[-1]
            per_datum = sm_exact_mll(c.x1, value, self.loc, c.noise_vec, c.noise_scalar,
                                     k.mixture_weights, k.mixture_means, k.mixture_scales, k.dim_order)
            return per_datum * n
        if isinstance(c, DenseCovariance) and c.fused and value.dim() == 1:
            # a composed stationary kernel with a device program: build, sweep and gradient contraction fused (no N x N
            # matrix in torch); anything else below takes the matrix torch builds through the dense back-end
            from ..mll_function import kernel_exact_mll
            n = value.shape[-1]
            prog = c.program
            ns = c.noise_scalar
            if ns is not None and not torch.is_tensor(ns):
                ns = torch.as_tensor(ns, dtype=value.dtype, device=value.device)
            return kernel_exact_mll(prog, c.x, value, self.loc, c.noise_vec, ns, prog.theta()) * n
        if isinstance(c, DenseCovariance) and c.is_square:
            # any other kernel: the matrix torch built goes through the same factorisation sweep (dense back-end)
            from ..mll_function import dense_exact_mll
            n = value.shape[-1]
            return dense_exact_mll(c.to_dense(), value - self.loc) * n
        raise NotImplementedError(
            "pgmuvi_amd evaluates log_prob for training-mode exact GPs (spectral-mixture: fused; other kernels: dense back-end); "
            f"got covariance of type {type(c).__name__}.")

    def __add__(self, other):
        if isinstance(other, (int, float)) or torch.is_tensor(other):
            return MultivariateNormal(self.loc + other, self._covar)
        raise TypeError(type(other))

    def __getitem__(self, idx):
        c = self._covar
        if isinstance(c, DiagCovariance):
            return MultivariateNormal(self.loc[idx], DiagCovariance(c.var[idx]))
        raise NotImplementedError

    def __repr__(self):
        return f"MultivariateNormal(loc: {tuple(self.loc.shape)})"
