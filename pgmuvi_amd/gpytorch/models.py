"""``gpytorch.models.ExactGP`` (``pgmuvi/gps.py:21, 205-220``) and import-level stubs.

Training mode: ``model(train_x)`` returns ``forward(train_x)`` -- the prior MVN with a
lazy covariance; nothing is computed until ``mll(output, y)``.
Eval mode: ``model(x_test)`` returns the posterior at the test inputs (SURVEY.md
section 8f row 1) computed by ``pgm_predict_f64`` from the factor of one fused
evaluation at the current hyper-parameters.
"""
from __future__ import annotations

import warnings

import torch

from . import settings
from .distributions import MultivariateNormal
from .lazy import DenseCovariance, DiagCovariance, LazySMCovariance
from .likelihoods import _GaussianLikelihoodBase
from .module import Module


class GP(Module):
    pass


class ExactGP(GP):
    def __init__(self, train_inputs, train_targets, likelihood):
        if train_inputs is not None and torch.is_tensor(train_inputs):
            train_inputs = (train_inputs,)
        if train_inputs is not None and not all(torch.is_tensor(t) for t in train_inputs):
            raise RuntimeError("Train inputs must be a tensor, or a list/tuple of tensors")
        if not isinstance(likelihood, _GaussianLikelihoodBase):
            raise RuntimeError("ExactGP can only handle Gaussian likelihoods")
        super().__init__()
        if train_inputs is not None:
            self.train_inputs = tuple(t.unsqueeze(-1) if t.ndimension() == 1 else t for t in train_inputs)
            self.train_targets = train_targets
        else:
            self.train_inputs = None
            self.train_targets = None
        self.likelihood = likelihood
        self.prediction_strategy = None

    def _apply(self, fn, *a, **k):
        if self.train_inputs is not None:
            self.train_inputs = tuple(fn(t) for t in self.train_inputs)
            self.train_targets = fn(self.train_targets)
        return super()._apply(fn, *a, **k)

    def set_train_data(self, inputs=None, targets=None, strict=True):
        if inputs is not None:
            if torch.is_tensor(inputs):
                inputs = (inputs,)
            inputs = tuple(t.unsqueeze(-1) if t.ndimension() == 1 else t for t in inputs)
            if strict and self.train_inputs is not None:
                for new, old in zip(inputs, self.train_inputs):
                    for attr in ("shape", "dtype", "device"):
                        if getattr(new, attr) != getattr(old, attr):
                            raise RuntimeError(f"Cannot modify {attr} of inputs (expected {getattr(old, attr)}, found {getattr(new, attr)}).")
            self.train_inputs = inputs
        if targets is not None:
            if strict and self.train_targets is not None:
                for attr in ("shape", "dtype", "device"):
                    if getattr(targets, attr) != getattr(self.train_targets, attr):
                        raise RuntimeError(f"Cannot modify {attr} of targets.")
            self.train_targets = targets
        self.prediction_strategy = None

    def train(self, mode=True):
        if mode:
            self.prediction_strategy = None
        return super().train(mode)

    def __call__(self, *args, **kwargs):
        inputs = [a.unsqueeze(-1) if a.ndimension() == 1 else a for a in args]
        if self.training:
            if self.train_inputs is None:
                raise RuntimeError("train_inputs, train_targets cannot be None in training mode. "
                                   "Call .eval() for prior predictions, or call .set_train_data() to add training data.")
            if settings.debug.on():
                if not all(a.shape == b.shape and torch.equal(a, b) for a, b in zip(self.train_inputs, inputs)):
                    raise RuntimeError("You must train on the training inputs!")
            # hand the module's own tensor to forward(): downstream identity checks stay cheap
            return super().__call__(*self.train_inputs, **kwargs) if len(inputs) == len(self.train_inputs) else super().__call__(*inputs, **kwargs)
        if self.train_inputs is None or self.train_targets is None:
            return super().__call__(*inputs, **kwargs)             # prior
        return self._posterior(inputs[0], **kwargs)

    # ---- eval mode -------------------------------------------------------------
    def _posterior(self, x_test, **kwargs):
        from .. import _hip
        from ..mll_function import _evaluate
        train_x = self.train_inputs[0]
        with torch.no_grad():
            prior = self.forward(train_x)
            marg = self.likelihood(prior)
            c = marg.lazy_covariance_matrix
            if isinstance(c, DenseCovariance):
                # dense back-end: factor A once, then K(x_train, x_test) and the prior variances from the kernel itself
                out = _hip.mll_dense(c.to_dense(), self.train_targets - marg.mean, 0.0, False)
                if bool((out["info"] != 0).any()):
                    from .utils.errors import NotPSDError
                    raise NotPSDError("Matrix not positive definite in eval-mode prediction.")
                test_prior = self.forward(x_test)
                cross = self._cross_covariance(train_x, x_test)
                kss = test_prior.lazy_covariance_matrix.diagonal_values()
                pm, pv = _hip.predict_dense(out["workspace"], cross, kss, test_prior.mean)
                return MultivariateNormal(pm.to(x_test.dtype), DiagCovariance(pv.to(x_test.dtype)))
            if not isinstance(c, LazySMCovariance):
                raise NotImplementedError("posterior prediction needs an exact GP with a kernel of pgmuvi_amd.gpytorch.kernels")
            k = c.kernel
            out, _ = _evaluate(train_x, self.train_targets, marg.mean, c.noise_vec, c.noise_scalar,
                               k.mixture_weights, k.mixture_means, k.mixture_scales, k.dim_order, True)
            test_prior = self.forward(x_test)
            pm, pv = _hip.predict(out["workspace"], x_test, test_prior.mean)
        return MultivariateNormal(pm.to(x_test.dtype), DiagCovariance(pv.to(x_test.dtype)))


    def _cross_covariance(self, x_train, x_test):
        """K(x_train, x_test) of the model's kernel (``covar_module``; models with another layout override this)."""
        k = getattr(self, "covar_module", None)
        if k is None:
            raise NotImplementedError("eval-mode prediction of a dense-kernel model needs a `covar_module`")
        return k(x_train, x_test).to_dense()


class ApproximateGP(GP):
    """Importable placeholder (``pgmuvi/gps.py:21``): the variational model is unreachable
    from ``fit()`` (``pgmuvi/trainers.py:120-126``) and out of scope."""

    def __init__(self, *args, **kwargs):
        raise NotImplementedError("ApproximateGP is outside the scope of pgmuvi_amd")
