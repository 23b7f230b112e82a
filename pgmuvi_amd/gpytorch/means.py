"""``gpytorch.means`` (row A2): the mean enters the hot path as a length-N vector."""
from __future__ import annotations

import torch

from .module import Module


class Mean(Module):
    def forward(self, x):
        raise NotImplementedError

    def __call__(self, x):
        if x.ndimension() == 1:
            x = x.unsqueeze(1)
        return super().__call__(x)


class ZeroMean(Mean):
    def forward(self, x):
        return torch.zeros(x.shape[:-1], dtype=x.dtype, device=x.device)


class ConstantMean(Mean):
    """m(x) = c; parameter ``raw_constant`` (``mean_module.raw_constant`` in pgmuvi's
    parameter dict), optional constraint/prior as in ``pgmuvi/lightcurve.py:3830-3838``."""

    def __init__(self, constant_prior=None, constant_constraint=None, batch_shape=torch.Size(), **kwargs):
        super().__init__()
        self.batch_shape = batch_shape
        self.register_parameter("raw_constant", torch.nn.Parameter(torch.zeros(batch_shape)))
        if constant_prior is not None:
            self.register_prior("mean_prior", constant_prior, self._constant_param, self._constant_closure)
        if constant_constraint is not None:
            self.register_constraint("raw_constant", constant_constraint)

    @property
    def constant(self):
        return self._constant_param(self)

    @constant.setter
    def constant(self, value):
        self._constant_closure(self, value)

    def _constant_param(self, m):
        c = m._constraints.get("raw_constant_constraint")
        return m.raw_constant if c is None else c.transform(m.raw_constant)

    def _constant_closure(self, m, value):
        if not torch.is_tensor(value):
            value = torch.as_tensor(value).to(m.raw_constant)
        c = m._constraints.get("raw_constant_constraint")
        m.initialize(raw_constant=value if c is None else c.inverse_transform(value))

    def forward(self, x):
        constant = self.constant.unsqueeze(-1)
        return constant.expand(torch.broadcast_shapes(constant.shape, x.shape[:-1]))


class LinearMean(Mean):
    def __init__(self, input_size, batch_shape=torch.Size(), bias=True):
        super().__init__()
        self.register_parameter("weights", torch.nn.Parameter(torch.randn(*batch_shape, input_size, 1)))
        if bias:
            self.register_parameter("bias", torch.nn.Parameter(torch.randn(*batch_shape, 1)))
        else:
            self.bias = None

    def forward(self, x):
        res = x.matmul(self.weights).squeeze(-1)
        if self.bias is not None:
            res = res + self.bias
        return res
