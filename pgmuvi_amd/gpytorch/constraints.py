"""``gpytorch.constraints`` (row A7 of SURVEY.md section 8a).

Semantics: Positive: softplus(raw); GreaterThan(lb): softplus(raw)+lb;
LessThan(ub): ub - softplus(-raw); Interval(lb,ub): lb + (ub-lb) sigmoid(raw).
Bounds are mutable tensor buffers (pgmuvi rewrites them,
``/root/reference/pgmuvi/lightcurve.py:3118-3166``).
"""
from __future__ import annotations

import math

import torch
from torch.nn.functional import softplus, sigmoid  # noqa: F401

from .module import Module


def inv_softplus(x):
    return x + torch.log(-torch.expm1(-x))


def inv_sigmoid(x):
    return torch.log(x) - torch.log(1 - x)


class Interval(Module):
    def __init__(self, lower_bound, upper_bound, transform=sigmoid, inv_transform=inv_sigmoid, initial_value=None):
        lower_bound = torch.as_tensor(lower_bound).float()
        upper_bound = torch.as_tensor(upper_bound).float()
        if torch.any(torch.ge(lower_bound, upper_bound)):
            raise ValueError("Got parameter bounds with empty intervals.")
        if type(self) is Interval:
            if torch.max(upper_bound) == math.inf or torch.min(lower_bound) == -math.inf:
                raise ValueError(
                    "Cannot make an Interval directly with non-finite bounds. Use a derived class like "
                    "GreaterThan or LessThan instead.")
        super().__init__()
        self.register_buffer("lower_bound", lower_bound)
        self.register_buffer("upper_bound", upper_bound)
        self._transform = transform
        self._inv_transform = inv_transform
        self._initial_value = initial_value

    @property
    def enforced(self):
        return self._transform is not None

    @property
    def initial_value(self):
        return self._initial_value

    def check(self, tensor):
        return bool(torch.all(tensor <= self.upper_bound.to(tensor.device)) and
                    torch.all(tensor >= self.lower_bound.to(tensor.device)))

    def check_raw(self, tensor):
        return self.check(self.transform(tensor))

    def intersect(self, other):
        if self.transform != other.transform:
            raise RuntimeError("Cant intersect Interval constraints with conflicting transforms")
        return Interval(torch.max(self.lower_bound, other.lower_bound), torch.min(self.upper_bound, other.upper_bound),
                        transform=self._transform, inv_transform=self._inv_transform)

    def _b(self, like):
        return self.lower_bound.to(like.device), self.upper_bound.to(like.device)

    def transform(self, tensor):
        if not self.enforced:
            return tensor
        lb, ub = self._b(tensor)
        return self._transform(tensor) * (ub - lb) + lb

    def inverse_transform(self, transformed):
        if not self.enforced:
            return transformed
        lb, ub = self._b(transformed)
        return self._inv_transform((transformed - lb) / (ub - lb))

    def __repr__(self):
        if self.lower_bound.numel() == 1 and self.upper_bound.numel() == 1:
            return f"{type(self).__name__}({self.lower_bound.item():.3E}, {self.upper_bound.item():.3E})"
        return super().__repr__()

    def __iter__(self):
        yield self.lower_bound
        yield self.upper_bound


class GreaterThan(Interval):
    def __init__(self, lower_bound, transform=softplus, inv_transform=inv_softplus, initial_value=None):
        super().__init__(lower_bound=lower_bound, upper_bound=math.inf, transform=transform,
                         inv_transform=inv_transform, initial_value=initial_value)

    def __repr__(self):
        if self.lower_bound.numel() == 1:
            return f"{type(self).__name__}({self.lower_bound.item():.3E})"
        return super().__repr__()

    def transform(self, tensor):
        return self._transform(tensor) + self.lower_bound.to(tensor.device) if self.enforced else tensor

    def inverse_transform(self, transformed):
        return self._inv_transform(transformed - self.lower_bound.to(transformed.device)) if self.enforced else transformed


class Positive(GreaterThan):
    def __init__(self, transform=softplus, inv_transform=inv_softplus, initial_value=None):
        super().__init__(lower_bound=0.0, transform=transform, inv_transform=inv_transform, initial_value=initial_value)

    def __repr__(self):
        return type(self).__name__ + "()"

    def transform(self, tensor):
        return self._transform(tensor) if self.enforced else tensor

    def inverse_transform(self, transformed):
        return self._inv_transform(transformed) if self.enforced else transformed


class LessThan(Interval):
    def __init__(self, upper_bound, transform=softplus, inv_transform=inv_softplus, initial_value=None):
        super().__init__(lower_bound=-math.inf, upper_bound=upper_bound, transform=transform,
                         inv_transform=inv_transform, initial_value=initial_value)

    def transform(self, tensor):
        return -self._transform(-tensor) + self.upper_bound.to(tensor.device) if self.enforced else tensor

    def inverse_transform(self, transformed):
        return -self._inv_transform(-(transformed - self.upper_bound.to(transformed.device))) if self.enforced else transformed

    def __repr__(self):
        return f"{type(self).__name__}({self.upper_bound.item():.3E})"


Constraint = Interval
