"""``gpytorch.Module`` surface used by pgmuvi.

Mirrors the behaviour pgmuvi relies on: ``register_constraint`` (stored under
``module._constraints["raw_x_constraint"]``, ``/root/reference/tests/
test_constraint_sets.py:95``), ``register_prior``, dotted ``initialize(**kwargs)``
(``pgmuvi/lightcurve.py:4156``), ``named_priors`` / ``named_constraints``
(``lightcurve.py:3371-3373``) and attribute lookup that falls back to properties
(``lightcurve.py:3008-3011``).  Restated from GPyTorch's published behaviour;
GPyTorch itself is not installable here (unverified against an installed copy).
"""
from __future__ import annotations

from collections import OrderedDict

import torch
from torch import nn


class Module(nn.Module):
    def __init__(self):
        super().__init__()
        self._added_loss_terms = OrderedDict()
        self._priors = OrderedDict()
        self._constraints = OrderedDict()
        self._strict_init = True

    def __call__(self, *inputs, **kwargs):
        return self.forward(*inputs, **kwargs)

    def forward(self, *inputs, **kwargs):
        raise NotImplementedError

    def register_parameter(self, name, parameter=None, param=None):
        """GPyTorch spells the keyword ``parameter`` (``pgmuvi/gps.py:1435-1462`` uses it); torch spells it ``param``."""
        return super().register_parameter(name, parameter if parameter is not None else param)

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError as e:
            try:
                return super().__getattribute__(name)
            except AttributeError:
                raise e

    # ---- constraints -----------------------------------------------------
    def register_constraint(self, param_name, constraint, replace=True):
        if param_name not in self._parameters:
            raise RuntimeError("Attempting to register constraint for nonexistent parameter.")
        cname = param_name + "_constraint"
        old = self._constraints.get(cname)
        if old is not None and not replace and hasattr(constraint, "intersect"):
            constraint = constraint.intersect(old)
        self.add_module(cname, constraint)
        self._constraints[cname] = constraint
        if constraint.initial_value is not None:
            self.initialize(**{param_name: constraint.inverse_transform(constraint.initial_value)})

    def constraint_for_parameter_name(self, param_name):
        base, module = param_name, self
        if "." in param_name:
            module, base = self._get_module_and_name(param_name)
        return module._constraints.get(base + "_constraint")

    def named_constraints(self, memo=None, prefix=""):
        for mname, module in self.named_modules(prefix=prefix):
            if isinstance(module, Module):
                for cname, c in module._constraints.items():
                    yield (mname + "." if mname else "") + cname, c

    def constraints(self):
        for _, c in self.named_constraints():
            yield c

    def named_parameters_and_constraints(self):
        for name, param in self.named_parameters():
            yield name, param, self.constraint_for_parameter_name(name)

    # ---- priors ------------------------------------------------------------
    def register_prior(self, name, prior, param_or_closure, setting_closure=None):
        if isinstance(param_or_closure, str):
            pname = param_or_closure
            if pname not in self._parameters and not hasattr(self, pname):
                raise AttributeError(
                    f"Unknown parameter {pname} for {self.__class__.__name__}. Make sure the parameter is "
                    "registered before registering a prior.")

            def closure(module, _n=pname):
                return getattr(module, _n)
            closure.param_name = pname            # (lets the native fit loop see which parameter a prior is on)

            if setting_closure is None:
                def setting_closure(module, val, _n=pname):
                    return module.initialize(**{_n: val})
        else:
            closure = param_or_closure
        self.add_module(name, prior)
        self._priors[name] = (prior, closure, setting_closure)

    def named_priors(self, memo=None, prefix=""):
        for mname, module in self.named_modules(prefix=prefix):
            if isinstance(module, Module):
                for pname, (prior, closure, inv_closure) in module._priors.items():
                    yield (mname + "." if mname else "") + pname, module, prior, closure, inv_closure

    # ---- hyper-parameters ----------------------------------------------------
    def named_hyperparameters(self):
        for name, p in self.named_parameters():
            if "variational_" not in name:
                yield name, p

    def hyperparameters(self):
        for _, p in self.named_hyperparameters():
            yield p

    def _get_module_and_name(self, name):
        head, tail = name.split(".", 1)
        if head not in self._modules:
            raise AttributeError(f"Invalid parameter name {name}. {type(self).__name__} has no module {head}")
        sub = self._modules[head]
        if "." in tail:
            if isinstance(sub, Module):
                return sub._get_module_and_name(tail)
            for part in tail.split(".")[:-1]:
                sub = getattr(sub, part)
            return sub, tail.rsplit(".", 1)[1]
        return sub, tail

    def initialize(self, **kwargs):
        for name, val in kwargs.items():
            if isinstance(val, int):
                val = float(val)
            if "." in name:
                module, base = self._get_module_and_name(name)
                module.initialize(**{base: val})
                continue
            if not hasattr(self, name):
                raise AttributeError(f"Unknown parameter {name} for {self.__class__.__name__}")
            if name not in self._parameters and name not in self._buffers:
                setattr(self, name, val)          # property setter: transforms to the raw parameter
            elif torch.is_tensor(val):
                c = self.constraint_for_parameter_name(name)
                if c is not None and c.enforced and not c.check_raw(val):
                    raise RuntimeError(
                        "Attempting to manually set a parameter value that is out of bounds of its current "
                        "constraints, {}. Most likely, you want to do the following:\n likelihood = "
                        "GaussianLikelihood(noise_constraint=gpytorch.constraints.GreaterThan(better_lower_bound))"
                        .format(c))
                target = self.__getattr__(name)
                try:
                    target.data.copy_(val.expand_as(target))
                except RuntimeError:
                    if not self._strict_init:
                        target.data = val
                    else:
                        target.data.copy_(val.view_as(target))
            elif isinstance(val, float):
                c = self.constraint_for_parameter_name(name)
                if c is not None and not c.check_raw(torch.tensor(val)):
                    raise RuntimeError(
                        "Attempting to manually set a parameter value that is out of bounds of its current "
                        "constraints, {}.".format(c))
                self.__getattr__(name).data.fill_(val)
            else:
                raise AttributeError(f"Type {type(val)} not valid for initializing parameter {name}")
            pname = name + "_prior"
            if pname in self._priors:
                prior, closure, _ = self._priors[pname]
                try:
                    prior._validate_sample(closure(self))
                except ValueError as e:
                    raise ValueError(f"Invalid input value for prior {pname}. Error:\n{e}")
        return self

    # added loss terms are not used on this path, kept for interface parity
    def added_loss_terms(self):
        return iter(())
