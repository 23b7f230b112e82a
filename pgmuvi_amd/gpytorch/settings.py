"""``gpytorch.settings`` context managers pgmuvi enters (``pgmuvi/lightcurve.py:5870,
5966, 9607``).  The HIP path is always the dense Cholesky path -- the semantics of
``fast_computations(False, False, False)`` -- so these are accepted and recorded but
change nothing; the Cholesky jitter policy values mirror GPyTorch's defaults."""
from __future__ import annotations

import torch


class _feature_flag:
    _default = False
    _state = None

    def __init__(self, state=True):
        self.prev = None
        self.state = state

    @classmethod
    def on(cls):
        return cls._default if cls._state is None else cls._state

    @classmethod
    def off(cls):
        return not cls.on()

    def __enter__(self):
        self.prev = type(self)._state
        type(self)._state = self.state
        return self

    def __exit__(self, *a):
        type(self)._state = self.prev
        return False


class _value_context:
    _global_value = None

    def __init__(self, value):
        self.value_ = value
        self.prev = None

    @classmethod
    def value(cls, *a):
        return cls._global_value

    def __enter__(self):
        self.prev = type(self)._global_value
        type(self)._global_value = self.value_
        return self

    def __exit__(self, *a):
        type(self)._global_value = self.prev
        return False


class _dtype_value_context:
    _values = {}

    def __init__(self, float_value=None, double_value=None, half_value=None):
        self.new = {torch.float: float_value, torch.double: double_value, torch.half: half_value}
        self.prev = None

    @classmethod
    def value(cls, dtype):
        dtype = dtype.dtype if torch.is_tensor(dtype) else dtype
        return cls._values[dtype]

    def __enter__(self):
        self.prev = dict(type(self)._values)
        for k, v in self.new.items():
            if v is not None:
                type(self)._values[k] = v
        return self

    def __exit__(self, *a):
        type(self)._values = self.prev
        return False


class max_cg_iterations(_value_context):
    _global_value = 1000


class max_cholesky_size(_value_context):
    _global_value = 800


class cholesky_max_tries(_value_context):
    _global_value = 3


class cholesky_jitter(_dtype_value_context):
    _values = {torch.float: 1e-6, torch.double: 1e-8, torch.half: 1e-3}


class min_fixed_noise(_dtype_value_context):
    _values = {torch.float: 1e-4, torch.double: 1e-6, torch.half: 1e-3}


class fast_pred_var(_feature_flag):
    _default = False


class fast_pred_samples(_feature_flag):
    _default = False


class debug(_feature_flag):
    _default = True


class skip_posterior_variances(_feature_flag):
    _default = False


class fast_computations:
    """fast_computations(covar_root_decomposition, log_prob, solves)."""

    class _sub(_feature_flag):
        _default = True

    class covar_root_decomposition(_sub):
        pass

    class log_prob(_sub):
        pass

    class solves(_sub):
        pass

    def __init__(self, covar_root_decomposition=True, log_prob=True, solves=True):
        self.ctx = [self.covar_root_decomposition(covar_root_decomposition), self.log_prob(log_prob), self.solves(solves)]

    def __enter__(self):
        for c in self.ctx:
            c.__enter__()
        return self

    def __exit__(self, *a):
        for c in reversed(self.ctx):
            c.__exit__(*a)
        return False


class check_cholesky_info(_feature_flag):
    """pgmuvi_amd extension: when on (default) every MLL evaluation reads back the
    factorisation status (one host sync, which the reference's loop has anyway at
    ``trainers.py:184``) and applies GPyTorch's jitter-retry policy; when off a failed
    factorisation surfaces as a NaN loss instead."""
    _default = True


class defer_cholesky_check(_feature_flag):
    """pgmuvi_amd extension (used by ``pgmuvi_amd.trainers.train``): when on, ``mll(output, y)`` does not wait for the
    factorisation status; the caller asks for it with ``pgmuvi_amd.mll_function.take_deferred_failure()`` once it has queued
    its own host work (``loss.backward()``) behind the evaluation, and repeats the evaluation in the ordinary mode -- where
    the jitter-retry policy applies -- if it failed."""
    _default = False
