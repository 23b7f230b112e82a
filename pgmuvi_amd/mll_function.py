"""The single autograd node the whole hot path funnels into.

``mll(output, y)`` of the reference (``/root/reference/pgmuvi/trainers.py:180``)
ends here: one call of ``pgm_mll_value_grad*_f64`` computes the value and, when any
input requires grad, every gradient in the same pass (rows A1+A3+A4+A5 of SURVEY.md
section 8a); ``loss.backward()`` (``trainers.py:181``) then only scales the stored
gradients.  Constraint transforms and the mean module stay in torch autograd (A2/A7).
"""
from __future__ import annotations

import threading
import warnings

import torch

from . import _hip
from .gpytorch import settings
from .gpytorch.utils.errors import NanError, NotPSDError, NumericalWarning


def _failed(out) -> bool:
    """Did a factorisation of this evaluation fail?  The one host synchronisation of ``mll(output, y)`` (GPyTorch's
    ``psd_safe_cholesky`` has the same one): it waits for the end of the factorisation sweep only, not for the inverse /
    gradient pass behind it, so the Python work between ``mll(...)`` and ``loss.cpu()`` -- backward, optimiser step --
    overlaps the rest of the evaluation."""
    ws = out.get("workspace")
    info = out["info"]
    status = ws.factorisation_failed(info.numel(), out.get("evaluation", -1)) if hasattr(ws, "factorisation_failed") else None
    return bool((info != 0).any()) if status is None else status


class _Deferred(threading.local):
    """Evaluations whose factorisation status nobody has asked for yet (``settings.defer_cholesky_check``): per thread -- a
    second thread's training loop must not collect this one's -- and bounded, so that a caller who switches the flag on and
    never asks does not pin every evaluation's workspace and gradient buffers.  An entry that has to leave is asked for its
    status first (its own ``info``, read from the device: the workspace's stamp may belong to a later evaluation by then) and a
    failure stays behind as a sticky flag, so an LBFGS closure with a line search, or several models per iteration, cannot
    lose a failed factorisation by running more than LIMIT evaluations between two queries."""
    LIMIT = 8

    def __init__(self):
        self.items = []
        self.evicted_failure = False


_deferred = _Deferred()


def _defer(out):
    _deferred.items.append(out)
    while len(_deferred.items) > _Deferred.LIMIT:
        old = _deferred.items.pop(0)
        _deferred.evicted_failure = _failed(old) or _deferred.evicted_failure


def drop_deferred():
    """Forget the pending evaluations (a loop's ``finally``: an exception between the evaluation and the status query)."""
    _deferred.items.clear()
    _deferred.evicted_failure = False


def take_deferred_failure() -> bool:
    """Did any evaluation run under ``settings.defer_cholesky_check`` since the last call (on this thread) fail to factor?
    Waits for the factorisation sweep of those evaluations only (by the time a training loop asks -- after
    ``loss.backward()`` -- it is usually over)."""
    failed, _deferred.evicted_failure = _deferred.evicted_failure, False
    items, _deferred.items = _deferred.items, []
    for out in items:
        failed = _failed(out) or failed
    return failed


def _evaluate(x, y, mean, noise_vec, noise_scalar, w, mu, v, dim_order, need_grad):
    """Runs the HIP evaluation with GPyTorch's psd_safe_cholesky retry policy:
    jitter 0 first, then cholesky_jitter * 10**i for i < cholesky_max_tries."""
    out = _hip.mll_value_grad(x, y, mean, noise_vec, noise_scalar, w, mu, v, dim_order, 0.0, need_grad)
    if settings.check_cholesky_info.off():
        return out, 0.0
    if settings.defer_cholesky_check.on():
        _defer(out)
        return out, 0.0
    if not _failed(out):
        return out, 0.0
    if bool(torch.isnan(y).any()) or bool(torch.isnan(w).any()):
        raise NanError("cholesky: NaN in the inputs of the marginal log likelihood.")
    base = settings.cholesky_jitter.value(torch.float64)
    jitter = 0.0
    for i in range(settings.cholesky_max_tries.value()):
        jitter = base * (10 ** i)
        warnings.warn(f"A not p.d., added jitter of {jitter:.1e} to the diagonal", NumericalWarning)
        out = _hip.mll_value_grad(x, y, mean, noise_vec, noise_scalar, w, mu, v, dim_order, jitter, need_grad)
        if not _failed(out):
            return out, jitter
    raise NotPSDError(f"Matrix not positive definite after repeatedly adding jitter up to {jitter:.1e}.")


class SMExactMLLFunction(torch.autograd.Function):
    """mll per datum = log N(y | mean, K_SM(x,x) + diag(noise_vec) + noise_scalar I) / N.

    Batched when ``y`` has a leading batch dimension (all problems the same N)."""

    @staticmethod
    def forward(ctx, x, y, mean, noise_vec, noise_scalar, w, mu, v, dim_order):
        if ctx.needs_input_grad[0]:
            raise NotImplementedError("gradients with respect to the inputs x are not part of the hot path")
        need_grad = any(ctx.needs_input_grad)
        out, jitter = _evaluate(x, y, mean, noise_vec, noise_scalar, w, mu, v, dim_order, need_grad)
        ctx.jitter_used = jitter
        ctx.shapes = (y.shape, None if mean is None else mean.shape, None if noise_vec is None else noise_vec.shape,
                      None if noise_scalar is None else noise_scalar.shape, w.shape, mu.shape, v.shape)
        ctx.dtypes = (y.dtype, mean.dtype, None if noise_vec is None else noise_vec.dtype,
                      None if noise_scalar is None else noise_scalar.dtype, w.dtype, mu.dtype, v.dtype)
        ctx.offs = None
        if need_grad:
            if y.dim() == 1 and "_buf" in out:                   # one light curve: all gradients sit in one buffer
                ctx.save_for_backward(out["_buf"])
                ctx.offs = out["_offs"]
            else:
                ctx.save_for_backward(out["g_w"], out["g_mu"], out["g_v"], out["g_noise"], out["g_mean"])
        SMExactMLLFunction.last_jitter = jitter
        return out["mll"].to(w.dtype)

    @staticmethod
    def backward(ctx, gout):
        ys, ms, nvs, nss, wsh, mush, vsh = ctx.shapes
        yd, md, nvd, nsd, wd, mud, vd = ctx.dtypes
        if ctx.offs is not None:
            # one light curve: ONE multiply scales every gradient (this runs once per training iteration on the host's
            # critical path: each small torch op costs more host time than the arithmetic it launches)
            (buf,) = ctx.saved_tensors
            o = ctx.offs
            sc = buf * gout.to(torch.float64)
            need = ctx.needs_input_grad
            g_noise, g_mean = sc[o[4]:o[5]], sc[o[5]:o[6]]
            gy = (-g_mean).reshape(ys).to(yd) if need[1] else None
            gm = g_mean.reshape(ys).sum_to_size(ms).to(md) if need[2] else None
            gnv = g_noise.reshape(ys).sum_to_size(nvs).to(nvd) if (nvs is not None and need[3]) else None
            gns = g_noise.sum().reshape(nss if len(nss) else ()).to(nsd) if (nss is not None and need[4]) else None
            gw = sc[o[1]:o[2]].reshape(wsh).to(wd) if need[5] else None
            gmu = sc[o[2]:o[3]].reshape(mush).to(mud) if need[6] else None
            gv = sc[o[3]:o[4]].reshape(vsh).to(vd) if need[7] else None
            return None, gy, gm, gnv, gns, gw, gmu, gv, None
        g_w, g_mu, g_v, g_noise, g_mean = ctx.saved_tensors
        go = gout.to(torch.float64)
        gb = go.unsqueeze(-1) if go.dim() > 0 else go          # broadcast over the trailing data/param dim
        need = ctx.needs_input_grad
        gy = (-(g_mean * gb)).reshape(ys).to(yd) if need[1] else None
        gm = (g_mean * gb).reshape(ys).sum_to_size(ms).to(md) if need[2] else None
        gnv = (g_noise * gb).reshape(ys).sum_to_size(nvs).to(nvd) if (nvs is not None and need[3]) else None
        gns = None
        if nss is not None and need[4]:
            gns = (g_noise.sum(-1) * go).reshape(nss if len(nss) else ()).to(nsd)
        gw = (g_w * gb).reshape(wsh).to(wd) if need[5] else None
        gbb = gb.unsqueeze(-1) if go.dim() > 0 else go
        gmu = (g_mu * gbb).reshape(mush).to(mud) if need[6] else None
        gv = (g_v * gbb).reshape(vsh).to(vd) if need[7] else None
        return None, gy, gm, gnv, gns, gw, gmu, gv, None


SMExactMLLFunction.last_jitter = 0.0


def sm_exact_mll(x, y, mean, noise_vec, noise_scalar, w, mu, v, dim_order=0):
    return SMExactMLLFunction.apply(x, y, mean, noise_vec, noise_scalar, w, mu, v, dim_order)


# ---- dense back-end: any kernel whose matrix torch built ------------------------
def _evaluate_dense(A, r, need_grad):
    out = _hip.mll_dense(A, r, 0.0, need_grad)
    if settings.check_cholesky_info.off():
        return out, 0.0
    if not bool((out["info"] != 0).any()):
        return out, 0.0
    if bool(torch.isnan(A).any()) or bool(torch.isnan(r).any()):
        raise NanError("cholesky: NaN in the inputs of the marginal log likelihood.")
    base = settings.cholesky_jitter.value(torch.float64)
    jitter = 0.0
    for i in range(settings.cholesky_max_tries.value()):
        jitter = base * (10 ** i)
        warnings.warn(f"A not p.d., added jitter of {jitter:.1e} to the diagonal", NumericalWarning)
        out = _hip.mll_dense(A, r, jitter, need_grad)
        if not bool((out["info"] != 0).any()):
            return out, jitter
    raise NotPSDError(f"Matrix not positive definite after repeatedly adding jitter up to {jitter:.1e}.")


def _evaluate_kernel(x, y, mean, noise_vec, noise_scalar, program, theta, need_grad):
    """The fused generic-kernel evaluation with the same jitter-retry policy as ``_evaluate``."""
    out = _hip.mll_kernel_value_grad(x, y, mean, noise_vec, noise_scalar, program, theta, 0.0, need_grad)
    if settings.check_cholesky_info.off():
        return out, 0.0
    if settings.defer_cholesky_check.on():
        _defer(out)
        return out, 0.0
    if not _failed(out):
        return out, 0.0
    if bool(torch.isnan(y).any()) or bool(torch.isnan(theta).any()):
        raise NanError("cholesky: NaN in the inputs of the marginal log likelihood.")
    base = settings.cholesky_jitter.value(torch.float64)
    jitter = 0.0
    for i in range(settings.cholesky_max_tries.value()):
        jitter = base * (10 ** i)
        warnings.warn(f"A not p.d., added jitter of {jitter:.1e} to the diagonal", NumericalWarning)
        out = _hip.mll_kernel_value_grad(x, y, mean, noise_vec, noise_scalar, program, theta, jitter, need_grad)
        if not _failed(out):
            return out, jitter
    raise NotPSDError(f"Matrix not positive definite after repeatedly adding jitter up to {jitter:.1e}.")


class KernelExactMLLFunction(torch.autograd.Function):
    """mll per datum = log N(y | mean, K_theta(x, x) + diag(noise_vec) + noise_scalar I) / N for a composed stationary kernel
    given as a device program (gpytorch.kernels.compile_program): matrix build, factorisation and the contraction of
    d mll / d theta in one C-ABI call -- no N x N matrix in torch, no autograd graph over one."""

    @staticmethod
    def forward(ctx, program, x, y, mean, noise_vec, noise_scalar, theta):
        if ctx.needs_input_grad[1]:
            raise NotImplementedError("gradients with respect to the inputs x are not part of the hot path")
        need_grad = any(ctx.needs_input_grad)
        out, jitter = _evaluate_kernel(x, y, mean, noise_vec, noise_scalar, program, theta, need_grad)
        ctx.shapes = (y.shape, mean.shape, None if noise_vec is None else noise_vec.shape,
                      None if noise_scalar is None else noise_scalar.shape, theta.shape)
        ctx.dtypes = (y.dtype, mean.dtype, None if noise_vec is None else noise_vec.dtype,
                      None if noise_scalar is None else noise_scalar.dtype, theta.dtype)
        if need_grad:
            ctx.save_for_backward(out["g_theta"], out["g_noise"], out["g_mean"])
        KernelExactMLLFunction.last_jitter = jitter
        return out["mll"].to(theta.dtype)

    @staticmethod
    def backward(ctx, gout):
        g_theta, g_noise, g_mean = ctx.saved_tensors
        ys, ms, nvs, nss, ths = ctx.shapes
        yd, md, nvd, nsd, thd = ctx.dtypes
        go = gout.to(torch.float64)
        need = ctx.needs_input_grad
        gy = (-(g_mean * go)).reshape(ys).to(yd) if need[2] else None
        gm = (g_mean * go).reshape(ys).sum_to_size(ms).to(md) if need[3] else None
        gnv = (g_noise * go).reshape(ys).sum_to_size(nvs).to(nvd) if (nvs is not None and need[4]) else None
        gns = (g_noise.sum() * go).reshape(nss if len(nss) else ()).to(nsd) if (nss is not None and need[5]) else None
        gth = (g_theta * go).reshape(ths).to(thd) if need[6] else None
        return None, None, gy, gm, gnv, gns, gth


KernelExactMLLFunction.last_jitter = 0.0


def kernel_exact_mll(program, x, y, mean, noise_vec, noise_scalar, theta):
    return KernelExactMLLFunction.apply(program, x, y, mean, noise_vec, noise_scalar, theta)


class DenseExactMLLFunction(torch.autograd.Function):
    """mll per datum = log N(r | 0, A) / N for a dense symmetric A: the factorisation sweep of the hot path on a matrix
    built elsewhere; backward hands dmll/dA = (alpha alpha^T - A^-1) / 2N and dmll/dr = -alpha / N to autograd."""

    @staticmethod
    def forward(ctx, A, r):
        need_grad = any(ctx.needs_input_grad)
        out, jitter = _evaluate_dense(A, r, need_grad)
        ctx.dt = (A.dtype, r.dtype)
        if need_grad:
            ctx.save_for_backward(out["g_a"], out["g_r"])
        DenseExactMLLFunction.last_workspace = out["workspace"]
        return out["mll"].to(A.dtype)

    @staticmethod
    def backward(ctx, gout):
        g_a, g_r = ctx.saved_tensors
        go = gout.to(torch.float64)
        ga = (g_a * go.reshape(go.shape + (1, 1))).to(ctx.dt[0]) if ctx.needs_input_grad[0] else None
        gr = (g_r * go.reshape(go.shape + (1,))).to(ctx.dt[1]) if ctx.needs_input_grad[1] else None
        return ga, gr


DenseExactMLLFunction.last_workspace = None


def dense_exact_mll(A, r):
    return DenseExactMLLFunction.apply(A, r)
