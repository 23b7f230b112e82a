"""Lazy covariance objects: in training mode K is never materialised in Python --
it is built tile by tile inside the fused HIP evaluation."""
from __future__ import annotations

import torch


class LazySMCovariance:
    """K_SM(x1, x2) [+ diag(noise_vec) + noise_scalar I], unevaluated."""

    def __init__(self, kernel, x1, x2, noise_vec=None, noise_scalar=None):
        self.kernel, self.x1, self.x2 = kernel, x1, x2
        self.noise_vec, self.noise_scalar = noise_vec, noise_scalar

    @property
    def is_square(self):
        return self.x2 is self.x1 or (self.x1.shape == self.x2.shape and bool(torch.equal(self.x1, self.x2)))

    @property
    def shape(self):
        return torch.Size([self.x1.shape[-2], self.x2.shape[-2]])

    def size(self, dim=None):
        return self.shape if dim is None else self.shape[dim]

    def add_noise(self, noise_vec=None, noise_scalar=None):
        nv, ns = self.noise_vec, self.noise_scalar
        if noise_vec is not None:
            nv = noise_vec if nv is None else nv + noise_vec
        if noise_scalar is not None:
            ns = noise_scalar if ns is None else ns + noise_scalar
        return LazySMCovariance(self.kernel, self.x1, self.x2, nv, ns)

    def to_dense(self):
        from .. import _hip
        k = self.kernel
        with torch.no_grad():
            same = self.is_square
            ns = 0.0 if self.noise_scalar is None else float(self.noise_scalar)
            K = _hip.sm_kernel_dense(self.x1, self.x1 if same else self.x2, k.mixture_weights, k.mixture_means,
                                     k.mixture_scales, self.noise_vec if same else None, ns if same else 0.0, k.dim_order)
        return K.to(self.x1.dtype)

    evaluate = to_dense

    def evaluate_kernel(self):
        return self

    def diagonal_values(self):
        k = self.kernel
        w = k.mixture_weights
        d = self.x1.shape[-1]
        val = (w.sum() ** d) if k.dim_order == 0 else w.sum()       # stationary: k(x,x) is constant
        out = val.expand(self.x1.shape[-2]).clone()
        if self.noise_vec is not None:
            out = out + self.noise_vec
        if self.noise_scalar is not None:
            out = out + self.noise_scalar
        return out

    def diagonal(self, *a, **k):
        return self.diagonal_values()


class DenseCovariance:
    """A covariance matrix torch has built (any kernel but the fused spectral-mixture one), with autograd history;
    noise is kept apart so that ``to_dense`` adds it to the diagonal once."""

    def __init__(self, K, square, noise_vec=None, noise_scalar=None, kernel=None, x=None, program=None):
        # ``K`` None: the matrix of ``kernel`` on ``x`` (square), evaluated by torch only if somebody asks for it; with a
        # compiled ``program`` the training-mode log-likelihood never does (fused generic-kernel path, kernels.compile_program)
        self._K, self.square = K, square
        self.noise_vec, self.noise_scalar = noise_vec, noise_scalar
        self.kernel, self.x, self.program = kernel, x, program

    @property
    def K(self):
        if self._K is None:
            self._K = self.kernel.forward(self.x, self.x)
        return self._K

    @property
    def fused(self):
        """Can the training-mode log-likelihood take the fused generic-kernel path (nothing has materialised K)?"""
        return self.program is not None and self._K is None and self.square

    @property
    def is_square(self):
        return self.square

    @property
    def shape(self):
        if self._K is None:
            n = self.x.shape[-2]
            return torch.Size([n, n])
        return self.K.shape

    def size(self, dim=None):
        return self.shape if dim is None else self.shape[dim]

    def add_noise(self, noise_vec=None, noise_scalar=None):
        nv, ns = self.noise_vec, self.noise_scalar
        if noise_vec is not None:
            nv = noise_vec if nv is None else nv + noise_vec
        if noise_scalar is not None:
            ns = noise_scalar if ns is None else ns + noise_scalar
        return DenseCovariance(self._K, self.square, nv, ns, self.kernel, self.x, self.program)

    def to_dense(self):
        K = self.K
        if self.square and (self.noise_vec is not None or self.noise_scalar is not None):
            d = torch.zeros(K.shape[-1], dtype=K.dtype, device=K.device)
            if self.noise_vec is not None:
                d = d + self.noise_vec
            if self.noise_scalar is not None:
                d = d + self.noise_scalar
            K = K + torch.diag_embed(d)
        return K

    evaluate = to_dense

    def evaluate_kernel(self):
        return self

    def diagonal_values(self):
        out = torch.diagonal(self.K, dim1=-2, dim2=-1)
        if self.noise_vec is not None:
            out = out + self.noise_vec
        if self.noise_scalar is not None:
            out = out + self.noise_scalar
        return out

    def diagonal(self, *a, **k):
        return self.diagonal_values()


class DiagCovariance:
    """Pointwise predictive variances (what ``fast_pred_var`` consumers read)."""

    def __init__(self, var):
        self.var = var

    @property
    def shape(self):
        n = self.var.shape[-1]
        return torch.Size([n, n])

    def diagonal_values(self):
        return self.var

    def to_dense(self):
        return torch.diag_embed(self.var)

    evaluate = to_dense

    def add_noise(self, noise_vec=None, noise_scalar=None):
        v = self.var
        if noise_vec is not None:
            v = v + noise_vec
        if noise_scalar is not None:
            v = v + noise_scalar
        return DiagCovariance(v)
