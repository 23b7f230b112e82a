"""``gpytorch.kernels``: the spectral-mixture kernel (row A1) plus import-level stubs
for the kernels ``pgmuvi/gps.py:7-19`` imports but the hot path never evaluates."""
from __future__ import annotations

import math

import torch

from .constraints import Positive
from .lazy import LazySMCovariance
from .module import Module


class Kernel(Module):
    """Base of every kernel.  ``_dense(x1, x2)`` is the covariance matrix as a torch expression (autograd through the
    kernel's parameters); ``__call__`` wraps it for the dense back-end.  The spectral-mixture kernel overrides
    ``forward`` with its lazy object (the fused HIP path) and keeps ``_dense`` for use inside sums and products."""
    has_lengthscale = False

    def __init__(self, ard_num_dims=None, batch_shape=torch.Size(), active_dims=None, lengthscale_prior=None,
                 lengthscale_constraint=None, eps=1e-6, **kwargs):
        super().__init__()
        self.ard_num_dims = ard_num_dims
        self.batch_shape = batch_shape
        if active_dims is not None and not torch.is_tensor(active_dims):
            active_dims = torch.tensor(active_dims, dtype=torch.long)
        self.register_buffer("active_dims", active_dims)      # (pgmuvi/gps.py:1325-1330 re-registers this buffer)
        self.eps = eps
        if self.has_lengthscale:
            nd = 1 if ard_num_dims is None else ard_num_dims
            self.register_parameter("raw_lengthscale", torch.nn.Parameter(torch.zeros(*self.batch_shape, 1, nd)))
            self.register_constraint("raw_lengthscale", lengthscale_constraint or Positive())
            if lengthscale_prior is not None:
                self.register_prior("lengthscale_prior", lengthscale_prior, lambda m: m.lengthscale,
                                    lambda m, v: m._set_lengthscale(v))

    @property
    def lengthscale(self):
        return self.raw_lengthscale_constraint.transform(self.raw_lengthscale) if self.has_lengthscale else None

    @lengthscale.setter
    def lengthscale(self, value):
        self._set_lengthscale(value)

    def _set_lengthscale(self, value):
        if not self.has_lengthscale:
            raise RuntimeError("Kernel has no lengthscale.")
        self._set_raw("raw_lengthscale", value)

    def _set_raw(self, raw_name, value):
        raw = getattr(self, raw_name)
        if not torch.is_tensor(value):
            value = torch.as_tensor(value).to(raw)
        self.initialize(**{raw_name: self._constraints[raw_name + "_constraint"].inverse_transform(value.to(raw))})

    def _dense(self, x1, x2):
        raise NotImplementedError

    def _select(self, x):
        if x.ndimension() == 1:
            x = x.unsqueeze(1)
        if self.active_dims is not None:
            x = x.index_select(-1, self.active_dims.to(x.device))
        return x

    def _dense_active(self, x1, x2):
        """Matrix of this kernel on its own ``active_dims`` of the full inputs (what sums and products call)."""
        return self._dense(self._select(x1), self._select(x2))

    def forward(self, x1, x2, diag=False, **params):
        K = self._dense(x1, x2)
        return torch.diagonal(K, dim1=-2, dim2=-1) if diag else K

    def __call__(self, x1, x2=None, diag=False, **params):
        from .lazy import DenseCovariance
        square = x2 is None or x2 is x1
        if square and not diag and self.training and type(self).forward is Kernel.forward:
            # training-mode K(x, x) of a composed stationary kernel: left unevaluated with its compiled program, so that
            # ``mll(output, y)`` can take the fused generic-kernel path (pgm_mll_kernel_value_grad_f64)
            xs = x1.unsqueeze(1) if x1.ndimension() == 1 else x1
            if xs.ndimension() == 2:
                prog = compile_program(self, xs.shape[-1])
                if prog is not None:
                    # (``x`` stays the full input: the program's leaves carry the column masks of every active_dims on the way)
                    return DenseCovariance(None, True, kernel=_FullInput(self), x=xs, program=prog)
        x1 = self._select(x1)
        x2 = x1 if x2 is None else self._select(x2)
        out = self.forward(x1, x2, diag=diag, **params)
        if diag or not torch.is_tensor(out):
            return out
        return DenseCovariance(out, square or (x1.shape == x2.shape and bool(torch.equal(x1, x2))))

    def __add__(self, other):
        return AdditiveKernel(self, other)

    def __mul__(self, other):
        return ProductKernel(self, other)

    @property
    def is_stationary(self):
        return self.has_lengthscale


class _FullInput:
    """``forward(x, x)`` of a kernel applied to the FULL input (its own ``active_dims`` selected first): what a lazily held
    ``DenseCovariance`` calls if somebody asks for the matrix after all."""

    def __init__(self, kernel):
        self.kernel = kernel

    def forward(self, x1, x2):
        k = self.kernel
        return k.forward(k._select(x1), k._select(x2))


# ---- composed stationary kernels as a device program (pgm_mll_kernel_value_grad_f64, csrc/pgm_generic.inc) ----------------
KP_MAXL, KP_MAXT, KP_MAXP = 6, 4, 12
_LEAF_KINDS = {"rbf": 1, "matern0.5": 2, "matern1.5": 3, "matern2.5": 4, "periodic": 5, "rq": 6, "cosine": 7, "linear": 8,
               "constant": 9}


class KernelProgram:
    """Sum-of-products expansion of a kernel tree: ``leaves`` [(kind, dims mask, [(module, attribute), ...])], ``terms``
    [(leaf indices, [(module, attribute) of the scale factors])], ``params`` the distinct (module, attribute) pairs in theta
    order.  ``theta()`` stacks the constrained parameter values (with autograd history); ``struct`` is the C-ABI image."""

    def __init__(self, leaves, terms, params, d):
        self.leaves, self.terms, self.params, self.d = leaves, terms, params, d
        index = {(id(m), a): i for i, (m, a) in enumerate(params)}
        self.leaf_par = [index[(id(ps[0][0]), ps[0][1])] for _, _, ps in leaves]
        self.term_scales = [[index[(id(m), a)] for m, a in sc] for _, sc in terms]

    def theta(self):
        return torch.stack([getattr(m, a).reshape(()) for m, a in self.params])

    @property
    def key(self):
        return (self.d, tuple((k, dm, p) for (k, dm, _), p in zip(self.leaves, self.leaf_par)),
                tuple((tuple(lv), tuple(sc)) for (lv, _), sc in zip(self.terms, self.term_scales)), len(self.params))


def _leaf_of(k):
    """(kind, [(module, attribute)]) of a leaf kernel the device program knows, else None."""
    def single(t):
        return t is not None and t.numel() == 1
    if type(k) is RBFKernel and single(k.raw_lengthscale):
        return "rbf", [(k, "lengthscale")]
    if type(k) is MaternKernel and single(k.raw_lengthscale):
        return f"matern{k.nu}", [(k, "lengthscale")]
    if type(k) is PeriodicKernel and single(k.raw_lengthscale) and single(k.raw_period_length):
        return "periodic", [(k, "period_length"), (k, "lengthscale")]
    if type(k) is RQKernel and single(k.raw_lengthscale) and single(k.raw_alpha):
        return "rq", [(k, "lengthscale"), (k, "alpha")]
    if type(k) is CosineKernel and single(k.raw_period_length):
        return "cosine", [(k, "period_length")]
    if type(k) is LinearKernel and single(k.raw_variance):
        return "linear", [(k, "variance")]
    if type(k) is ConstantKernel and single(k.raw_constant):
        return "constant", [(k, "constant")]
    return None


def _tree_signature(k):
    """Cheap structural fingerprint of a kernel tree (modules, classes, active_dims, parameter sizes): the compiled program of
    a model is reused across iterations as long as this does not change."""
    ad = getattr(k, "active_dims", None)
    sig = [id(k), type(k).__name__, None if ad is None else tuple(int(i) for i in ad.reshape(-1).tolist()),
           tuple(p.numel() for p in k._parameters.values() if p is not None)]
    for sub in ([k.base_kernel] if hasattr(k, "base_kernel") else list(getattr(k, "kernels", []))):
        sig.append(_tree_signature(sub))
    return tuple(sig)


def compile_program(kernel, d):
    """Cached ``_compile_program``: one compilation per kernel tree and input width."""
    sig = (d, _tree_signature(kernel))
    cached = kernel.__dict__.get("_program_cache")
    if cached is not None and cached[0] == sig:
        return cached[1]
    prog = _compile_program(kernel, d)
    kernel.__dict__["_program_cache"] = (sig, prog)
    return prog


def _compile_program(kernel, d):
    """``KernelProgram`` of ``kernel`` on ``d``-column inputs, or None when the tree holds something the device program does
    not cover (then the matrix is built by torch and goes through the dense back-end): a leaf other than RBF / Matern /
    periodic / RQ / cosine / linear / constant with one lengthscale, batch shapes, more than 6 leaves, 4 product terms, 12
    parameters or 3 scale factors per term, d > 2."""
    if d < 1 or d > 2:
        return None
    leaves, leaf_ids = [], {}

    def cols_of(k, cols):
        if getattr(k, "active_dims", None) is None:
            return cols
        idx = [int(i) for i in k.active_dims.reshape(-1).tolist()]
        if any(i >= len(cols) for i in idx):
            return None
        return [cols[i] for i in idx]

    def expand(k, cols):
        if len(tuple(getattr(k, "batch_shape", ()))) != 0:
            return None
        cols = cols_of(k, cols)
        if cols is None:
            return None
        if type(k) is ScaleKernel:
            if k.raw_outputscale.numel() != 1:
                return None
            base = expand(k.base_kernel, cols)
            return None if base is None else [(lv, sc + [(k, "outputscale")]) for lv, sc in base]
        if type(k) is AdditiveKernel:
            out = []
            for sub in k.kernels:
                t = expand(sub, cols)
                if t is None:
                    return None
                out += t
            return out
        if type(k) is ProductKernel:
            out = [([], [])]
            for sub in k.kernels:
                t = expand(sub, cols)
                if t is None:
                    return None
                out = [(a + lv, b + sc) for a, b in out for lv, sc in t]
                if len(out) > KP_MAXT:
                    return None
            return out
        leaf = _leaf_of(k)
        if leaf is None:
            return None
        kind, pars = leaf
        mask = sum(1 << c for c in set(cols))
        if mask == 0 and kind != "constant":
            return None
        key = (id(k), mask)
        if key not in leaf_ids:
            leaf_ids[key] = len(leaves)
            leaves.append((_LEAF_KINDS[kind], mask, pars))
        return [([leaf_ids[key]], [])]

    terms = expand(kernel, list(range(d)))
    if not terms or len(terms) > KP_MAXT or len(leaves) > KP_MAXL:
        return None
    if any(len(set(lv)) != len(lv) or len(sc) > 3 or not lv for lv, sc in terms):      # (k * k of one module: not a product of distinct leaves)
        return None
    params, seen = [], set()
    for _, _, pars in leaves:                     # a leaf's parameters are consecutive in theta
        for m, a in pars:
            if (id(m), a) in seen:
                return None
            seen.add((id(m), a)); params.append((m, a))
    for _, sc in terms:
        for m, a in sc:
            if (id(m), a) not in seen:
                seen.add((id(m), a)); params.append((m, a))
    if len(params) > KP_MAXP:
        return None
    return KernelProgram(leaves, terms, params, d)


def _sq_dist(a, b):
    """Squared Euclidean distances (n, m) of the rows of a (n, d) and b (m, d), by differences (exact zeros on the
    diagonal, no cancellation) -- d is 1 or 2 here."""
    diff = a.unsqueeze(-2) - b.unsqueeze(-3)
    return (diff * diff).sum(-1)


class SpectralMixtureKernel(Kernel):
    r"""k(x, x') = prod_d sum_q w_q exp(-2 pi^2 (x_d v_qd - x'_d v_qd)^2) cos(2 pi (x_d mu_qd - x'_d mu_qd)).

    Constructed by pgmuvi at ``gps.py:208`` (``SMK(num_mixtures=Q)``) and ``:305``
    (``SMK(ard_num_dims=2, ...)``).  Parameters ``raw_mixture_weights (Q,)``,
    ``raw_mixture_means (Q,1,d)``, ``raw_mixture_scales (Q,1,d)`` with ``Positive``
    constraints, exposed through ``mixture_*`` properties with setters.
    ``dim_order=1`` selects the alternative reading sum_q w_q prod_d.
    """
    is_stationary = True

    def __init__(self, num_mixtures=None, ard_num_dims=1, mixture_scales_prior=None, mixture_scales_constraint=None,
                 mixture_means_prior=None, mixture_means_constraint=None, mixture_weights_prior=None,
                 mixture_weights_constraint=None, dim_order=0, **kwargs):
        if num_mixtures is None:
            raise RuntimeError("num_mixtures is a required argument")
        if mixture_means_prior is not None or mixture_scales_prior is not None or mixture_weights_prior is not None:
            raise NotImplementedError("Priors not implemented for SpectralMixtureKernel (register them on the module)")
        super().__init__(ard_num_dims=ard_num_dims, **kwargs)
        self.num_mixtures = num_mixtures
        self.dim_order = dim_order
        Q, d = num_mixtures, self.ard_num_dims
        self.register_parameter("raw_mixture_weights", torch.nn.Parameter(torch.zeros(*self.batch_shape, Q)))
        ms = torch.Size([*self.batch_shape, Q, 1, d])
        self.register_parameter("raw_mixture_means", torch.nn.Parameter(torch.zeros(ms)))
        self.register_parameter("raw_mixture_scales", torch.nn.Parameter(torch.zeros(ms)))
        self.register_constraint("raw_mixture_scales", mixture_scales_constraint or Positive())
        self.register_constraint("raw_mixture_means", mixture_means_constraint or Positive())
        self.register_constraint("raw_mixture_weights", mixture_weights_constraint or Positive())

    # ---- constrained views ------------------------------------------------
    @property
    def mixture_scales(self):
        return self.raw_mixture_scales_constraint.transform(self.raw_mixture_scales)

    @mixture_scales.setter
    def mixture_scales(self, value):
        self._set("raw_mixture_scales", value)

    @property
    def mixture_means(self):
        return self.raw_mixture_means_constraint.transform(self.raw_mixture_means)

    @mixture_means.setter
    def mixture_means(self, value):
        self._set("raw_mixture_means", value)

    @property
    def mixture_weights(self):
        return self.raw_mixture_weights_constraint.transform(self.raw_mixture_weights)

    @mixture_weights.setter
    def mixture_weights(self, value):
        self._set("raw_mixture_weights", value)

    def _set(self, raw_name, value):
        raw = getattr(self, raw_name)
        if not torch.is_tensor(value):
            value = torch.as_tensor(value).to(raw)
        self.initialize(**{raw_name: self._constraints[raw_name + "_constraint"].inverse_transform(value.to(raw))})

    # ---- data-driven initialisation (gps.py:209) ---------------------------
    def initialize_from_data(self, train_x, train_y, **kwargs):
        """Scales ~ 1/|N(0, max_dist^2)|, means ~ U(0, 0.5/min_dist), weights = std(y)/Q
        (GPyTorch's heuristic, restated from its documentation; random, unseeded)."""
        with torch.no_grad():
            if not torch.is_tensor(train_x) or not torch.is_tensor(train_y):
                raise RuntimeError("train_x and train_y should be tensors")
            if train_x.ndimension() == 1:
                train_x = train_x.unsqueeze(-1)
            xs = train_x.sort(dim=-2)[0]
            max_dist = xs[..., -1, :] - xs[..., 0, :]
            dists = xs[..., 1:, :] - xs[..., :-1, :]
            dists = torch.where(dists.eq(0.0), torch.tensor(1.0e10, dtype=train_x.dtype, device=train_x.device), dists)
            min_dist = dists.sort(dim=-2)[0][..., 0, :]
            min_dist = min_dist.unsqueeze(-2).unsqueeze(-3)
            max_dist = max_dist.unsqueeze(-2).unsqueeze(-3)
            raw = self.raw_mixture_scales
            self.mixture_scales = torch.randn_like(raw).mul_(max_dist.to(raw)).abs_().reciprocal_()
            self.mixture_means = torch.rand_like(self.raw_mixture_means).mul_(0.5).div(min_dist.to(raw))
            self.mixture_weights = train_y.std().div(self.num_mixtures)

    def _dense(self, x1, x2):
        """The same matrix as a torch expression (only used when the kernel sits inside a sum or product; on its own
        it takes the fused HIP path).  GPyTorch's evaluation order: scale, then subtract."""
        w = self.mixture_weights
        mu, v = self.mixture_means.squeeze(-2), self.mixture_scales.squeeze(-2)       # (Q, d)
        a1, a2 = x1.unsqueeze(-3) * v.unsqueeze(-2), x2.unsqueeze(-3) * v.unsqueeze(-2)     # (Q, n, d)
        c1, c2 = x1.unsqueeze(-3) * mu.unsqueeze(-2), x2.unsqueeze(-3) * mu.unsqueeze(-2)
        e = torch.exp(-2.0 * math.pi ** 2 * (a1.unsqueeze(-2) - a2.unsqueeze(-3)) ** 2)     # (Q, n, m, d)
        c = torch.cos(2.0 * math.pi * (c1.unsqueeze(-2) - c2.unsqueeze(-3)))
        if self.dim_order == 0:
            return (w.reshape(-1, 1, 1, 1) * e * c).sum(0).prod(-1)
        return (w.reshape(-1, 1, 1) * (e * c).prod(-1)).sum(0)

    def forward(self, x1, x2, diag=False, **params):
        d = x1.shape[-1]
        if d != self.ard_num_dims:
            raise RuntimeError(
                "The SpectralMixtureKernel expected the input to have {} dimensionality "
                "(based on the ard_num_dims argument). Got {}.".format(self.ard_num_dims, d))
        lazy = LazySMCovariance(self, x1, x2)
        return lazy.diagonal_values() if diag else lazy


class RBFKernel(Kernel):
    r"""exp(-1/2 |(x - x') / l|^2)."""
    has_lengthscale = True

    def _dense(self, x1, x2):
        l = self.lengthscale
        return torch.exp(-0.5 * _sq_dist(x1 / l, x2 / l))


class MaternKernel(Kernel):
    r"""Matern with nu in {1/2, 3/2, 5/2}: polynomial(sqrt(2 nu) r) exp(-sqrt(2 nu) r), r = |(x - x') / l|."""
    has_lengthscale = True

    def __init__(self, nu=2.5, **kwargs):
        if nu not in {0.5, 1.5, 2.5}:
            raise RuntimeError("nu expected to be 0.5, 1.5, or 2.5")
        super().__init__(**kwargs)
        self.nu = nu

    def _dense(self, x1, x2):
        l = self.lengthscale
        r = (_sq_dist(x1 / l, x2 / l) + 1e-30).sqrt()        # (the offset keeps d sqrt / d 0 finite; 1e-15 in r)
        e = torch.exp(-math.sqrt(2.0 * self.nu) * r)
        if self.nu == 0.5:
            return e
        if self.nu == 1.5:
            return (1.0 + math.sqrt(3.0) * r) * e
        return (1.0 + math.sqrt(5.0) * r + 5.0 / 3.0 * r * r) * e


class PeriodicKernel(Kernel):
    r"""exp(-2 sum_i sin^2(pi (x_i - x'_i) / p) / lambda)  -- lambda is GPyTorch's ``lengthscale`` (it enters
    unsquared in current GPyTorch; restated from its documentation, unverified against an installed copy)."""
    has_lengthscale = True

    def __init__(self, period_length_prior=None, period_length_constraint=None, **kwargs):
        super().__init__(**kwargs)
        nd = 1 if self.ard_num_dims is None else self.ard_num_dims
        self.register_parameter("raw_period_length", torch.nn.Parameter(torch.zeros(*self.batch_shape, 1, nd)))
        self.register_constraint("raw_period_length", period_length_constraint or Positive())
        if period_length_prior is not None:
            self.register_prior("period_length_prior", period_length_prior, lambda m: m.period_length,
                                lambda m, v: m._set_raw("raw_period_length", v))

    @property
    def period_length(self):
        return self.raw_period_length_constraint.transform(self.raw_period_length)

    @period_length.setter
    def period_length(self, value):
        self._set_raw("raw_period_length", value)

    def _dense(self, x1, x2):
        p, lam = self.period_length, self.lengthscale
        diff = (x1 * (math.pi / p)).unsqueeze(-2) - (x2 * (math.pi / p)).unsqueeze(-3)        # (n, m, d)
        return torch.exp((-2.0 * torch.sin(diff) ** 2 / lam.unsqueeze(-2)).sum(-1))


class CosineKernel(Kernel):
    r"""cos(pi |x - x'| / p)."""

    def __init__(self, period_length_prior=None, period_length_constraint=None, **kwargs):
        super().__init__(**kwargs)
        self.register_parameter("raw_period_length", torch.nn.Parameter(torch.zeros(*self.batch_shape, 1, 1)))
        self.register_constraint("raw_period_length", period_length_constraint or Positive())

    @property
    def period_length(self):
        return self.raw_period_length_constraint.transform(self.raw_period_length)

    @period_length.setter
    def period_length(self, value):
        self._set_raw("raw_period_length", value)

    def _dense(self, x1, x2):
        r = (_sq_dist(x1, x2) + 1e-30).sqrt()
        return torch.cos(math.pi * r / self.period_length.reshape(()))


class RQKernel(Kernel):
    r"""(1 + |(x - x') / l|^2 / (2 alpha))^(-alpha)."""
    has_lengthscale = True

    def __init__(self, alpha_constraint=None, **kwargs):
        super().__init__(**kwargs)
        self.register_parameter("raw_alpha", torch.nn.Parameter(torch.zeros(*self.batch_shape, 1)))
        self.register_constraint("raw_alpha", alpha_constraint or Positive())

    @property
    def alpha(self):
        return self.raw_alpha_constraint.transform(self.raw_alpha)

    @alpha.setter
    def alpha(self, value):
        self._set_raw("raw_alpha", value)

    def _dense(self, x1, x2):
        l, al = self.lengthscale, self.alpha.reshape(())
        return (1.0 + _sq_dist(x1 / l, x2 / l) / (2.0 * al)) ** (-al)


class LinearKernel(Kernel):
    r"""v x x'^T."""

    def __init__(self, variance_prior=None, variance_constraint=None, **kwargs):
        super().__init__(**kwargs)
        self.register_parameter("raw_variance", torch.nn.Parameter(torch.zeros(*self.batch_shape, 1, 1)))
        self.register_constraint("raw_variance", variance_constraint or Positive())

    @property
    def variance(self):
        return self.raw_variance_constraint.transform(self.raw_variance)

    @variance.setter
    def variance(self, value):
        self._set_raw("raw_variance", value)

    def _dense(self, x1, x2):
        return self.variance.reshape(()) * (x1 @ x2.transpose(-1, -2))


class ConstantKernel(Kernel):
    def __init__(self, constant_prior=None, constant_constraint=None, **kwargs):
        super().__init__(**kwargs)
        self.register_parameter("raw_constant", torch.nn.Parameter(torch.zeros(tuple(self.batch_shape))))
        self.register_constraint("raw_constant", constant_constraint or Positive())

    @property
    def constant(self):
        return self.raw_constant_constraint.transform(self.raw_constant)

    @constant.setter
    def constant(self, value):
        self._set_raw("raw_constant", value)

    def _dense(self, x1, x2):
        return self.constant.reshape(()) * torch.ones(x1.shape[-2], x2.shape[-2], dtype=x1.dtype, device=x1.device)


class ScaleKernel(Kernel):
    r"""outputscale * base_kernel."""

    def __init__(self, base_kernel, outputscale_prior=None, outputscale_constraint=None, **kwargs):
        if getattr(base_kernel, "active_dims", None) is not None:
            kwargs["active_dims"] = base_kernel.active_dims
        super().__init__(**kwargs)
        self.base_kernel = base_kernel
        self.register_parameter("raw_outputscale", torch.nn.Parameter(torch.zeros(tuple(self.batch_shape))))
        self.register_constraint("raw_outputscale", outputscale_constraint or Positive())
        if outputscale_prior is not None:
            self.register_prior("outputscale_prior", outputscale_prior, lambda m: m.outputscale,
                                lambda m, v: m._set_raw("raw_outputscale", v))

    @property
    def outputscale(self):
        return self.raw_outputscale_constraint.transform(self.raw_outputscale)

    @outputscale.setter
    def outputscale(self, value):
        self._set_raw("raw_outputscale", value)

    def _dense(self, x1, x2):
        # (x1, x2 already carry this kernel's active_dims; the base kernel's own selection applies on top, as in GPyTorch)
        return self.outputscale.reshape(()) * self.base_kernel._dense_active(x1, x2)


class _Composite(Kernel):
    def __init__(self, *kernels):
        super().__init__()
        self.kernels = torch.nn.ModuleList(kernels)


class AdditiveKernel(_Composite):
    def _dense(self, x1, x2):
        out = None
        for k in self.kernels:
            m = k._dense_active(x1, x2)
            out = m if out is None else out + m
        return out


class ProductKernel(_Composite):
    def _dense(self, x1, x2):
        out = None
        for k in self.kernels:
            m = k._dense_active(x1, x2)
            out = m if out is None else out * m
        return out


class _OutOfScopeKernel(Kernel):
    """Importable placeholder: the structured-kernel-interpolation models of pgmuvi/gps.py (SURVEY.md section 2 rows
    5-7) are not exact GPs on a dense matrix and stay out of scope."""

    def __init__(self, *args, **kwargs):
        raise NotImplementedError(
            f"{type(self).__name__} is outside the scope of pgmuvi_amd (exact GPs on the factorisation back-end only)")


class GridInterpolationKernel(_OutOfScopeKernel): pass
