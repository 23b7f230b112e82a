"""Optimiser loop around the hot path -- mirror of ``pgmuvi.trainers.train``
(``/root/reference/pgmuvi/trainers.py:12-209``): same arguments, same ``results``
dictionary (``"loss"``, ``"delta_loss"``, one list per parameter), same early-stop
rule (``np.std(loss[-stopavg:]) < stop`` after ``miniter``), same errors.  The loop
body (``trainers.py:177-182``) is the metric's unit of work: one fused HIP evaluation
per iteration.
"""
from __future__ import annotations

import numpy as np
import torch

from . import gpytorch


def _iterate(n, progress):
    if progress:
        try:
            from tqdm import tqdm
            return tqdm(range(n))
        except Exception:
            pass
    return range(n)


# ---- the pieces all three loops share (what the reference's ``train`` does around its loop body, ``trainers.py:82-209``) ----
def _problem(lightcurve, model, likelihood, train_x, train_y):
    """(model, likelihood, x, y) from a Lightcurve or from the four loose arguments (all of them, or ValueError)."""
    loose = (model, likelihood, train_x, train_y)
    if lightcurve is not None:
        if any(v is not None for v in loose):
            print("train(): a lightcurve was given together with model / likelihood / train_x / train_y; "
                  "the lightcurve is used and the loose arguments are ignored.")
        return lightcurve.model, lightcurve.likelihood, lightcurve._xdata_transformed, lightcurve._ydata_transformed
    if any(v is None for v in loose):
        raise ValueError("train() without a lightcurve needs all of model, likelihood, train_x and train_y.")
    return loose


def _mll_objective(lossfn, likelihood, model):
    """Only the exact marginal log-likelihood is trainable (as in the reference): "mll" -> the shim's
    ExactMarginalLogLikelihood; "elbo" and ready-made MLL objects -> NotImplementedError; anything else -> ValueError."""
    if lossfn == "mll":
        return gpytorch.mlls.ExactMarginalLogLikelihood(likelihood, model)
    if lossfn == "elbo" or isinstance(lossfn, gpytorch.mlls.marginal_log_likelihood.MarginalLogLikelihood):
        raise NotImplementedError("only lossfn='mll' (exact marginal log-likelihood) is implemented")
    raise ValueError("lossfn must be 'mll' ('elbo' and MLL objects are not implemented yet)")


_OPTIMISERS = {
    "SGD": lambda params, lr, eps, **kw: torch.optim.SGD(params, lr=lr),
    "Adam": lambda params, lr, eps, **kw: torch.optim.Adam(params, lr=lr, eps=eps, **kw),
    "AdamW": lambda params, lr, eps, **kw: torch.optim.AdamW(params, lr=lr, eps=eps, **kw),
}


def _optimiser(optim, params, lr, eps, instances=True, **kw):
    if isinstance(optim, str) and optim in _OPTIMISERS:
        params = list(params)
        # Parameters on the GPU: torch's fused Adam/AdamW -- the step is ONE kernel launch instead of the dozen small ones of
        # the default (foreach) form, 0.10 ms of host time per iteration of a loop that is host-bound below N ~ 2000 (N=1024:
        # 1517 -> 1781 it/s).  Same update rule, sums in another order: losses of a 200-step fit agree to 4e-14.
        if optim != "SGD" and "fused" not in kw and "capturable" not in kw and params \
                and all(p.is_cuda and torch.is_floating_point(p) for p in params):
            kw = dict(kw, fused=True)
        return _OPTIMISERS[optim](params, lr, eps, **kw)
    if optim == "NUTS":
        raise NotImplementedError("optim='NUTS': sampling is pgmuvi_amd.mcmc's job, not train()'s")
    if instances and isinstance(optim, torch.optim.Optimizer):
        return optim
    raise ValueError("optim must be 'SGD', 'Adam', 'AdamW'" + (" or a torch optimiser instance" if instances else ""))


_NP_DTYPE = {torch.float64: np.float64, torch.float32: np.float32, torch.float16: np.float16}


class _Log:
    """The ``results`` dictionary of the reference's loop: "loss", "delta_loss" and one list per parameter -- with a
    Lightcurve the keys and (constrained) values of ``get_parameters()``, starting with the values before the first step;
    without one the raw parameters under their full names (and an empty list under the reference's short key)."""

    def __init__(self, lightcurve, model):
        self.lightcurve, self.model = lightcurve, model
        self.results = {"loss": [], "delta_loss": []}
        if lightcurve is not None:
            for key, value in lightcurve.get_parameters().items():
                self.results[key] = [value.cpu().detach().numpy()]
        else:
            for name, _ in model.named_parameters():
                self.results[name.split(".")[1] if "raw" in name else name] = []
                self.results.setdefault(name, [])

    def loss(self, value):
        if self.results["loss"]:
            self.results["delta_loss"].append(value - self.results["loss"][-1])
        self.results["loss"].append(value)

    def loss_and_parameters(self, loss):
        """One iteration's entries, read from the device now (one ``.cpu()`` per tensor, as the reference's loop does,
        ``trainers.py:184-195``)."""
        self.loss(loss.cpu().detach().numpy())
        self.parameters_now()

    # -- the same through ONE asynchronous transfer per iteration (model mode on the GPU) --------------------------------
    def snapshot(self, loss, slot):
        """Queues the copy of this iteration's loss and raw parameters into pinned host buffer ``slot`` behind the
        optimiser step; returns the handle ``take`` waits on."""
        if not hasattr(self, "_named"):                      # (the walk over the module tree costs 30 us; the parameters stay the same objects)
            self._named = list(self.model.named_parameters())
        pieces = [loss.detach().reshape(1)] + [p.detach().reshape(-1) for _, p in self._named]
        flat = torch.cat([t if t.dtype is torch.float64 else t.double() for t in pieces])
        if not hasattr(self, "_pinned"):
            self._pinned = [torch.empty(flat.numel(), dtype=torch.float64).pin_memory() for _ in range(2)]
        self._pinned[slot].copy_(flat, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return ev, slot, loss.dtype

    def take(self, handle):
        """Appends the entries of a snapshot (same values, dtypes and keys as ``loss_and_parameters``)."""
        ev, slot, loss_dtype = handle
        ev.synchronize()
        host = self._pinned[slot].numpy()
        self.loss(np.asarray(host[0], dtype=_NP_DTYPE[loss_dtype]))
        off = 1
        for name, p in self._named:
            k = p.numel()
            self.results[name].append(host[off:off + k].astype(_NP_DTYPE[p.dtype]).reshape(tuple(p.shape)))
            off += k

    def restore_last(self):
        """The model goes back to the parameters of the last logged iteration."""
        with torch.no_grad():
            for name, p in self.model.named_parameters():
                p.copy_(torch.as_tensor(self.results[name][-1]).to(p.device))

    def parameters_now(self):
        if self.lightcurve is not None:
            for key, value in self.lightcurve.get_parameters().items():
                self.results[key].append(value.cpu().detach().numpy())
        else:
            for name, p in self.model.named_parameters():
                self.results[name].append(p.cpu().detach().numpy())

    def converged(self, i, miniter, stop, stopavg):
        """The reference's stop rule (``trainers.py:200-207``): after ``miniter`` iterations, the standard deviation of the
        last ``stopavg`` losses below ``stop``."""
        if not stop or i <= miniter:
            return False
        spread = np.std(self.results["loss"][-stopavg:])
        if spread < stop:
            print(f"Average change in loss over the last {stopavg} iterations was {spread}.\n"
                  f" This is < {stop}, so we will end training here.")
            return True
        return False


def train(lightcurve=None, model=None, likelihood=None, train_x=None, train_y=None, maxiter=100, miniter=10,
          stop=None, lr=1e-4, lossfn="mll", optim="SGD", eps=1e-8, stopavg=9, progress=True, **kwargs):
    """The reference's loop with its arguments, ``results`` and stop rule; the body is ``trainers.py:177-182`` verbatim in
    meaning: zero_grad, ``model(x)``, ``-mll``, backward, step -- one fused HIP evaluation per iteration -- then the
    per-iteration host read of the loss and the parameters (``:184-195``)."""
    model, likelihood, train_x, train_y = _problem(lightcurve, model, likelihood, train_x, train_y)
    model.train()
    likelihood.train()
    objective = _mll_objective(lossfn, likelihood, model)
    optimizer = _optimiser(optim, model.parameters(), lr, eps)
    from . import mll_function
    from .gpytorch import settings
    log = _Log(lightcurve, model)
    mll_function.take_deferred_failure()                     # (nothing of an earlier caller's is left pending)
    # Model mode on the GPU with an optimiser made here: the loop runs one iteration ahead of its own log.  The reference's
    # loop copies the loss and every parameter to the host at the end of each iteration (five synchronising copies); here
    # they leave in one asynchronous transfer that is read an iteration later, so the next evaluation is queued while the
    # GPU still works on this one and the GPU never waits for the host.  Results, stop iteration and final parameters are
    # the reference's: when the stop rule fires on iteration i (noticed during i + 1) the model is put back to iteration i's
    # parameters -- which the log holds -- and iteration i + 1 is dropped.
    ahead = lightcurve is None and isinstance(optim, str) and train_x.is_cuda
    pending = None
    for i in _iterate(maxiter, progress):
        optimizer.zero_grad()
        # The evaluation is launched without waiting for its factorisation status; backward is queued behind it, and only
        # then is the status asked for (by then the sweep is usually over: the host no longer idles through it, nor the
        # GPU through the host's backward).  A failed factorisation -- rare -- repeats the iteration's forward and backward
        # in the ordinary mode, where GPyTorch's jitter-retry policy (warning, NotPSDError) applies: the step is taken on
        # the same value and gradients as without the deferral.
        try:
            with settings.defer_cholesky_check(True):
                loss = -objective(model(train_x), train_y)
                # (anomaly detection wants to see the first NaN where it arises: there the status is read before backward,
                #  as in the ordinary mode, and a failed evaluation is not back-propagated at all)
                failed = mll_function.take_deferred_failure() if torch.is_anomaly_enabled() else None
                if not failed:
                    loss.backward()
        except BaseException:
            mll_function.drop_deferred()                     # (nothing of this iteration stays pinned behind the exception)
            raise
        if failed or (failed is None and mll_function.take_deferred_failure()):
            optimizer.zero_grad()
            loss = -objective(model(train_x), train_y)
            loss.backward()
        optimizer.step()
        if not ahead:
            log.loss_and_parameters(loss)
            if log.converged(i, miniter, stop, stopavg):
                break
            continue
        handle = log.snapshot(loss, i % 2)
        if pending is not None:
            log.take(pending)
            if log.converged(i - 1, miniter, stop, stopavg):
                log.restore_last()
                handle = None
                break
        pending = handle
    else:
        handle = pending
    if ahead and handle is not None:                         # the last iteration's entries (no stop before it)
        log.take(handle)
        # (its own stop test only prints: there is no later iteration to leave out)
        log.converged(len(log.results["loss"]) - 1, miniter, stop, stopavg)
    return log.results


def train_device(lightcurve=None, model=None, likelihood=None, train_x=None, train_y=None, maxiter=100, miniter=10,
                 stop=None, lr=1e-4, lossfn="mll", optim="SGD", eps=1e-8, stopavg=9, check_every=25, **kwargs):
    """Device-resident variant of :func:`train`.

    One full iteration -- constraint transforms, the fused HIP evaluation, backward, the optimiser
    step -- is recorded once as a device graph and replayed; the loss and every parameter are
    logged into device buffers each iteration and fetched in blocks of ``check_every``
    iterations, so the per-iteration host synchronisation of the reference loop
    (``pgmuvi/trainers.py:184-195``) disappears.  Same arguments and ``results`` dictionary as
    :func:`train`; differences: the stop rule is evaluated every ``check_every`` iterations (on
    exactly the same window as the reference; up to ``check_every - 1`` iterations run past it on
    the device and are discarded -- the model ends at the stop iteration's parameters), a failed
    factorisation surfaces as a non-finite loss (``NanError``, with the last good parameters
    restored) instead of the jitter retry, and ``optim`` must be one of the string choices.
    """
    from . import _hip
    from .gpytorch import settings
    from .gpytorch.utils.errors import NanError
    model, likelihood, train_x, train_y = _problem(lightcurve, model, likelihood, train_x, train_y)
    if not train_x.is_cuda:
        raise RuntimeError("train_device needs the model and data on the GPU")
    # (constraint bounds registered after the model moved to the GPU are host tensors: inside the captured iteration their
    #  transfer would be an illegal host-to-device copy)
    model.to(train_x.device); likelihood.to(train_x.device)
    model.train(); likelihood.train()
    mll = _mll_objective(lossfn, likelihood, model)
    params = [p for p in model.parameters() if p.requires_grad]
    optimizer = _optimiser(optim, params, lr, eps, instances=False, **({} if optim == "SGD" else {"capturable": True}))

    names = [n for n, _ in model.named_parameters()]
    named = dict(model.named_parameters())
    dev = train_x.device
    static_loss = torch.zeros((), dtype=train_y.dtype, device=dev)

    def iteration():
        optimizer.zero_grad(set_to_none=False)
        loss = -mll(model(train_x), train_y)
        loss.backward()
        optimizer.step()
        static_loss.copy_(loss.detach())

    with settings.check_cholesky_info(False):
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            state = {n: p.detach().clone() for n, p in named.items()}
            for _ in range(3):                    # warm-up: allocates the workspace, optimiser state, grads
                iteration()
            with torch.no_grad():                 # undo the warm-up steps
                for n, p in named.items():
                    p.copy_(state[n])
                for st in optimizer.state.values():
                    for k, v in st.items():
                        if torch.is_tensor(v):
                            v.zero_()
        torch.cuda.current_stream(dev).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            iteration()
        with torch.no_grad():                     # capture does not execute, but keep the state pristine anyway
            for n, p in named.items():
                p.copy_(state[n])

        # the captured iteration has the addresses of the workspace it ran on baked in: hold every workspace the warm-up
        # touched for as long as the graph is replayed (the cache may otherwise drop -- and thereby free -- it)
        held = _hip.cached_workspaces(dev)
        log = _Log(lightcurve, model)
        results = log.results
        loss_hist = torch.zeros(maxiter, dtype=train_y.dtype, device=dev)
        par_hist = {n: torch.zeros((maxiter,) + tuple(p.shape), dtype=p.dtype, device=dev) for n, p in named.items()}
        done, stopped = 0, False
        while done < maxiter and not stopped:
            blk = min(check_every, maxiter - done)
            for i in range(done, done + blk):
                graph.replay()
                loss_hist[i].copy_(static_loss)
                for n, p in named.items():
                    par_hist[n][i].copy_(p.detach())
            host_loss = loss_hist[done:done + blk].cpu().numpy()          # the only synchronisation of the block
            if not np.isfinite(host_loss).all():
                # the model goes back to the last parameters a finite loss produced (NaN gradients have stepped it since)
                good = done + int(np.argmax(~np.isfinite(host_loss)))
                with torch.no_grad():
                    for n, p in named.items():
                        p.copy_(par_hist[n][good - 1] if good > 0 else state[n])
                raise NanError("non-finite loss in the device-resident loop (factorisation failed or NaN parameters)")
            for off in range(blk):
                i = done + off
                log.loss(host_loss[off])
                if log.converged(i, miniter, stop, stopavg):
                    stopped = True
                    blk = off + 1
                    break
            done += blk
        n_done = len(results["loss"])
        if n_done and n_done < done:              # stopped inside a block: the model ends at the stop iteration, like the reference's
            with torch.no_grad():
                for n, p in named.items():
                    p.copy_(par_hist[n][n_done - 1])
        if lightcurve is None:
            for n in names:
                h = par_hist[n][:n_done].cpu().numpy()
                results[n].extend(list(h))
        else:
            # one entry per iteration under every key of the lightcurve's parameter view, as the reference's loop logs them
            raw_hist = {n: par_hist[n][:n_done].cpu().numpy() for n in names}
            for key, rows in _lightcurve_traces(lightcurve, raw_hist).items():
                results[key].extend(rows)
        del held
    return results


def _constraint_descriptor(module, raw_name, numel):
    """(kind, a, b) per element of a raw parameter for pgm_fit_create: GPyTorch's transforms with the bounds as the
    shim stores them (float32 tensors; an Interval's width is formed in float32, like ``Interval.transform``)."""
    from .gpytorch import constraints as C
    c = module._constraints.get(raw_name + "_constraint")
    if c is None or not c.enforced:
        return [(0, 0.0, 0.0)] * numel
    lb = c.lower_bound.detach().cpu().reshape(-1)
    ub = c.upper_bound.detach().cpu().reshape(-1)
    if lb.numel() not in (1, numel) or ub.numel() not in (1, numel):
        raise NotImplementedError("constraint bounds must be scalars or match the parameter")
    out = []
    for i in range(numel):
        l, u = lb[i % lb.numel()], ub[i % ub.numel()]
        if isinstance(c, C.LessThan):
            out.append((2, float(u), 0.0))
        elif isinstance(c, C.GreaterThan):                       # Positive is GreaterThan(0)
            out.append((1, float(l), 0.0))
        else:
            out.append((3, float(l), float(u - l)))
    return out


def _prior_descriptor(model, likelihood, pieces, noise_mod):
    """(kind, loc, scale) per entry of the native loop's raw vector from the priors registered on the model and the
    likelihood -- what ``ExactMarginalLogLikelihood`` would add (``named_priors`` of both, ``log_prob(closure(module)).sum()``).
    Plain Normal / LogNormal priors on a named parameter of the pieces only; anything else is outside the native loop."""
    from .gpytorch import priors as PR
    offsets, off = {}, 0
    for mod, name in pieces:
        numel = getattr(mod, name).numel()
        offsets[(id(mod), name[4:] if name.startswith("raw_") else name)] = (off, numel)
        off += numel
    if noise_mod is not None:                                    # ``likelihood.noise`` is the same parameter as ``noise_covar.noise``
        offsets[(id(likelihood), "noise")] = offsets[(id(noise_mod), "noise")]
    kind, loc, scale = [0] * off, [0.0] * off, [1.0] * off
    found, seen = False, set()
    for owner in (model, likelihood):
        for pname, module, prior, closure, _ in owner.named_priors():
            if id(prior) in seen:                                # (the likelihood is also a sub-module of an ExactGP)
                continue
            seen.add(id(prior))
            target = getattr(closure, "param_name", None)
            key = (id(module), target)
            if target is None or key not in offsets:
                raise NotImplementedError(f"prior {pname}: not on a parameter of the native loop; use train_device")
            if type(prior) not in (PR.NormalPrior, PR.LogNormalPrior) or getattr(prior, "_transform", None) is not None:
                raise NotImplementedError(f"prior {pname}: only plain Normal / LogNormal priors run in the native loop; use train_device")
            o, numel = offsets[key]
            shape = getattr(module, target).shape
            l = torch.as_tensor(prior.loc).detach().double().cpu().expand(shape).reshape(-1)
            s_ = torch.as_tensor(prior.scale).detach().double().cpu().expand(shape).reshape(-1)
            for i in range(numel):
                if kind[o + i] != 0:
                    raise NotImplementedError(f"prior {pname}: a second prior on the same parameter; use train_device")
                kind[o + i] = 1 if type(prior) is PR.NormalPrior else 2
                loc[o + i], scale[o + i] = float(l[i]), float(s_[i])
            found = True
    return (kind, loc, scale) if found else None


def _native_fit_handle(model, likelihood, train_x, train_y, maxiter, lr, optim, eps=1e-8):
    """The ``pgm_fit`` handle of a model (constraint / prior tables, raw start vector) and where each piece of the raw vector
    lives in the model: (fit, pieces, raw0).  Raises ``NotImplementedError`` for models outside the native loop's scope."""
    from . import _hip
    from .gpytorch import kernels, likelihoods, means
    k = getattr(model, "covar_module", None)
    mm = getattr(model, "mean_module", None)
    linear = type(mm) is means.LinearMean
    if not isinstance(k, kernels.SpectralMixtureKernel) or not (type(mm) is means.ConstantMean or (linear and mm.bias is not None)):
        raise NotImplementedError("train_native handles Constant/LinearMean + SpectralMixtureKernel models; use train_device")
    model.train(); likelihood.train()
    Q = k.num_mixtures
    if isinstance(likelihood, likelihoods.FixedNoiseGaussianLikelihood):
        if getattr(likelihood, "second_noise_covar", None) is not None:
            raise NotImplementedError("learn_additional_noise is not handled by train_native; use train_device")
        noise, noise_mod = likelihood.noise.detach(), None
    elif isinstance(likelihood, likelihoods.GaussianLikelihood):
        noise, noise_mod = None, likelihood.noise_covar
    else:
        raise NotImplementedError("train_native needs a (FixedNoise)GaussianLikelihood")
    n = train_y.shape[-1]
    if noise is not None and noise.numel() != n:
        raise NotImplementedError("the fixed noise must have one entry per training point")
    # raw vector [mean | w | mu | v | (noise)] in the order of the C ABI, and where each piece lives in the model
    pieces = ([(mm, "weights"), (mm, "bias")] if linear else [(mm, "raw_constant")]) + \
        [(k, "raw_mixture_weights"), (k, "raw_mixture_means"), (k, "raw_mixture_scales")]
    if noise_mod is not None:
        pieces.append((noise_mod, "raw_noise"))
    raw0, desc = [], []
    for mod, name in pieces:
        p = getattr(mod, name)
        raw0.extend(p.detach().double().cpu().reshape(-1).tolist())
        desc.extend(_constraint_descriptor(mod, name, p.numel()))
    prior_tab = _prior_descriptor(model, likelihood, pieces, noise_mod)
    wd = 0.01 if optim == "AdamW" else 0.0                      # torch's defaults, as trainers.py:147-151 constructs them
    fit = _hip.NativeFit(train_x, train_y, noise, Q, k.dim_order, raw0, [t[0] for t in desc], [t[1] for t in desc],
                         [t[2] for t in desc], noise_mod is not None, optim, lr, (0.9, 0.999), eps, wd, maxiter, linear_mean=linear)
    if prior_tab is not None:
        fit.set_priors(*prior_tab)
    return fit, pieces, raw0


def _lightcurve_traces(lightcurve, raw_hist):
    """Per-iteration values of every ``Lightcurve.get_parameters()`` key (``pgmuvi/lightcurve.py:8999-9077``) from the raw
    parameter histories of a device-resident loop (``raw_hist``: parameter name -> array (iterations, *shape)): without data
    transforms the constrained values of all iterations come from one vectorised constraint transform per parameter; with
    ``xtransform`` / ``ytransform`` set each iteration goes through ``get_parameters`` itself."""
    model = lightcurve.model
    named = dict(model.named_parameters())
    n_keep = len(next(iter(raw_hist.values()))) if raw_hist else 0
    traces = {}
    plain = getattr(lightcurve, "xtransform", None) is None and getattr(lightcurve, "ytransform", None) is None
    if plain:
        for pname, p in named.items():
            rows = torch.as_tensor(np.asarray(raw_hist[pname])).to(p.dtype).reshape((n_keep,) + tuple(p.shape))   # (the model's dtype, as get_parameters sees it)
            if "raw" in pname:
                comps = pname.split(".")
                key = ".".join(c.lstrip("raw_") for c in comps)                        # (the reference's own key rule)
                mod = model
                for c in comps[:-1]:
                    mod = getattr(mod, c)
                con = getattr(mod, "_constraints", {}).get(comps[-1] + "_constraint")
                vals = rows if con is None else con.transform(rows)
            else:
                key, vals = pname, rows
            arr = vals.detach().cpu().numpy()
            traces[key] = [arr[i] for i in range(n_keep)]
        return traces
    keep = {n: p.detach().clone() for n, p in named.items()}
    try:
        for i in range(n_keep):
            with torch.no_grad():
                for n, p in named.items():
                    p.copy_(torch.as_tensor(np.asarray(raw_hist[n][i]), dtype=p.dtype).reshape(p.shape).to(p.device))
            for key, value in lightcurve.get_parameters().items():
                traces.setdefault(key, []).append(value.cpu().detach().numpy())
    finally:
        with torch.no_grad():
            for n, p in named.items():
                p.copy_(keep[n])
    return traces


def train_native(lightcurve=None, model=None, likelihood=None, train_x=None, train_y=None, maxiter=100, miniter=10, stop=None,
                 lr=1e-4, lossfn="mll", optim="SGD", eps=1e-8, stopavg=9, check_every=25, **kwargs):
    """:func:`train` with the whole optimiser loop on the device (``pgm_fit_*``): constraint
    transforms, evaluation, chain rule, SGD / Adam / AdamW step and the loss / parameter log are one hipGraph replay per
    iteration; the host only reads the log every ``check_every`` iterations for the stop rule of ``pgmuvi/trainers.py:200-207``.
    For constant- or linear-mean spectral-mixture exact GPs with a fixed-noise or learned-scalar-noise Gaussian likelihood, with or
    without plain Normal / LogNormal priors on those parameters (MAP, the priors ``set_default_priors`` registers); anything
    else raises ``NotImplementedError`` (use :func:`train_device`).  Same ``results`` as :func:`train` -- one entry per
    iteration for the loss and for every parameter key, in lightcurve mode the keys and values of ``get_parameters()`` --;
    the model's raw parameters hold the final values afterwards.  A failed factorisation skips its step on the device (no
    parameter or moment is touched by NaN gradients) and surfaces as ``NanError`` at the next read of the log, with the last
    good parameters in the model."""
    from . import _hip
    from .gpytorch.utils.errors import NanError
    model, likelihood, train_x, train_y = _problem(lightcurve, model, likelihood, train_x, train_y)
    _mll_objective(lossfn, likelihood, model)                  # (argument check only: the native loop is the exact MLL)
    if optim not in _hip.NativeFit.OPT:
        _optimiser(optim, [], lr, eps, instances=False)         # raises the loop's own error for anything else
    fit, pieces, raw0 = _native_fit_handle(model, likelihood, train_x, train_y, maxiter, lr, optim, eps)
    log = _Log(lightcurve, model)
    results = log.results
    names = [n_ for n_, _ in model.named_parameters()]

    def write_back(row):
        off = 0
        for mod, name in pieces:
            p = getattr(mod, name)
            with torch.no_grad():
                p.copy_(torch.as_tensor(row[off:off + p.numel()], dtype=p.dtype).reshape(p.shape).to(p.device))
            off += p.numel()

    done, stopped = 0, False
    hist = np.zeros((0, len(raw0)))
    try:
        while done < maxiter and not stopped:
            blk = min(check_every, maxiter - done)
            fit.run(blk)
            k_done, losses, hist, raw, info = fit.read()
            new = losses[done:k_done]
            if info != 0 or not np.isfinite(new).all():
                good = int(np.argmax(~np.isfinite(losses[:k_done]))) if not np.isfinite(losses[:k_done]).all() else k_done
                write_back(hist[good - 1] if good > 0 else np.asarray(raw0))           # the last parameters a finite loss produced
                raise NanError("non-finite loss in the native loop (factorisation failed or NaN parameters)")
            for off, value in enumerate(new):
                log.loss(value)
                if log.converged(done + off, miniter, stop, stopavg):
                    stopped = True
                    break
            done = k_done
        n_keep = len(results["loss"])
        # the parameters of the last logged iteration go back into the model, the per-iteration history into the results
        write_back(hist[n_keep - 1] if n_keep > 0 else np.asarray(raw0))
        if lightcurve is None:
            off = 0
            offsets = {}
            for mod, name in pieces:
                p = getattr(mod, name)
                offsets[id(p)] = (off, p.numel(), tuple(p.shape))
                off += p.numel()
            for n_, p in model.named_parameters():
                o, cnt, shp = offsets[id(p)]
                results[n_].extend([hist[i, o:o + cnt].reshape(shp).copy() for i in range(n_keep)])
        else:
            raw_hist, off = {}, 0
            by_id = {id(p): n_ for n_, p in model.named_parameters()}
            for mod, name in pieces:
                p = getattr(mod, name)
                raw_hist[by_id[id(p)]] = hist[:n_keep, off:off + p.numel()].reshape((n_keep,) + tuple(p.shape))
                off += p.numel()
            for key, rows in _lightcurve_traces(lightcurve, raw_hist).items():
                results[key].extend(rows)
    finally:
        fit.close()
    return results
