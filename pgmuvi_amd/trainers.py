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


def train(lightcurve=None, model=None, likelihood=None, train_x=None, train_y=None, maxiter=100, miniter=10,
          stop=None, lr=1e-4, lossfn="mll", optim="SGD", eps=1e-8, stopavg=9, progress=True, **kwargs):
    given = [model is not None, likelihood is not None, train_x is not None, train_y is not None]
    if lightcurve is not None:
        if any(given):
            print("A lightcurve object was passed to train(), but one or more of model, likelihood, train_x and "
                  "train_y were also passed. The lightcurve object will be used, and the other parameters will be ignored.")
        model, likelihood = lightcurve.model, lightcurve.likelihood
        train_x, train_y = lightcurve._xdata_transformed, lightcurve._ydata_transformed
    elif not all(given):
        raise ValueError("If a lightcurve object is not passed to train(), **all** of model, likelihood, train_x "
                         "and train_y **must** be passed to train().")

    model.train()
    likelihood.train()

    if isinstance(lossfn, str):
        if lossfn == "mll":
            lossfn = gpytorch.mlls.ExactMarginalLogLikelihood(likelihood, model)
        elif lossfn == "elbo":
            raise NotImplementedError("Currently only maximisation of the marginal log-likelihood is implemented. "
                                      "Using elbo will be implemented soon")
        else:
            raise ValueError("lossfn must be either 'mll', 'elbo', or a gpytorch, torch or pyro loss function.")
    elif isinstance(lossfn, gpytorch.mlls.marginal_log_likelihood.MarginalLogLikelihood):
        raise NotImplementedError("Currently only maximisation of the marginal log-likelihood is implemented. "
                                  "Passing arbitrary MLL objects will be implemented soon.")
    else:
        raise ValueError("lossfn must be either 'mll', 'elbo', or a gpytorch, torch or pyro loss function.")

    bad_optim = "optim must be either 'SGD', 'Adam', 'AdamW', 'NUTS', or an instance of a torch or pyro optimiser."
    if isinstance(optim, str):
        if optim == "SGD":
            optimizer = torch.optim.SGD(model.parameters(), lr=lr)
        elif optim == "Adam":
            optimizer = torch.optim.Adam(model.parameters(), lr=lr, eps=eps)
        elif optim == "AdamW":
            optimizer = torch.optim.AdamW(model.parameters(), lr=lr, eps=eps)
        elif optim == "NUTS":
            raise NotImplementedError("Optimisation with NUTS/MCMC is not yet implemented.")
        else:
            raise ValueError(bad_optim)
    elif isinstance(optim, torch.optim.Optimizer):
        optimizer = optim
    else:
        raise ValueError(bad_optim)

    results = {"loss": [], "delta_loss": []}
    if lightcurve is not None:
        for key, value in lightcurve.get_parameters().items():
            results[key] = [value.cpu().detach().numpy()]
    else:
        for name, _ in model.named_parameters():
            key = name.split(".")[1] if "raw" in name else name
            results[key] = []
            results.setdefault(name, [])

    for i in _iterate(maxiter, progress):
        optimizer.zero_grad()
        output = model(train_x)
        loss = -lossfn(output, train_y)
        loss.backward()
        optimizer.step()
        value = loss.cpu().detach().numpy()
        if i > 0:
            results["delta_loss"].append(value - results["loss"][-1])
        results["loss"].append(value)
        if lightcurve is not None:
            for key, val in lightcurve.get_parameters().items():
                results[key].append(val.cpu().detach().numpy())
        else:
            for name, param in model.named_parameters():
                results[name].append(param.cpu().detach().numpy())
        if stop and i > miniter:
            stopval = np.std(results["loss"][-stopavg:])
            if stopval < stop:
                print(f"Average change in loss over the last {stopavg} iterations was {stopval}.\n"
                      f" This is < {stop}, so we will end training here.")
                break
    return results


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
    from .gpytorch import settings
    from .gpytorch.utils.errors import NanError
    if lightcurve is not None:
        model, likelihood = lightcurve.model, lightcurve.likelihood
        train_x, train_y = lightcurve._xdata_transformed, lightcurve._ydata_transformed
    elif any(v is None for v in (model, likelihood, train_x, train_y)):
        raise ValueError("If a lightcurve object is not passed to train(), **all** of model, likelihood, train_x "
                         "and train_y **must** be passed to train().")
    if lossfn != "mll":
        raise NotImplementedError("Currently only maximisation of the marginal log-likelihood is implemented.")
    if not train_x.is_cuda:
        raise RuntimeError("train_device needs the model and data on the GPU")
    # (constraint bounds registered after the model moved to the GPU are host tensors: inside the captured iteration their
    #  transfer would be an illegal host-to-device copy)
    model.to(train_x.device); likelihood.to(train_x.device)
    model.train(); likelihood.train()
    mll = gpytorch.mlls.ExactMarginalLogLikelihood(likelihood, model)
    params = [p for p in model.parameters() if p.requires_grad]
    if optim == "SGD":
        optimizer = torch.optim.SGD(params, lr=lr)
    elif optim == "Adam":
        optimizer = torch.optim.Adam(params, lr=lr, eps=eps, capturable=True)
    elif optim == "AdamW":
        optimizer = torch.optim.AdamW(params, lr=lr, eps=eps, capturable=True)
    else:
        raise ValueError("optim must be either 'SGD', 'Adam' or 'AdamW' for the device-resident loop.")

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

        results = {"loss": [], "delta_loss": []}
        if lightcurve is not None:
            for key, value in lightcurve.get_parameters().items():
                results[key] = [value.cpu().detach().numpy()]
        else:
            for n in names:
                results[n.split(".")[1] if "raw" in n else n] = []
                results.setdefault(n, [])
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
                value = host_loss[off]
                if i > 0:
                    results["delta_loss"].append(value - results["loss"][-1])
                results["loss"].append(value)
                if stop and i > miniter and np.std(results["loss"][-stopavg:]) < stop:
                    print(f"Average change in loss over the last {stopavg} iterations was "
                          f"{np.std(results['loss'][-stopavg:])}.\\n This is < {stop}, so we will end training here.")
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
    if lightcurve is not None:
        model, likelihood = lightcurve.model, lightcurve.likelihood
        train_x, train_y = lightcurve._xdata_transformed, lightcurve._ydata_transformed
    elif any(v is None for v in (model, likelihood, train_x, train_y)):
        raise ValueError("If a lightcurve object is not passed to train(), **all** of model, likelihood, train_x "
                         "and train_y **must** be passed to train().")
    if lossfn != "mll":
        raise NotImplementedError("Currently only maximisation of the marginal log-likelihood is implemented.")
    if optim not in _hip.NativeFit.OPT:
        raise ValueError("optim must be either 'SGD', 'Adam' or 'AdamW' for the native loop.")
    fit, pieces, raw0 = _native_fit_handle(model, likelihood, train_x, train_y, maxiter, lr, optim, eps)
    results = {"loss": [], "delta_loss": []}
    names = [n_ for n_, _ in model.named_parameters()]
    if lightcurve is not None:
        for key, value in lightcurve.get_parameters().items():
            results[key] = [value.cpu().detach().numpy()]
    else:
        for n_ in names:
            results[n_.split(".")[1] if "raw" in n_ else n_] = []
            results.setdefault(n_, [])

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
                i = done + off
                if i > 0:
                    results["delta_loss"].append(value - results["loss"][-1])
                results["loss"].append(value)
                if stop and i > miniter and np.std(results["loss"][-stopavg:]) < stop:
                    print(f"Average change in loss over the last {stopavg} iterations was "
                          f"{np.std(results['loss'][-stopavg:])}.\n This is < {stop}, so we will end training here.")
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
