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
