#!/usr/bin/env python3
"""Posterior prediction (pgm_predict_f64) at N=1024 / 4096 for 500 ... 20 000 test points."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pgmuvi_amd import _hip, synthetic as syn
dev = torch.device("cuda:0"); D = torch.float64
for n in (1024, 4096):
    t, y, e = syn.cfg2(n_obs=n)
    x, yy, nz = t.double().reshape(n, 1).to(dev), y.double().to(dev), (e.double() ** 2).to(dev)
    h = syn.cfg_hypers(2, y.double())
    w, mu, v = h["w"].to(dev), h["mu"].reshape(4, 1).to(dev), h["v"].reshape(4, 1).to(dev)
    m = torch.zeros(n, dtype=D, device=dev)
    out = _hip.mll_value_grad(x, yy, m, nz, None, w, mu, v, 0, 0.0, True)
    ws = out["workspace"]
    for M in (500, 5000, 20000):
        xt = torch.linspace(float(x.min()), float(x.max()), M, dtype=D, device=dev).reshape(M, 1)
        mt = torch.zeros(M, dtype=D, device=dev)
        pm, pv = _hip.predict(ws, xt, mt); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): _hip.predict(ws, xt, mt)
        torch.cuda.synchronize()
        print(f"N={n} M={M}: predict {(time.perf_counter()-t0)/5*1e3:.2f} ms", flush=True)
    _hip.release_workspaces()
