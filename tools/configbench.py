#!/usr/bin/env python3
"""Timing of the other BASELINE.json configurations on one GPU (they are parity-test cases, not bench lines):
config 3 (batch of N=2048 light curves, here 64 per launch set), config 4 (8 bands x 1024 points, 2-D SM kernel, Q=3)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pgmuvi_amd import _hip, synthetic as syn
dev = torch.device("cuda:0"); D = torch.float64

def timeit(f, reps):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

for B in (8, 64, 512):
    xs, ys, ns, ws_, mus, vs, ms = [], [], [], [], [], [], []
    for i in range(B):
        (t, y, e), per = syn.cfg3_lightcurve(i, n_obs=2048)
        h = syn.cfg_hypers(3, y.double(), lead_period=per)
        xs.append(t.double().reshape(-1, 1)); ys.append(y.double()); ns.append(e.double() ** 2)
        ws_.append(h["w"]); mus.append(h["mu"].reshape(4, 1)); vs.append(h["v"].reshape(4, 1)); ms.append(h["mean"].expand(2048))
    st = lambda L: torch.stack(L).to(dev).contiguous()
    x, y, nz, w, mu, v, m = st(xs), st(ys), st(ns), st(ws_), st(mus), st(vs), st(ms)
    dt = timeit(lambda: _hip.mll_value_grad(x, y, m, nz, None, w, mu, v, 0, 0.0, True), 5)
    print(f"config 3: {B} light curves x N=2048 per launch set: {dt*1e3:.2f} ms = {B/dt:.0f} evals/s ({B*2048**3/dt/1e12:.1f} TFLOP/s algorithmic)")
    _hip.release_workspaces()

X, Y, E = syn.cfg4()
h = syn.cfg_hypers(4, Y.double())
x, y, nz = X.double().to(dev), Y.double().to(dev), (E.double() ** 2).to(dev)
n = y.shape[0]
for order in (0, 1):
    dt = timeit(lambda: _hip.mll_value_grad(x, y, h["mean"].to(dev).expand(n), nz, None, h["w"].to(dev), h["mu"].reshape(-1, 2).to(dev), h["v"].reshape(-1, 2).to(dev), order, 0.0, True), 3)
    print(f"config 4: N={n}, d=2, Q={h['w'].numel()}, dim_order={order}: {dt*1e3:.2f} ms/eval = {1/dt:.1f} evals/s ({n**3/dt/1e12:.1f} TFLOP/s algorithmic)")

# many short light curves (the survey case: thousands of sources with ~100 epochs each), one launch set
for B, nn in ((2048, 89), (1024, 256)):
    xs, ys, ns = [], [], []
    for i in range(8):
        (t, y, e), per = syn.cfg3_lightcurve(i, n_obs=nn)
        xs.append(t.double().reshape(-1, 1)); ys.append(y.double()); ns.append(e.double() ** 2)
    rep = B // 8
    x = torch.stack(xs).repeat(rep, 1, 1).to(dev); y = torch.stack(ys).repeat(rep, 1).to(dev); nz = torch.stack(ns).repeat(rep, 1).to(dev)
    h = syn.cfg_hypers(3, ys[0])
    w = h["w"].expand(B, 4).contiguous().to(dev); mu = h["mu"].reshape(1, 4, 1).expand(B, 4, 1).contiguous().to(dev); v = h["v"].reshape(1, 4, 1).expand(B, 4, 1).contiguous().to(dev)
    m = torch.zeros(B, nn, dtype=D, device=dev)
    torch.cuda.synchronize()
    dt = timeit(lambda: _hip.mll_value_grad(x, y, m, nz, None, w, mu, v, 0, 0.0, True), 50)          # (short calls: many repetitions)
    print(f"many short light curves: {B} x N={nn} per launch set: {dt*1e3:.2f} ms = {B/dt:.0f} evals/s")
    _hip.release_workspaces()
