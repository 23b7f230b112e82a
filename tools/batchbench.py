#!/usr/bin/env python3
"""ms per call for batches of B light curves of N points (value + gradient) across the shapes where the sweep's schedule changes
(fused / windowed / panels; PGM_PANEL=0 or 4 forces one): the numbers behind the mode choice in run_sweep."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pgmuvi_amd import _hip, synthetic as syn
dev = torch.device("cuda:0")
def run(B, n):
    xs, ys, ns, ws_, mus, vs, ms = [], [], [], [], [], [], []
    for i in range(B):
        (t, y, e), per = syn.cfg3_lightcurve(i, n_obs=n)
        h = syn.cfg_hypers(3, y.double(), lead_period=per)
        xs.append(t.double().reshape(-1, 1)); ys.append(y.double()); ns.append(e.double() ** 2)
        ws_.append(h["w"]); mus.append(h["mu"].reshape(4, 1)); vs.append(h["v"].reshape(4, 1)); ms.append(h["mean"].expand(n))
    st = lambda L: torch.stack(L).to(dev).contiguous()
    x, y, nz, w, mu, v, m = st(xs), st(ys), st(ns), st(ws_), st(mus), st(vs), st(ms)
    f = lambda: _hip.mll_value_grad(x, y, m, nz, None, w, mu, v, 0, 0.0, True)
    o = f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"PANEL={os.environ.get('PGM_PANEL','auto')} B={B} N={n}: {dt*1e3:.3f} ms  mll0={float(o['mll'][0]):.12f} gw={float(o['g_w'].sum()):.10e}", flush=True)
    _hip.release_workspaces()
for B, n in ((2, 2048), (4, 2048), (8, 2048), (16, 2048), (2, 4096), (8, 4096), (8, 1024), (32, 1024), (64, 512), (256, 128), (512, 300),
             (1, 600), (1, 1500), (3, 3000), (40, 384), (25, 640), (4, 1500), (10, 1500)):
    run(B, n)
