import sys, time, torch
sys.path.insert(0, '.')
from pgmuvi_amd import _hip, synthetic as syn
dev = torch.device('cuda:0')
def timed(B, n, reps):
    xs, ys, ns, ws_, mus, vs, ms = [], [], [], [], [], [], []
    for i in range(min(B, 8)):
        (t, y, e), per = syn.cfg3_lightcurve(i, n_obs=n)
        h = syn.cfg_hypers(3, y.double(), lead_period=per)
        xs.append(t.double().reshape(-1, 1)); ys.append(y.double()); ns.append(e.double() ** 2)
        ws_.append(h["w"]); mus.append(h["mu"].reshape(4, 1)); vs.append(h["v"].reshape(4, 1)); ms.append(h["mean"].expand(n))
    rep = (B + len(xs) - 1) // len(xs)
    st = lambda L: torch.stack(L).repeat(rep, *([1] * L[0].dim()))[:B].to(dev).contiguous()
    x, y, nz, w, mu, v, m = st(xs), st(ys), st(ns), st(ws_), st(mus), st(vs), st(ms)
    if B == 1:
        x, y, nz, w, mu, v, m = x[0], y[0], nz[0], w[0], mu[0], v[0], m[0]
    f = lambda: _hip.mll_value_grad(x, y, m, nz, None, w, mu, v, 0, 0.0, True)
    out = f(); torch.cuda.synchronize()
    best = float("inf")
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            f()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps * 1e3)
    _hip.release_workspaces()
    return best
for B, n in [(1, 4096), (1, 1024), (16, 2048), (64, 2048), (1024, 256), (2048, 89), (1, 8192)]:
    print(f"({B}, {n}): {timed(B, n, 5):.3f}", flush=True)
