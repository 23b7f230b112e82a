"""Ragged batches on one GPU: B light curves with N ~ U{n_lo..n_hi} through the ragged entry point --
one trimmed launch set (every member stops at its own last block row), and with PGM_RAGGED_TRIM=0 the padded launch sets of
(nearly) equal block rows, one after the other and two at a time -- against the same number of light curves padded to n_hi in
the equal-length batched call.  tools/raggedbench.py [B] [n_lo] [n_hi] [reps]"""
import os
import sys
import time

import torch

sys.path.insert(0, ".")
from pgmuvi_amd import _hip, synthetic as syn          # noqa: E402
from pgmuvi_amd.batch import default_chunk, evaluate_batch, evaluate_ragged, pad_curves, ragged_lengths   # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n_lo = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
n_hi = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
dev = torch.device("cuda:0")
lengths = ragged_lengths(B, n_lo, n_hi)
curves = []
for i, n in enumerate(lengths):
    (t, y, e), per = syn.cfg3_lightcurve(i % 512, n_obs=n)
    h = syn.cfg_hypers(3, y.double(), lead_period=per)
    curves.append(dict(x=t.double(), y=y.double(), noise=e.double() ** 2, mean=h["mean"], w=h["w"], mu=h["mu"], v=h["v"]))
padded, lens = pad_curves(curves, device=dev)
chunk = default_chunk(n_hi, device=dev)


def timed(f):
    f(); torch.cuda.synchronize()
    best = 1e30
    for _ in range(reps):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


work = sum(float(n) ** 3 for n in lengths)
rate = lambda ms: f"{ms:.2f} ms per pass = {B / ms * 1e3:.0f} evaluations/s, {work / ms * 1e-9:.1f} TFLOP/s on the light curves' own N^3 ({work / ms * 1e-9 / 78.6:.3f} of the fp64 MFMA peak)"
set_of, nbs = _hip.ragged_plan(lengths, min(B, chunk))
print(f"{B} light curves, N ~ U{{{n_lo}..{n_hi}}}: {len(nbs)} launch set(s), block rows {nbs}, members {[set_of.count(k) for k in range(len(nbs))]}")
rag = timed(lambda: evaluate_ragged(padded=padded, lengths=lens, chunk=chunk))
out = evaluate_ragged(padded=padded, lengths=lens, chunk=chunk)
torch.cuda.synchronize()
print(f"ragged entry point (trimmed set: the default): {rate(rag)}; info max {int(out['info'].abs().max())}")
_hip.release_workspaces()
os.environ["PGM_RAGGED_TRIM"] = "0"
set_of, nbs = _hip.ragged_plan(lengths, min(B, chunk))
print(f"PGM_RAGGED_TRIM=0: {len(nbs)} padded launch sets, block rows {nbs}, members {[set_of.count(k) for k in range(len(nbs))]}")
rag1 = timed(lambda: evaluate_ragged(padded=padded, lengths=lens, chunk=chunk, streams=1))
rag2 = timed(lambda: evaluate_ragged(padded=padded, lengths=lens, chunk=chunk))
out1 = evaluate_ragged(padded=padded, lengths=lens, chunk=chunk, streams=1)
torch.cuda.synchronize()
assert torch.equal(out["mll"], out1["mll"])                    # (the value does not depend on the sets)
print(f"   the sets one after the other on one stream: {rate(rag1)}")
print(f"   two sets at a time (two streams, two workspaces): {rate(rag2)}")
del os.environ["PGM_RAGGED_TRIM"]
_hip.release_workspaces()
# the same light curves padded to n_hi points each (what the equal-length call forces on a caller): the cfg-3 recipe at n_hi
xs, ys, ms, ns, ws, mus, vs = [], [], [], [], [], [], []
for i in range(min(B, 64)):
    (t, y, e), per = syn.cfg3_lightcurve(i, n_obs=n_hi)
    h = syn.cfg_hypers(3, y.double(), lead_period=per)
    xs.append(t.double().reshape(n_hi, 1)); ys.append(y.double()); ns.append(e.double() ** 2); ms.append(h["mean"].expand(n_hi))
    ws.append(h["w"]); mus.append(h["mu"].reshape(4, 1)); vs.append(h["v"].reshape(4, 1))
rep = (B + len(xs) - 1) // len(xs)
st = lambda L: torch.stack(L).repeat(rep, *([1] * L[0].dim()))[:B].to(dev).contiguous()
x, y, m, nz, w, mu, v = st(xs), st(ys), st(ms), st(ns), st(ws), st(mus), st(vs)
eq = timed(lambda: evaluate_batch(x, y, m, nz, w, mu, v, chunk=chunk))
print(f"padded to N={n_hi}, equal-length call: {eq:.2f} ms per pass = {B / eq * 1e3:.0f} evaluations/s   (ragged / padded time: {rag / eq:.3f}; "
      f"sum N^3 / B n_hi^3 = {work / (B * float(n_hi) ** 3):.3f})")
