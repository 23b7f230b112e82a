"""The ragged batch of tools/raggedbench.py, a few passes and nothing else: for rocprofv3 --kernel-trace --stats.
python3 tools/lab/ragprof.py [B] [n_lo] [n_hi] [passes]"""
import sys

import torch

sys.path.insert(0, ".")
from pgmuvi_amd import synthetic as syn          # noqa: E402
from pgmuvi_amd.batch import default_chunk, evaluate_ragged, pad_curves, ragged_lengths   # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n_lo = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
n_hi = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
passes = int(sys.argv[4]) if len(sys.argv) > 4 else 3
dev = torch.device("cuda:0")
lengths = ragged_lengths(B, n_lo, n_hi)
curves = []
for i, n in enumerate(lengths):
    (t, y, e), per = syn.cfg3_lightcurve(i % 512, n_obs=n)
    h = syn.cfg_hypers(3, y.double(), lead_period=per)
    curves.append(dict(x=t.double(), y=y.double(), noise=e.double() ** 2, mean=h["mean"], w=h["w"], mu=h["mu"], v=h["v"]))
padded, lens = pad_curves(curves, device=dev)
chunk = default_chunk(n_hi, device=dev)
for _ in range(passes):
    out = evaluate_ragged(padded=padded, lengths=lens, chunk=chunk)
torch.cuda.synchronize()
print("info max", int(out["info"].abs().max()), "mll[0]", float(out["mll"][0]))
