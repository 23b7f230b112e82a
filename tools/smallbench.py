"""A/B of the one-launch path for light curves of at most 128 points (k_small, PGM_SMALL=1, the default) against the launch
sequence of every other size (PGM_SMALL=0; 2 = the one launch whatever the shape, where the default 1 follows small_ok's measured table), in one process on one box:

    python tools/smallbench.py            (GPU box)

* the reference's published workload (N=89, Q=2, 1000 AdamW iterations; bench.py's `reference_published_workload`) through
  `train()` and `train_native`;
* evaluations per second of one light curve through the Python binding at N = 17, 89, 128 (Q = 2 and 4);
* thousands of short light curves per call (batch on gridDim.z): 4096 x N=100, Q=2.
The switch is read when a workspace is made; workspaces are released between the two settings.
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pgmuvi_amd import _hip  # noqa: E402

D = torch.float64
dev = torch.device("cuda:0")


def problem(n, q, B, seed=0):
    g = torch.Generator().manual_seed(seed + n + q)
    x = torch.sort(torch.rand(B, n, generator=g, dtype=D) * 400, dim=1)[0].unsqueeze(-1)
    y = torch.randn(B, n, generator=g, dtype=D)
    nz = 0.01 + 0.05 * torch.rand(B, n, generator=g, dtype=D)
    w = 0.1 + torch.rand(B, q, generator=g, dtype=D)
    mu = 0.005 + 0.2 * torch.rand(B, q, 1, generator=g, dtype=D)
    v = 0.002 + 0.02 * torch.rand(B, q, 1, generator=g, dtype=D)
    return [t.to(dev) for t in (x, y, torch.zeros(B, n, dtype=D), nz)] + [None] + [t.to(dev) for t in (w, mu, v)]


def rate(n, q, B, reps):
    a = problem(n, q, B)
    f = lambda: _hip.mll_value_grad(*a, 0, 0.0, True)
    out = f(); torch.cuda.synchronize()
    assert int(out["info"].abs().max()) == 0
    best = float("inf")
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            f()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps)
    return best, float(out["mll"].reshape(-1)[0])


res = {}
for sw in (os.environ.get("SMALLBENCH_ORDER", "1,0").split(",")):
    os.environ["PGM_SMALL"] = sw
    _hip.release_workspaces()
    r = {"reference_published_workload": bench.reference_published_workload(dev)}
    _hip.release_workspaces()
    for n, q in ((17, 2), (89, 2), (89, 4), (128, 4)):
        dt, val = rate(n, q, 1, 2000)
        r[f"single_n{n}_q{q}"] = {"us_per_eval_python_binding": round(dt * 1e6, 2), "mll": val}
    dt, val = rate(100, 2, 4096, 20)
    r["batch4096_n100_q2"] = {"ms_per_call": round(dt * 1e3, 3), "evals_per_s": round(4096 / dt), "mll0": val}
    _hip.release_workspaces()
    res["PGM_SMALL=" + sw] = r
os.environ.pop("PGM_SMALL", None)
print(json.dumps(res, indent=1))
