"""rocprofv3 target: the reference's published workload (N=89, Q=2) through train_native, 300 iterations, and 300 plain evaluations at
N = 17 / 89 / 128 -- kernel durations of the one-launch path (k_small) under `--kernel-trace --stats`."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from pgmuvi_amd import _hip
dev = torch.device("cuda:0")
r = bench.reference_published_workload(dev, iters=300)
print({k: r[k] for k in ("train", "train_native")})
sys.path.insert(0, os.path.join(ROOT, "tools"))
D = torch.float64
for n, q in ((17, 2), (89, 2), (128, 4)):
    g = torch.Generator().manual_seed(n)
    x = torch.sort(torch.rand(n, generator=g, dtype=D) * 400)[0].reshape(n, 1).to(dev)
    y = torch.randn(n, generator=g, dtype=D).to(dev)
    nz = (0.01 + 0.05 * torch.rand(n, generator=g, dtype=D)).to(dev)
    w = (0.1 + torch.rand(q, generator=g, dtype=D)).to(dev); mu = (0.005 + 0.2 * torch.rand(q, 1, generator=g, dtype=D)).to(dev); v = (0.002 + 0.02 * torch.rand(q, 1, generator=g, dtype=D)).to(dev)
    for _ in range(300):
        _hip.mll_value_grad(x, y, torch.zeros(n, dtype=D, device=dev), nz, None, w, mu, v, 0, 0.0, True)
    torch.cuda.synchronize()
