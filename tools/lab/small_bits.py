"""Is the one launch (k_small) the launch sequence bit for bit in the VALUE?  Random light curves of 1 .. 128 points, 1 .. 8 mixtures, one and
two input dimensions (both orders), vector / scalar noise: PGM_SMALL=2 against PGM_SMALL=0 on two workspaces, values compared as bits,
gradients to rounding.   python tools/lab/small_bits.py [cases] [seed]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pgmuvi_amd import _hip
D = torch.float64
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
gen = torch.Generator().manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
dev = torch.device("cuda:0")
ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=gen))
os.environ["PGM_SMALL"] = "2"; wa = {}
os.environ["PGM_SMALL"] = "0"; wb = {}
def ws_of(table, sw, q, d):
    if (q, d) not in table:
        os.environ["PGM_SMALL"] = sw
        table[(q, d)] = _hip.Workspace(dev, 128, q, d, 1)
    return table[(q, d)]
diff, worst = 0, 0.0
for c in range(cases):
    n = ri(1, 128) if c % 4 else ri(97, 128)
    d = 1 + (c % 3 == 1); q = ri(1, 8 if d == 1 else 6); order = ri(0, 1) if d == 2 else 0
    x = torch.rand(n, d, generator=gen, dtype=D) * 600.0
    if d == 1: x = torch.sort(x[:, 0])[0].reshape(n, 1)
    else: x[:, 1] = torch.randint(1, 4, (n,), generator=gen).double() * 0.5
    y = torch.randn(n, generator=gen, dtype=D); nz = 0.01 + 0.05 * torch.rand(n, generator=gen, dtype=D)
    use_vec = c % 5 != 4; ns = None if (use_vec and c % 2) else 0.02 + 0.1 * torch.rand((), generator=gen, dtype=D)
    w = 0.1 + torch.rand(q, generator=gen, dtype=D); mu = 0.005 + 0.3 * torch.rand(q, d, generator=gen, dtype=D); v = 0.001 + 0.02 * torch.rand(q, d, generator=gen, dtype=D)
    mean = torch.full((n,), float(torch.randn((), generator=gen, dtype=D)) * 0.3, dtype=D)
    a = (x.to(dev), y.to(dev), mean.to(dev), nz.to(dev) if use_vec else None, None if ns is None else ns.to(dev), w.to(dev), mu.to(dev), v.to(dev), order, 0.0, True)
    oa = _hip.mll_value_grad(*a, workspace=ws_of(wa, "2", q, d)); ob = _hip.mll_value_grad(*a, workspace=ws_of(wb, "0", q, d))
    torch.cuda.synchronize()
    assert int(oa["info"]) == 0 and int(ob["info"]) == 0
    if float(oa["mll"]) != float(ob["mll"]):
        diff += 1
        print(f"case {c}: n={n} q={q} d={d} order={order} vec={use_vec} scalar={ns is not None}: {float(oa['mll'])!r} vs {float(ob['mll'])!r}")
    for k in ("g_w", "g_mu", "g_v", "g_mean", "g_noise"):
        den = float(ob[k].abs().max()) + 1e-300
        worst = max(worst, float((oa[k] - ob[k]).abs().max()) / den)
print(f"{cases} light curves: {diff} values differ in their bits; worst gradient deviation {worst:.2e} (relative to the largest entry)")
