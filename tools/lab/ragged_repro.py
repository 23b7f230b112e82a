"""Reproduces one case of tests/test_gpu_fuzz.py::test_random_ragged_batches_against_their_single_evaluations (same generator, same
draw order) and prints one member's value through every path, bits and all:  python tools/lab/ragged_repro.py SEED CASE MEMBER"""
import os, struct, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pgmuvi_amd import _hip
D = torch.float64
seed, want, member = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(seed + 1)
ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=gen))
hexd = lambda v: struct.pack(">d", float(v)).hex()
for c in range(want + 1):
    B = (ri(28, 70) if c % 3 == 1 else ri(12, 70)) if c % 4 else ri(3, 11)
    mix = c % 3
    if mix == 0:
        lengths = [ri(20, 1400) for _ in range(B)]
    elif mix == 1:
        lengths = [(640 if ri(0, 2) else ri(100, 900)) for _ in range(B)]
    else:
        lengths = [ri(20, 300) for _ in range(B)]
        lengths[ri(0, B - 1)] = 1400; lengths[ri(0, B - 1)] = 1290
    q = ri(1, 4); d = 2 if c % 3 == 1 else 1
    order = ri(0, 1) if d == 2 else 0
    S = max(lengths)
    use_vec = c % 5 != 4
    use_scalar = (not use_vec) or c % 2 == 0
    x = torch.zeros(B, S, d, dtype=D); y = torch.zeros(B, S, dtype=D); nz = torch.zeros(B, S, dtype=D)
    for b, n in enumerate(lengths):
        xb = torch.rand(n, d, generator=gen, dtype=D) * 800.0
        if d == 1:
            xb = torch.sort(xb[:, 0])[0].reshape(n, 1)
        else:
            xb[:, 1] = torch.randint(1, 4, (n,), generator=gen).double() * 0.5
        x[b, :n] = xb
        y[b, :n] = torch.randn(n, generator=gen, dtype=D)
        nz[b, :n] = 0.01 + 0.05 * torch.rand(n, generator=gen, dtype=D)
    ns = 0.02 + 0.1 * torch.rand(B, generator=gen, dtype=D)
    w = 0.1 + torch.rand(B, q, generator=gen, dtype=D)
    mu = 0.005 + 0.3 * torch.rand(B, q, d, generator=gen, dtype=D)
    v = 0.001 + 0.02 * torch.rand(B, q, d, generator=gen, dtype=D)
    mean = (torch.randn(B, 1, generator=gen, dtype=D) * 0.3).expand(B, S).contiguous()
    need_grad = c % 7 != 6
    if c < want:
        # (the test draws `pick` and, for the picked member, nothing else from this generator)
        if True:
            set_pick = ri(0, B - 1)
        continue
print(f"case {want}: B={B} q={q} d={d} order={order} use_vec={use_vec} use_scalar={use_scalar} need_grad={need_grad} lengths={lengths}")
b, n = member, lengths[member]
out = _hip.mll_value_grad_ragged(x.to(dev), y.to(dev), mean.to(dev), nz.to(dev) if use_vec else None, ns.to(dev) if use_scalar else None,
                                 lengths, w.to(dev), mu.to(dev), v.to(dev), order, 0.0, need_grad)
torch.cuda.synchronize()
print("sets", _hip.ragged_plan(lengths, B))
print(f"ragged member {b} (n={n}): mll {float(out['mll'][b])!r} {hexd(out['mll'][b])}")
args = lambda: (x[b, :n].to(dev), y[b, :n].to(dev), mean[b, :n].to(dev), nz[b, :n].to(dev) if use_vec else None,
                ns[b].to(dev) if use_scalar else None, w[b].to(dev), mu[b].to(dev), v[b].to(dev), order, 0.0, need_grad)
for sw in ("1", "2", "0"):
    os.environ["PGM_SMALL"] = sw
    _hip.release_workspaces()
    s = _hip.mll_value_grad(*args()); torch.cuda.synchronize()
    print(f"single PGM_SMALL={sw}: mll {float(s['mll'])!r} {hexd(s['mll'])}")
os.environ.pop("PGM_SMALL")
_hip.release_workspaces()
# the member inside an equal-length batch of 3 (fused sweep, batch on gridDim.z) and alone in a ragged call
rep = lambda t: t.unsqueeze(0).repeat(3, *([1] * t.dim())).contiguous()
a = args()
eq = _hip.mll_value_grad(rep(a[0]), rep(a[1]), rep(a[2]), None if a[3] is None else rep(a[3]), None if a[4] is None else a[4].reshape(1).repeat(3), rep(a[5]), rep(a[6]), rep(a[7]), order, 0.0, need_grad)
torch.cuda.synchronize()
print(f"batch of 3 equal: mll {float(eq['mll'][0])!r} {hexd(eq['mll'][0])}")
# probes: alpha / n (g_mean) through the one launch and through the launch sequence -- do the factors differ, or only the value's last sum?
res = {}
for sw in ("2", "0"):
    os.environ["PGM_SMALL"] = sw
    _hip.release_workspaces()
    s = _hip.mll_value_grad(*args()); torch.cuda.synchronize()
    res[sw] = {k: s[k].clone().cpu() for k in ("mll", "g_mean", "g_noise", "g_w")}
os.environ.pop("PGM_SMALL")
dm = (res["2"]["g_mean"] - res["0"]["g_mean"]).abs()
print(f"g_mean: {int((dm != 0).sum())} of {dm.numel()} entries differ, max |d| {float(dm.max()):.3e} (max |g_mean| {float(res['0']['g_mean'].abs().max()):.3e})")
dn = (res["2"]["g_noise"] - res["0"]["g_noise"]).abs()
print(f"g_noise: {int((dn != 0).sum())} of {dn.numel()} entries differ, max |d| {float(dn.max()):.3e}")
# z and the log det through both paths (pgm_debug_peek), and the value's arithmetic restated on the host
import ctypes, math, numpy as np
lib = ctypes.CDLL(_hip.lib_path())
lib.pgm_debug_peek.restype = ctypes.c_int
lib.pgm_debug_peek.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
peek = {}
for sw in ("2", "0"):
    os.environ["PGM_SMALL"] = sw
    _hip.release_workspaces()
    s = _hip.mll_value_grad(*args()); torch.cuda.synchronize()
    ws = s["workspace"]
    z = np.zeros(128); ld = np.zeros(1)
    assert lib.pgm_debug_peek(ws.handle, 0, 128, z.ctypes.data_as(ctypes.c_void_p), None) == 0
    assert lib.pgm_debug_peek(ws.handle, 2, 1, ld.ctypes.data_as(ctypes.c_void_p), None) == 0
    peek[sw] = (z.copy(), float(ld[0]), float(s["mll"]))
os.environ.pop("PGM_SMALL")
z2, l2, m2 = peek["2"]; z0, l0, m0 = peek["0"]
print(f"z: {int((z2 != z0).sum())} of 128 entries differ (max |d| {np.abs(z2 - z0).max():.3e}); beyond n: {z2[n:].tolist()[:4]} ... ; log det {l2!r} vs {l0!r}")
def value(z, ld):
    sq = z * z                                   # rounded products
    lo = sq[:64].copy(); lo[0] = lo[0] + ld      # thread 0 adds the log det
    def wave(vv):                                # the butterfly of wave_sum: xor 32, 16, 8, 4, 2, 1? -- order restated below both ways
        v = vv.copy()
        for off in (32, 16, 8, 4, 2, 1):
            v = v + v[np.arange(64) ^ off]
        return v[0]
    tot = 0.0 + wave(lo) + wave(sq[64:128])
    return -0.5 * (tot + n * math.log(2 * math.pi)) / n, -0.5 * math.fma(n, math.log(2 * math.pi), tot) / n if hasattr(math, "fma") else None
print("host restatement (separate multiply-add, fused):", [None if v is None else (repr(v), hexd(v)) for v in value(z0, l0)])
print("device: one launch", repr(m2), hexd(m2), " launch sequence", repr(m0), hexd(m0))
from fractions import Fraction
def fma(a, b, c): return float(Fraction(a) * Fraction(b) + Fraction(c))
def variants(z, ld):
    out = {}
    for fuse0 in (0, 1):
        for order in ((32, 16, 8, 4, 2, 1), (1, 2, 4, 8, 16, 32), (16, 8, 4, 2, 1, 32), (1, 2, 4, 8, 32, 16)):
            for fuse_tot in (0, 1):
                sq = z * z
                lo = sq[:64].copy()
                lo[0] = fma(z[0], z[0], ld) if fuse0 else lo[0] + ld
                def wave(vv):
                    v = vv.copy()
                    for off in order:
                        v = v + v[np.arange(64) ^ off]
                    return v[0]
                tot = 0.0 + wave(lo) + wave(sq[64:128])
                inner = fma(float(n), math.log(2 * math.pi), tot) if fuse_tot else tot + n * math.log(2 * math.pi)
                out[(fuse0, order, fuse_tot)] = -0.5 * inner / n
    return out
for k, v in variants(z0, l0).items():
    print(k, hexd(v), "<- one launch" if hexd(v) == hexd(m2) else ("<- launch sequence" if hexd(v) == hexd(m0) else ""))
