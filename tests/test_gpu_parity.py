"""GPU parity tests (run with ``-m gpu`` on the MI355X): the HIP path, called through the
C ABI (``pgmuvi_amd._hip``) and through the GPyTorch-shaped surface, against the CPU
oracle and the committed golden fixtures.

Tolerances (fp64 path).  BASELINE.json's north-star asks |d log-lik| < 1e-4 and SURVEY.md
section 8d a relative gradient error < 1e-6; the tests hold the kernels to 1e-9 on the
per-datum MLL and 1e-7 relative on gradients (fp64 round-off of an N=4096 Cholesky is
~1e-12).  Nothing here reads /root/reference.
"""
import math
import os
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import sm_mll_oracle as orc
from pgmuvi_amd import _hip, synthetic as syn
from pgmuvi_amd import gpytorch as g
from pgmuvi_amd.batch import evaluate_batch, evaluate_ragged, pad_curves

D = torch.float64
MLL_TOL = 1e-9
GRAD_RTOL = 1e-7


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("-m gpu tests need the MI355X")
    return torch.device("cuda:0")


def _load(golden_dir, name):
    return {k: v for k, v in np.load(os.path.join(golden_dir, name)).items()}


def _rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-300))


def _hip_eval(dev, x, y, mean, noise, w, mu, v, order=0, noise_scalar=None, need_grad=True):
    n = y.shape[0]
    out = _hip.mll_value_grad(x.to(dev), y.to(dev), torch.as_tensor(mean, dtype=D).expand(n).to(dev),
                              None if noise is None else noise.to(dev), noise_scalar, w.to(dev), mu.to(dev), v.to(dev),
                              order, 0.0, need_grad)
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("name", ["cfg1", "cfg2_n64", "cfg2_n200", "cfg2_n512", "cfg2_n4096",
                                  "cfg4_n256_order0", "cfg4_n256_order1", "cfg4_n8192_order0", "cfg4_n8192_order1"])
def test_golden_fixtures(dev, golden_dir, name):
    """Every committed fixture (inputs made with the reference's generator helpers,
    expectations from the oracle), including the full-size config 2 (N=4096, Q=4) and the full-size
    config 4 (8 bands x 1024 = 8192 points, d=2, Q=3, both dimension orders: value and EVERY gradient)."""
    exp = _load(golden_dir, f"expect_{name}.npz")
    tag = name.replace("_order0", "").replace("_order1", "")
    order = 1 if name.endswith("order1") else 0
    inp = _load(golden_dir, f"inputs_{tag}.npz")
    x = torch.as_tensor(inp["x"], dtype=D); y = torch.as_tensor(inp["y"], dtype=D)
    noise = torch.as_tensor(inp["yerr"], dtype=D) ** 2
    k = 0
    while f"mll_{k}" in exp:
        w, mu, v = (torch.as_tensor(exp[f"{p}_{k}"]) for p in ("w", "mu", "v"))
        out = _hip_eval(dev, x, y, torch.as_tensor(exp[f"meanc_{k}"]), noise, w, mu, v, order)
        assert int(out["info"]) == 0
        assert abs(float(out["mll"]) - float(exp[f"mll_{k}"])) < MLL_TOL
        for p in ("w", "mu", "v", "noise", "mean"):
            assert _rel(out[f"g_{p}"].reshape(-1), torch.as_tensor(exp[f"g_{p}_{k}"]).reshape(-1)) < GRAD_RTOL, (p, k)
        k += 1


# (1500, 2900: ragged sizes in the fused sweep, where update tiles ride in all three launch kinds of the chain)
# (17 ... 100: every count of 16-row sub-block steps the last diagonal block of a light curve can have, 1 ... 8)
@pytest.mark.parametrize("n", [1, 2, 3, 17, 40, 50, 70, 100, 127, 128, 129, 255, 257, 383, 500, 640, 900, 1100, 1300, 1500, 2900])
def test_ragged_sizes_vs_oracle(dev, n):
    gen = torch.Generator().manual_seed(n)
    x = torch.sort(torch.rand(n, generator=gen, dtype=D) * 500)[0]
    y = torch.randn(n, generator=gen, dtype=D)
    noise = 0.01 + 0.05 * torch.rand(n, generator=gen, dtype=D)
    w = torch.tensor([0.6, 0.3], dtype=D); mu = torch.tensor([[0.02], [0.11]], dtype=D); v = torch.tensor([[0.003], [0.01]], dtype=D)
    val, gr = orc.mll_value_grad_closed_form(x, y, 0.2, noise, w, mu, v)
    out = _hip_eval(dev, x, y, 0.2, noise, w, mu, v)
    assert abs(float(out["mll"]) - float(val)) < MLL_TOL
    for p in ("w", "mu", "v", "noise", "mean"):
        assert _rel(out[f"g_{p}"].reshape(-1), gr[p].reshape(-1)) < GRAD_RTOL, p
    out0 = _hip_eval(dev, x, y, 0.2, noise, w, mu, v, need_grad=False)
    assert float(out0["mll"]) == float(out["mll"])            # value-only path is the same arithmetic


def test_max_mixtures_and_scalar_noise(dev):
    gen = torch.Generator().manual_seed(77)
    n, Q = 300, 16
    x = torch.sort(torch.rand(n, generator=gen, dtype=D) * 100)[0]
    y = torch.randn(n, generator=gen, dtype=D)
    w = 0.05 + 0.1 * torch.rand(Q, generator=gen, dtype=D)
    mu = (0.01 + 0.5 * torch.rand(Q, 1, generator=gen, dtype=D))
    v = (0.005 + 0.05 * torch.rand(Q, 1, generator=gen, dtype=D))
    ns = torch.tensor(0.07, dtype=D)
    val, gr = orc.mll_value_grad_closed_form(x, y, 0.0, ns, w, mu, v)
    out = _hip_eval(dev, x, y, 0.0, None, w, mu, v, noise_scalar=ns.to(dev))
    assert abs(float(out["mll"]) - float(val)) < MLL_TOL
    assert _rel(out["g_noise"].sum().reshape(1), gr["noise"].reshape(1)) < GRAD_RTOL
    for p in ("w", "mu", "v"):
        assert _rel(out[f"g_{p}"].reshape(-1), gr[p].reshape(-1)) < GRAD_RTOL, p
    with pytest.raises(RuntimeError):      # Q*d beyond pgm_max_qd() is a bad argument, not a crash
        _hip_eval(dev, x, y, 0.0, None, torch.ones(17, dtype=D), torch.ones(17, 1, dtype=D), torch.ones(17, 1, dtype=D),
                  noise_scalar=ns.to(dev))


def test_full_size_properties(dev):
    """Config-2 size (N=4096, Q=4): size-independent properties of the MLL.
    (a) permutation invariance: reordering the observations leaves the value and the
    hyper-parameter gradient unchanged and permutes g_noise / g_mean;
    (b) additivity over independent blocks: with two far-separated copies of the data and
    a kernel that has decayed to exactly 0 between them, N*mll = sum of the halves."""
    t, y, e = syn.cfg2(n_obs=4096)
    x, y, noise = t.double(), y.double(), e.double() ** 2
    h = syn.cfg_hypers(2, y)
    w, mu, v = h["w"], h["mu"].reshape(4, 1), h["v"].reshape(4, 1)
    base = _hip_eval(dev, x, y, h["mean"], noise, w, mu, v)
    perm = torch.randperm(4096, generator=torch.Generator().manual_seed(0))
    pout = _hip_eval(dev, x[perm], y[perm], h["mean"], noise[perm], w, mu, v)
    assert abs(float(base["mll"]) - float(pout["mll"])) < 1e-10
    for p in ("w", "mu", "v"):
        assert _rel(pout[f"g_{p}"], base[f"g_{p}"]) < 1e-8
    assert _rel(pout["g_noise"], base["g_noise"][perm.to(dev)]) < 1e-8
    assert _rel(pout["g_mean"], base["g_mean"][perm.to(dev)]) < 1e-8
    # (b) two halves 1e7 days apart: exp(-2 pi^2 v^2 tau^2) underflows to exactly 0
    h1 = _hip_eval(dev, x[:2048], y[:2048], h["mean"], noise[:2048], w, mu, v)
    h2 = _hip_eval(dev, x[2048:], y[2048:], h["mean"], noise[2048:], w, mu, v)
    xs = torch.cat([x[:2048], x[2048:] + 1e7])
    both = _hip_eval(dev, xs, y, h["mean"], noise, w, mu, v)
    assert abs(4096 * float(both["mll"]) - 2048 * (float(h1["mll"]) + float(h2["mll"]))) < 1e-7


def test_batched_equals_singles_and_oracle(dev):
    B, n = 5, 384
    xs, ys, ns, ws, mus, vs, means = [], [], [], [], [], [], []
    for i in range(B):
        (t, y, e), per = syn.cfg3_lightcurve(i, n_obs=n)
        h = syn.cfg_hypers(3, y.double(), lead_period=per)
        xs.append(t.double().reshape(n, 1)); ys.append(y.double()); ns.append(e.double() ** 2)
        ws.append(h["w"]); mus.append(h["mu"].reshape(4, 1)); vs.append(h["v"].reshape(4, 1)); means.append(h["mean"].expand(n))
    st = lambda L: torch.stack(L).to(dev)
    out = evaluate_batch(st(xs), st(ys), st(means), st(ns), st(ws), st(mus), st(vs))
    out2 = evaluate_batch(st(xs), st(ys), st(means), st(ns), st(ws), st(mus), st(vs), chunk=2)
    torch.cuda.synchronize()
    # (the value does not depend on how many light curves share the call; the gradient sums do, in their rounding: a call with
    #  a handful of work items forms them per sixteenth tile and k-block, one with five light curves per quarter tile)
    assert torch.equal(out["mll"], out2["mll"]) and _rel(out["g_mu"].reshape(-1), out2["g_mu"].reshape(-1)) < 1e-10
    for i in range(B):
        single = _hip_eval(dev, xs[i], ys[i], means[i], ns[i], ws[i], mus[i], vs[i])
        assert float(single["mll"]) == float(out["mll"][i])          # bitwise: same kernels, batch on gridDim.z
        assert _rel(single["g_w"], out["g_w"][i]) < 1e-10
        val, gr = orc.mll_value_grad_closed_form(xs[i], ys[i], means[i], ns[i], ws[i], mus[i], vs[i])
        assert abs(float(val) - float(out["mll"][i])) < MLL_TOL
        assert _rel(out["g_v"][i].reshape(-1), gr["v"].reshape(-1)) < GRAD_RTOL
    # a handful of longer curves: the batch takes the fused sweep too (update tiles of all problems as fillers of the
    # diagonal-block launches); a single curve of this length also starts its inverse pass inside the sweep, so its
    # gradient sums are split differently: same factor (value bit for bit), gradients to rounding
    B, n = 3, 1100
    xs, ys, ns, ws, mus, vs, means = [], [], [], [], [], [], []
    for i in range(B):
        (t, y, e), per = syn.cfg3_lightcurve(10 + i, n_obs=n)
        h = syn.cfg_hypers(3, y.double(), lead_period=per)
        xs.append(t.double().reshape(n, 1)); ys.append(y.double()); ns.append(e.double() ** 2)
        ws.append(h["w"]); mus.append(h["mu"].reshape(4, 1)); vs.append(h["v"].reshape(4, 1)); means.append(h["mean"].expand(n))
    out = evaluate_batch(st(xs), st(ys), st(means), st(ns), st(ws), st(mus), st(vs))
    torch.cuda.synchronize()
    for i in range(B):
        single = _hip_eval(dev, xs[i], ys[i], means[i], ns[i], ws[i], mus[i], vs[i])
        assert float(single["mll"]) == float(out["mll"][i])
        for p in ("w", "mu", "v", "noise", "mean"):
            assert _rel(single[f"g_{p}"].reshape(-1), out[f"g_{p}"][i].reshape(-1)) < 1e-11, p
        val, gr = orc.mll_value_grad_closed_form(xs[i], ys[i], means[i], ns[i], ws[i], mus[i], vs[i])
        assert abs(float(val) - float(out["mll"][i])) < MLL_TOL
        for p in ("w", "mu", "v"):
            assert _rel(out[f"g_{p}"][i].reshape(-1), gr[p].reshape(-1)) < GRAD_RTOL, p


def _ragged_curves(lengths, first=0):
    curves = []
    for k, n in enumerate(lengths):
        (t, y, e), per = syn.cfg3_lightcurve(first + k, n_obs=n)
        h = syn.cfg_hypers(3, y.double(), lead_period=per)
        curves.append(dict(x=t.double(), y=y.double(), noise=e.double() ** 2, mean=h["mean"], w=h["w"], mu=h["mu"].reshape(4, 1),
                           v=h["v"].reshape(4, 1)))
    return curves


def test_ragged_batch_equals_singles_and_oracle(dev):
    """SURVEY.md section 8e, ragged N: 24 light curves with N in [200, 2300] in ONE call of the ragged entry point -- sorted
    by block rows, run in launch sets that share a chain length, shorter members padded with identity blocks (whole block
    rows of them where lengths joined a longer set).  Every value equals the light curve's own single evaluation bit for
    bit (the padding adds exact zeros to the sums of the value), every gradient equals it to rounding (a set's
    inverse/gradient pass is split for the set's block rows) and the oracle at the suite's tolerances."""
    rng = np.random.default_rng(11)
    lengths = [int(v) for v in rng.integers(200, 2301, size=24)]
    lengths[3], lengths[17] = 2300, 200                                  # (the ends of the range are in)
    curves = _ragged_curves(lengths)
    set_of, nbs = _hip.ragged_plan(lengths, 24)
    assert len(nbs) < len(set((n + 127) // 128 for n in lengths))        # (some lengths joined a longer set: padded block rows)
    out = evaluate_ragged(curves, device=dev)
    torch.cuda.synchronize()
    assert int(out["info"].abs().max()) == 0
    for b, (c, n) in enumerate(zip(curves, lengths)):
        single = _hip_eval(dev, c["x"].reshape(n, 1), c["y"], c["mean"], c["noise"], c["w"], c["mu"], c["v"])
        assert float(single["mll"]) == float(out["mll"][b]), (b, n)      # bit for bit
        for p in ("w", "mu", "v"):
            assert _rel(single[f"g_{p}"].reshape(-1), out[f"g_{p}"][b].reshape(-1)) < 1e-10, (b, n, p)
        assert out["g_noise"][b].shape == (n,) and _rel(single["g_noise"], out["g_noise"][b]) < 1e-10
        assert _rel(single["g_mean"], out["g_mean"][b]) < 1e-10
        val, gr = orc.mll_value_grad_closed_form(c["x"].reshape(n, 1), c["y"], c["mean"].expand(n), c["noise"], c["w"], c["mu"], c["v"])
        assert abs(float(val) - float(out["mll"][b])) < MLL_TOL, (b, n)
        for p in ("w", "mu", "v"):
            assert _rel(out[f"g_{p}"][b].reshape(-1), gr[p].reshape(-1)) < GRAD_RTOL, (b, n, p)
        assert _rel(out["g_noise"][b], gr["noise"]) < GRAD_RTOL and _rel(out["g_mean"][b], gr["mean"]) < GRAD_RTOL
    # the same call again (nothing uploaded: same lengths) and with a workspace that holds 5 light curves (sets are split)
    again = evaluate_ragged(curves, device=dev)
    small = evaluate_ragged(curves, device=dev, chunk=5)
    torch.cuda.synchronize()
    assert torch.equal(again["mll"], out["mll"]) and torch.equal(again["g_mu"], out["g_mu"])
    assert torch.equal(small["mll"], out["mll"])
    for p in ("w", "mu", "v"):
        assert _rel(small[f"g_{p}"].reshape(-1), out[f"g_{p}"].reshape(-1)) < 1e-10


def test_trimmed_ragged_set_against_padded_sets_and_singles(dev, monkeypatch):
    """40 light curves of 2 .. 12 block rows: by default ONE trimmed launch set of the panel sweep (every member stops at its own
    last block row: the workgroups of tiles it does not have leave at once) -- against the padded sets of ``PGM_RAGGED_TRIM=0``
    (two sets, of 12 and of 4 block rows, the shorter members padded with identity blocks): the same
    value bit for bit -- every tile receives its sources in ascending order whatever the sweep -- and gradients equal to the
    rounding of their differently split sums; a few members against their own single evaluation and the oracle; a member
    that is not positive definite inside a trimmed set; value-only."""
    rng = np.random.default_rng(23)
    lengths = [int(v) for v in rng.integers(130, 1537, size=40)]
    lengths[7], lengths[31] = 1536, 130
    curves = _ragged_curves(lengths, first=100)
    outs = {}
    for mode in ("1", "0"):
        _hip.release_workspaces()
        monkeypatch.setenv("PGM_RAGGED_TRIM", mode)
        set_of, nbs = _hip.ragged_plan(lengths, 64)
        assert nbs == ([12] if mode == "1" else [12, 4])
        o = evaluate_ragged(curves, device=dev)
        o0 = evaluate_ragged(curves, device=dev, need_grad=False)
        torch.cuda.synchronize()
        assert torch.equal(o["mll"], o0["mll"])
        outs[mode] = o
    monkeypatch.delenv("PGM_RAGGED_TRIM")
    _hip.release_workspaces()
    a, b_ = outs["1"], outs["0"]
    assert int(a["info"].abs().max()) == 0 and torch.equal(a["mll"], b_["mll"])
    for p in ("w", "mu", "v"):
        assert _rel(a[f"g_{p}"].reshape(-1), b_[f"g_{p}"].reshape(-1)) < 1e-10, p
    for i, n in enumerate(lengths):
        assert a["g_noise"][i].shape == (n,) and _rel(a["g_noise"][i], b_["g_noise"][i]) < 1e-10 and _rel(a["g_mean"][i], b_["g_mean"][i]) < 1e-10
    for i in (7, 31, 0, 19):
        c, n = curves[i], lengths[i]
        single = _hip_eval(dev, c["x"].reshape(n, 1), c["y"], c["mean"], c["noise"], c["w"], c["mu"], c["v"])
        assert float(single["mll"]) == float(a["mll"][i]), (i, n)
        for p in ("w", "mu", "v"):
            assert _rel(single[f"g_{p}"].reshape(-1), a[f"g_{p}"][i].reshape(-1)) < 1e-10, (i, n, p)
        val, gr = orc.mll_value_grad_closed_form(c["x"].reshape(n, 1), c["y"], c["mean"].expand(n), c["noise"], c["w"], c["mu"], c["v"])
        assert abs(float(val) - float(a["mll"][i])) < MLL_TOL, (i, n)
        for p in ("w", "mu", "v"):
            assert _rel(a[f"g_{p}"][i].reshape(-1), gr[p].reshape(-1)) < GRAD_RTOL, (i, n, p)
        assert _rel(a["g_noise"][i], gr["noise"]) < GRAD_RTOL and _rel(a["g_mean"][i], gr["mean"]) < GRAD_RTOL
    # a member that is not positive definite: its pivot, NaNs; the others keep their bits
    bad = [dict(c) for c in curves]
    bad[12]["noise"] = -5.0 * torch.ones(lengths[12], dtype=D)
    ob = evaluate_ragged(bad, device=dev)
    torch.cuda.synchronize()
    info = ob["info"].cpu().tolist()
    assert info[12] > 0 and sum(1 for v_ in info if v_ != 0) == 1 and math.isnan(float(ob["mll"][12]))
    keep = [i for i in range(40) if i != 12]
    assert torch.equal(ob["mll"][keep], a["mll"][keep]) and torch.equal(ob["g_mu"][keep], a["g_mu"][keep])


def test_trimmed_ragged_sets_of_short_light_curves_and_of_more_than_the_workspace_holds(dev):
    """The ends of the trimmed form: 160 light curves of 20 .. 300 points (one to three block rows: the set's sweep is a single
    panel, most members leave after their first block row) and the same batch through a workspace of 64 slots (three trimmed
    sets, one after the other) -- every value bit for bit the light curve's own single evaluation, gradients to rounding, a few
    against the oracle."""
    rng = np.random.default_rng(5)
    lengths = [int(v) for v in rng.integers(20, 301, size=160)]
    lengths[0], lengths[1], lengths[2] = 300, 20, 128
    curves = _ragged_curves(lengths, first=200)
    set_of, nbs = _hip.ragged_plan(lengths, 160)
    assert nbs == [3]
    set_of, nbs = _hip.ragged_plan(lengths, 64)
    assert len(nbs) == 3 and [set_of.count(k) for k in range(3)] == [64, 64, 32]
    out = evaluate_ragged(curves, device=dev)
    part = evaluate_ragged(curves, device=dev, chunk=64)
    torch.cuda.synchronize()
    assert int(out["info"].abs().max()) == 0 and torch.equal(out["mll"], part["mll"])
    for p in ("w", "mu", "v"):
        assert _rel(out[f"g_{p}"].reshape(-1), part[f"g_{p}"].reshape(-1)) < 1e-10, p
    for i in list(range(0, 160, 9)) + [1, 2]:
        c, n = curves[i], lengths[i]
        single = _hip_eval(dev, c["x"].reshape(n, 1), c["y"], c["mean"], c["noise"], c["w"], c["mu"], c["v"])
        assert float(single["mll"]) == float(out["mll"][i]), (i, n)
        for p in ("w", "mu", "v"):
            assert _rel(single[f"g_{p}"].reshape(-1), out[f"g_{p}"][i].reshape(-1)) < 1e-10, (i, n, p)
        assert _rel(single["g_noise"], out["g_noise"][i]) < 1e-10 and _rel(single["g_mean"], out["g_mean"][i]) < 1e-10
    for i in (0, 1, 77):
        c, n = curves[i], lengths[i]
        val, gr = orc.mll_value_grad_closed_form(c["x"].reshape(n, 1), c["y"], c["mean"].expand(n), c["noise"], c["w"], c["mu"], c["v"])
        assert abs(float(val) - float(out["mll"][i])) < MLL_TOL, (i, n)
        for p in ("w", "mu", "v"):
            assert _rel(out[f"g_{p}"][i].reshape(-1), gr[p].reshape(-1)) < GRAD_RTOL, (i, n, p)


def test_ragged_set_is_bit_for_bit_the_equal_length_batch(dev):
    """Inside a launch set every light curve runs the schedule of the set's (block rows, members): a member whose own block
    rows are the set's gets, gradients included, the bits of the equal-length batched call of the same shape; and padded
    arrays straight from the caller (no list of curves) give the same as the list form."""
    lengths = [1500, 1290, 1409, 1536, 1300, 1480]                       # 12, 11, 12, 12, 11, 12 block rows -> one set of 12
    set_of, nbs = _hip.ragged_plan(lengths, 16)
    assert nbs == [12] and set(set_of) == {0}
    curves = _ragged_curves(lengths, first=40)
    out = evaluate_ragged(curves, device=dev)
    padded, lens = pad_curves(curves, device=dev)
    out2 = evaluate_ragged(padded=padded, lengths=lens)
    torch.cuda.synchronize()
    assert torch.equal(out["mll"], out2["mll"]) and torch.equal(out["g_v"], out2["g_v"])
    for b in (0, 3):
        c, n = curves[b], lengths[b]
        rep = lambda t: t.to(dev).unsqueeze(0).repeat(len(lengths), *([1] * t.dim())).contiguous()
        eq = _hip.mll_value_grad(rep(c["x"].reshape(n, 1)), rep(c["y"]), rep(c["mean"].expand(n)), rep(c["noise"]), None,
                                 rep(c["w"]), rep(c["mu"]), rep(c["v"]), 0, 0.0, True)
        torch.cuda.synchronize()
        assert float(eq["mll"][b]) == float(out["mll"][b])
        for p in ("w", "mu", "v"):
            assert torch.equal(eq[f"g_{p}"][b], out[f"g_{p}"][b]), (b, p)
        assert torch.equal(eq["g_noise"][b], out["g_noise"][b]) and torch.equal(eq["g_mean"][b], out["g_mean"][b])


def test_ragged_batch_reports_the_member_that_fails(dev):
    """A member whose matrix is not positive definite: its ``info`` is the pivot, its outputs NaN; the others are untouched."""
    lengths = [300, 700, 520, 150]
    curves = _ragged_curves(lengths, first=60)
    curves[2]["noise"] = -5.0 * torch.ones(lengths[2], dtype=D)
    out = evaluate_ragged(curves, device=dev)
    good = evaluate_ragged([curves[0], curves[1], curves[3]], device=dev)
    torch.cuda.synchronize()
    info = out["info"].cpu().tolist()
    assert info[2] > 0 and [info[0], info[1], info[3]] == [0, 0, 0]
    assert math.isnan(float(out["mll"][2])) and bool(torch.isnan(out["g_w"][2]).all()) and bool(torch.isnan(out["g_noise"][2]).all())
    for a, b in ((0, 0), (1, 1), (3, 2)):
        assert abs(float(out["mll"][a]) - float(good["mll"][b])) < 1e-13
    # argument checks of the entry point: a length beyond the pitch, an empty light curve
    padded, lens = pad_curves(curves[:2], device=dev)
    with pytest.raises(ValueError):
        evaluate_ragged(padded=padded, lengths=[lens[0], padded["y"].shape[1] + 1])
    with pytest.raises(ValueError):
        evaluate_ragged(padded=padded, lengths=[0, lens[1]])


def test_dense_kernel_entry_point(dev):
    gen = torch.Generator().manual_seed(5)
    x1 = torch.rand(70, 2, generator=gen, dtype=D) * 10
    x2 = torch.rand(45, 2, generator=gen, dtype=D) * 10
    w = torch.rand(3, generator=gen, dtype=D); mu = torch.rand(3, 2, generator=gen, dtype=D); v = torch.rand(3, 2, generator=gen, dtype=D) * 0.2
    for order in (0, 1):
        K = _hip.sm_kernel_dense(x1.to(dev), x2.to(dev), w, mu, v, dim_order=order).cpu()
        assert torch.allclose(K, orc.sm_kernel(x1, x2, w, mu, v, order), atol=1e-13)
    xd = x1.to(dev)
    K = _hip.sm_kernel_dense(xd, xd, w, mu, v, noise=torch.full((70,), 0.5, dtype=D), noise_scalar=0.25).cpu()
    assert torch.allclose(K, orc.sm_kernel(x1, x1, w, mu, v) + 0.75 * torch.eye(70, dtype=D), atol=1e-13)


def _make_model(dev, x, y, lik, Q, d=1, mean="constant", dtype=D):
    class Model(g.models.ExactGP):
        def __init__(self):
            super().__init__(x, y, lik)
            self.mean_module = g.means.ConstantMean() if mean == "constant" else g.means.LinearMean(input_size=d)
            self.covar_module = g.kernels.SpectralMixtureKernel(num_mixtures=Q, ard_num_dims=d)

        def forward(self, xx):
            return g.distributions.MultivariateNormal(self.mean_module(xx), self.covar_module(xx))

    return Model().to(dtype).to(dev)


def test_surface_fixed_noise_matches_oracle_autograd(dev, golden_dir):
    """model(x) -> mll(output, y) -> backward() (pgmuvi/trainers.py:179-181) against oracle
    autograd through the same raw parameters and softplus/sigmoid transforms."""
    inp = _load(golden_dir, "inputs_cfg2_n512.npz")
    x = torch.as_tensor(inp["x"], dtype=D); y = torch.as_tensor(inp["y"], dtype=D); noise = torch.as_tensor(inp["yerr"], dtype=D) ** 2
    h = syn.cfg_hypers(2, y)
    lik = g.likelihoods.FixedNoiseGaussianLikelihood(noise.to(dev))
    model = _make_model(dev, x.to(dev), y.to(dev), lik, 4)
    model.mean_module.register_constraint("raw_constant", g.constraints.Interval(float(y.min()), float(y.max())))
    model.covar_module.register_constraint("raw_mixture_means", g.constraints.GreaterThan(1.0 / 3450.0))
    model.initialize(**{"covar_module.mixture_weights": h["w"].to(dev), "covar_module.mixture_means": h["mu"].to(dev),
                        "covar_module.mixture_scales": h["v"].to(dev), "mean_module.constant": h["mean"].to(dev)})
    model.train(); lik.train()
    mll = g.mlls.ExactMarginalLogLikelihood(lik, model)
    loss = -mll(model(x.to(dev)), y.to(dev))
    loss.backward()
    raw = {n: p.detach().cpu().clone().requires_grad_(True) for n, p in model.named_parameters()}
    assert set(raw) == {"mean_module.raw_constant", "covar_module.raw_mixture_weights",
                        "covar_module.raw_mixture_means", "covar_module.raw_mixture_scales"}
    # GPyTorch keeps constraint bounds as float32 tensors: (ub - lb) is formed in fp32
    c = orc.interval(raw["mean_module.raw_constant"], torch.tensor(float(y.min())).float(), torch.tensor(float(y.max())).float())
    w = orc.positive(raw["covar_module.raw_mixture_weights"])
    mu = orc.greater_than(raw["covar_module.raw_mixture_means"], torch.tensor(1.0 / 3450.0).float()).reshape(4, 1)
    v = orc.positive(raw["covar_module.raw_mixture_scales"]).reshape(4, 1)
    ref = -orc.mll(x, y, c, noise, w, mu, v)
    ref.backward()
    assert abs(float(loss.detach()) - float(ref.detach())) < MLL_TOL
    for n, p in model.named_parameters():
        assert _rel(p.grad, raw[n].grad) < GRAD_RTOL, n


def test_surface_gaussian_likelihood_linear_mean_2d(dev, golden_dir):
    inp = _load(golden_dir, "inputs_cfg4_n256.npz")
    x = torch.as_tensor(inp["x"], dtype=D); y = torch.as_tensor(inp["y"], dtype=D)
    lik = g.likelihoods.GaussianLikelihood().double().to(dev)
    model = _make_model(dev, x.to(dev), y.to(dev), lik, 3, d=2, mean="linear")
    h = syn.cfg_hypers(4, y)
    model.initialize(**{"covar_module.mixture_weights": h["w"].to(dev), "covar_module.mixture_means": h["mu"].to(dev),
                        "covar_module.mixture_scales": h["v"].to(dev), "likelihood.noise_covar.noise": torch.tensor([0.03], dtype=D, device=dev)})
    names = [n for n, _ in model.named_parameters()]
    assert names[0] == "likelihood.noise_covar.raw_noise" and "mean_module.weights" in names and "mean_module.bias" in names
    model.train(); lik.train()
    mll = g.mlls.ExactMarginalLogLikelihood(lik, model)
    loss = -mll(model(x.to(dev)), y.to(dev))
    loss.backward()
    raw = {n: p.detach().cpu().clone().requires_grad_(True) for n, p in model.named_parameters()}
    meanv = (x @ raw["mean_module.weights"]).squeeze(-1) + raw["mean_module.bias"]
    sig2 = orc.greater_than(raw["likelihood.noise_covar.raw_noise"], torch.tensor(1e-4).float()).reshape(())
    ref = -orc.mll(x, y, meanv, sig2, orc.positive(raw["covar_module.raw_mixture_weights"]),
                   orc.positive(raw["covar_module.raw_mixture_means"]).reshape(3, 2), orc.positive(raw["covar_module.raw_mixture_scales"]).reshape(3, 2))
    ref.backward()
    assert abs(float(loss) - float(ref)) < MLL_TOL
    for n, p in model.named_parameters():
        assert _rel(p.grad, raw[n].grad) < GRAD_RTOL, n
    with pytest.raises(RuntimeError, match="ard_num_dims"):     # reference tests/test_2d_integration.py:167-186
        model.covar_module(x[:, :1].to(dev))


def test_priors_are_added_before_division_by_n(dev):
    x = torch.linspace(0, 40, 60, dtype=D); y = torch.sin(x / 3)
    lik = g.likelihoods.FixedNoiseGaussianLikelihood(torch.full((60,), 0.02, dtype=D, device=dev))
    model = _make_model(dev, x.to(dev), y.to(dev), lik, 2)
    model.initialize(**{"covar_module.mixture_weights": torch.tensor([0.5, 0.2], dtype=D), "covar_module.mixture_means": torch.tensor([0.05, 0.2], dtype=D).reshape(2, 1, 1),
                        "covar_module.mixture_scales": torch.tensor([0.01, 0.02], dtype=D).reshape(2, 1, 1)})
    mll = g.mlls.ExactMarginalLogLikelihood(lik, model)
    model.train()
    base = float(mll(model(x.to(dev)), y.to(dev)))
    model.covar_module.register_prior("mixture_weights_prior", g.priors.LogNormalPrior(torch.tensor(0.0, dtype=D, device=dev), torch.tensor(1.0, dtype=D, device=dev)), "mixture_weights")
    with_prior = float(mll(model(x.to(dev)), y.to(dev)))
    lp = torch.distributions.LogNormal(0.0, 1.0).log_prob(torch.tensor([0.5, 0.2], dtype=D)).sum()
    assert abs(with_prior - (base + float(lp) / 60)) < 1e-10


def test_not_psd_raises_and_jitter_policy(dev):
    x = torch.linspace(0, 10, 50, dtype=D); y = torch.randn(50, generator=torch.Generator().manual_seed(1), dtype=D)
    lik = g.likelihoods.GaussianLikelihood(noise_constraint=g.constraints.Interval(-10.0, -1.0)).double().to(dev)   # negative "noise"
    model = _make_model(dev, x.to(dev), y.to(dev), lik, 1)
    mll = g.mlls.ExactMarginalLogLikelihood(lik, model)
    model.train()
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        with pytest.raises(g.utils.errors.NotPSDError):
            mll(model(x.to(dev)), y.to(dev))
    assert sum("added jitter" in str(r.message) for r in rec) == 3       # 1e-8, 1e-7, 1e-6 (fp64 policy)
    out = _hip_eval(dev, x, y, 0.0, torch.full((50,), -3.0, dtype=D), torch.ones(1, dtype=D), torch.ones(1, 1, dtype=D), torch.ones(1, 1, dtype=D))
    assert int(out["info"]) >= 1 and math.isnan(float(out["mll"]))


def test_posterior_prediction_vs_oracle(dev, golden_dir):
    inp = _load(golden_dir, "inputs_cfg2_n512.npz")
    x = torch.as_tensor(inp["x"], dtype=D); y = torch.as_tensor(inp["y"], dtype=D); noise = torch.as_tensor(inp["yerr"], dtype=D) ** 2
    h = syn.cfg_hypers(2, y)
    lik = g.likelihoods.FixedNoiseGaussianLikelihood(noise.to(dev))
    model = _make_model(dev, x.to(dev), y.to(dev), lik, 4)
    model.initialize(**{"covar_module.mixture_weights": h["w"].to(dev), "covar_module.mixture_means": h["mu"].to(dev),
                        "covar_module.mixture_scales": h["v"].to(dev), "mean_module.constant": h["mean"].to(dev)})
    model.eval(); lik.eval()
    xs = torch.linspace(float(x.min()), float(x.max()), 1000, dtype=D)
    with torch.no_grad(), g.settings.fast_pred_var(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        pred = lik(model(xs.to(dev)))
    pm, pv = orc.posterior(x, y, h["mean"], noise, h["w"], h["mu"].reshape(4, 1), h["v"].reshape(4, 1), xs, h["mean"])
    assert torch.allclose(pred.mean.cpu(), pm, atol=1e-8)
    assert torch.allclose(pred.variance.cpu(), pv, atol=1e-8)
    lo, hi = pred.confidence_region()
    assert torch.allclose((hi - lo).cpu(), 4 * pv.clamp_min(0).sqrt(), atol=1e-7)


@pytest.mark.parametrize("n", [1, 40, 89, 128])
def test_prediction_after_a_one_launch_evaluation(dev, n):
    """``pgm_predict_f64`` behind k_small: the one-launch evaluation does not write the identity padding of the diagonal block's
    inverse images (rows of sub-blocks the light curve does not have), prediction completes it (k_small_pad) -- here with STALE
    values there on purpose: a 128-point evaluation on the same workspace first.  Posterior mean and variance against the oracle."""
    _hip.release_workspaces()
    ws = _hip.Workspace(dev, 128, 2, 1, 1)
    gen = torch.Generator().manual_seed(50 + n)
    w = torch.tensor([0.6, 0.3], dtype=D); mu = torch.tensor([[0.02], [0.11]], dtype=D); v = torch.tensor([[0.003], [0.01]], dtype=D)
    for m in (128, n):
        x = torch.sort(torch.rand(m, generator=gen, dtype=D) * 300)[0]
        y = torch.randn(m, generator=gen, dtype=D)
        noise = 0.01 + 0.05 * torch.rand(m, generator=gen, dtype=D)
        out = _hip.mll_value_grad(x.reshape(m, 1).to(dev), y.to(dev), torch.full((m,), 0.2, dtype=D, device=dev), noise.to(dev), None,
                                  w.to(dev), mu.to(dev), v.to(dev), 0, 0.0, True, workspace=ws)
        assert int(out["info"]) == 0
    xs = torch.linspace(-10.0, 310.0, 257, dtype=D)
    pm, pv = _hip.predict(ws, xs.reshape(-1, 1).to(dev), torch.full((257,), 0.2, dtype=D, device=dev))
    torch.cuda.synchronize()
    rm, rv = orc.posterior(x, y, 0.2, noise, w, mu, v, xs, 0.2)
    assert torch.allclose(pm.cpu(), rm, atol=1e-9) and torch.allclose(pv.cpu(), rv, atol=1e-9)
    ws.close()


def test_posterior_prediction_at_the_reference_size(dev):
    """Eval-mode prediction the way ``Lightcurve.plot()`` / ``to_table()`` ask for it (/root/reference/pgmuvi/lightcurve.py:9607-9623:
    ``torch.linspace(x.min(), x.max(), 10000)`` under ``fast_pred_var``) at the headline size: N = 4096 training points (config 2's
    light curve), 10 000 test points, against the oracle's dense posterior."""
    t, y, e = syn.cfg2(n_obs=4096)
    x, y, noise = t.double(), y.double(), e.double() ** 2
    h = syn.cfg_hypers(2, y)
    lik = g.likelihoods.FixedNoiseGaussianLikelihood(noise.to(dev))
    model = _make_model(dev, x.to(dev), y.to(dev), lik, 4)
    model.initialize(**{"covar_module.mixture_weights": h["w"].to(dev), "covar_module.mixture_means": h["mu"].to(dev),
                        "covar_module.mixture_scales": h["v"].to(dev), "mean_module.constant": h["mean"].to(dev)})
    model.eval(); lik.eval()
    xs = torch.linspace(float(x.min()), float(x.max()), 10000, dtype=D)
    with torch.no_grad(), g.settings.fast_pred_var(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        pred = lik(model(xs.to(dev)))
        mean, var = pred.mean.cpu(), pred.variance.cpu()
    pm, pv = orc.posterior(x, y, h["mean"], noise, h["w"], h["mu"].reshape(4, 1), h["v"].reshape(4, 1), xs, h["mean"])
    assert mean.shape == (10000,) and bool(torch.isfinite(mean).all()) and bool(torch.isfinite(var).all())
    # posterior mean |values| ~ 1, latent variances 1e-4 .. 1: both to 1e-8 absolute (observed ~1e-12)
    assert float((mean - pm).abs().max()) < 1e-8, float((mean - pm).abs().max())
    assert float((var - pv).abs().max()) < 1e-8, float((var - pv).abs().max())
    assert float(var.min()) > -1e-10                              # a variance: no cancellation below zero beyond round-off


def test_train_loop_mirrors_reference_results(dev):
    """pgmuvi.trainers.train semantics (reference tests/test_2d_integration.py:110,131-135:
    loss list non-empty and decreasing)."""
    from pgmuvi_amd.trainers import train
    t, y, e = syn.cfg2(n_obs=256)
    lik = g.likelihoods.FixedNoiseGaussianLikelihood((e.double() ** 2).to(dev))
    model = _make_model(dev, t.double().to(dev), y.double().to(dev), lik, 4)
    h = syn.cfg_hypers(2, y.double())
    model.initialize(**{"covar_module.mixture_weights": h["w"].to(dev) * 0.5, "covar_module.mixture_means": h["mu"].to(dev) * 1.05,
                        "covar_module.mixture_scales": h["v"].to(dev) * 2.0})
    res = train(model=model, likelihood=lik, train_x=t.double().to(dev), train_y=y.double().to(dev), maxiter=40, lr=0.02,
                optim="AdamW", progress=False)
    assert len(res["loss"]) == 40 and len(res["delta_loss"]) == 39
    assert np.mean(res["loss"][-5:]) < np.mean(res["loss"][:5])
    assert "covar_module.raw_mixture_means" in res and len(res["covar_module.raw_mixture_means"]) == 40
    with pytest.raises(NotImplementedError):
        train(model=model, likelihood=lik, train_x=t, train_y=y, lossfn="elbo")
    with pytest.raises(ValueError):
        train(model=model, likelihood=lik)


def test_float32_model_is_upcast(dev):
    t, y, e = syn.cfg2(n_obs=200)
    lik = g.likelihoods.FixedNoiseGaussianLikelihood((e ** 2).to(dev))
    model = _make_model(dev, t.to(dev), y.to(dev), lik, 4, dtype=torch.float32)
    h = syn.cfg_hypers(2, y.double())
    model.initialize(**{"covar_module.mixture_weights": h["w"].float().to(dev), "covar_module.mixture_means": h["mu"].float().to(dev),
                        "covar_module.mixture_scales": h["v"].float().to(dev)})
    model.train()
    mll = g.mlls.ExactMarginalLogLikelihood(lik, model)
    loss = -mll(model(t.to(dev)), y.to(dev))
    loss.backward()
    assert loss.dtype == torch.float32 and model.covar_module.raw_mixture_means.grad.dtype == torch.float32
    w = model.covar_module.mixture_weights.detach().cpu().double(); mu = model.covar_module.mixture_means.detach().cpu().double().reshape(4, 1)
    v = model.covar_module.mixture_scales.detach().cpu().double().reshape(4, 1)
    ref = -orc.mll(t.double(), y.double(), 0.0, (e ** 2).double(), w, mu, v)
    assert abs(float(loss) - float(ref)) < 1e-5       # fp32 parameters, fp64 arithmetic inside


def test_cpu_tensors_are_refused(dev):
    x = torch.linspace(0, 1, 8, dtype=D)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _hip.mll_value_grad(x, x, x, x, None, torch.ones(1, dtype=D), torch.ones(1, 1, dtype=D), torch.ones(1, 1, dtype=D))


def test_config4_shape_2d_8_bands(dev):
    """Config 4 (2-D multiwavelength, 8 bands): 8 x 256 points against the oracle, and the
    full 8 x 1024 = 8192-point problem through size-independent properties."""
    x, y, e = syn.cfg4(n_per_band=256)
    xd, yd, nz = x.double(), y.double(), e.double() ** 2
    h = syn.cfg_hypers(4, yd)
    w, mu, v = h["w"], h["mu"].reshape(3, 2), h["v"].reshape(3, 2)
    for order in (0, 1):
        val, gr = orc.mll_value_grad_closed_form(xd, yd, h["mean"], nz, w, mu, v, order)
        out = _hip_eval(dev, xd, yd, h["mean"], nz, w, mu, v, order)
        assert int(out["info"]) == 0 and abs(float(out["mll"]) - float(val)) < MLL_TOL
        for p in ("w", "mu", "v", "noise", "mean"):
            assert _rel(out[f"g_{p}"].reshape(-1), gr[p].reshape(-1)) < GRAD_RTOL, (p, order)
    x, y, e = syn.cfg4()                                   # N = 8192, d = 2, Q = 3
    xd, yd, nz = x.double(), y.double(), e.double() ** 2
    assert xd.shape == (8192, 2)
    h = syn.cfg_hypers(4, yd)
    base = _hip_eval(dev, xd, yd, h["mean"], nz, w, mu, v)
    assert int(base["info"]) == 0 and math.isfinite(float(base["mll"]))
    perm = torch.randperm(8192, generator=torch.Generator().manual_seed(4))
    pout = _hip_eval(dev, xd[perm], yd[perm], h["mean"], nz[perm], w, mu, v)
    assert abs(float(base["mll"]) - float(pout["mll"])) < 1e-9
    for p in ("w", "mu", "v"):
        assert _rel(pout[f"g_{p}"], base[f"g_{p}"]) < 1e-7
    assert _rel(pout["g_noise"], base["g_noise"][perm.to(dev)]) < 1e-7
    v0 = _hip_eval(dev, xd, yd, h["mean"], nz, w, mu, v, need_grad=False)
    assert float(v0["mll"]) == float(base["mll"])
    # directional derivative along the gradient of (w, mu, v) against central differences of the value
    g = {p: base[f"g_{p}"].cpu() for p in ("w", "mu", "v")}
    gn = math.sqrt(sum(float((t ** 2).sum()) for t in g.values()))
    eps = 1e-6 / gn
    plus = _hip_eval(dev, xd, yd, h["mean"], nz, w + eps * g["w"], mu + eps * g["mu"], v + eps * g["v"], need_grad=False)
    minus = _hip_eval(dev, xd, yd, h["mean"], nz, w - eps * g["w"], mu - eps * g["mu"], v - eps * g["v"], need_grad=False)
    fd = (float(plus["mll"]) - float(minus["mll"])) / (2 * eps)
    assert abs(fd - gn * gn) < 1e-4 * gn * gn


def test_config3_shape_batch_of_2048_point_curves(dev):
    """Config 3 shape: a shard of N=2048, Q=4 light curves evaluated together; first and last
    against the oracle, all of them finite and mutually independent of batch position."""
    B, n = 6, 2048
    xs, ys, ns, ws, mus, vs, means = [], [], [], [], [], [], []
    for i in range(B):
        (t, y, e), per = syn.cfg3_lightcurve(i, n_obs=n)
        h = syn.cfg_hypers(3, y.double(), lead_period=per)
        xs.append(t.double().reshape(n, 1)); ys.append(y.double()); ns.append(e.double() ** 2)
        ws.append(h["w"]); mus.append(h["mu"].reshape(4, 1)); vs.append(h["v"].reshape(4, 1)); means.append(h["mean"].expand(n))
    st = lambda L: torch.stack(L).to(dev)
    out = evaluate_batch(st(xs), st(ys), st(means), st(ns), st(ws), st(mus), st(vs))
    rev = evaluate_batch(st(xs[::-1]), st(ys[::-1]), st(means[::-1]), st(ns[::-1]), st(ws[::-1]), st(mus[::-1]), st(vs[::-1]))
    torch.cuda.synchronize()
    assert int(out["info"].abs().max()) == 0 and bool(torch.isfinite(out["mll"]).all())
    assert torch.equal(out["mll"], rev["mll"].flip(0)) and torch.equal(out["g_mu"], rev["g_mu"].flip(0))
    for i in (0, B - 1):
        val, gr = orc.mll_value_grad_closed_form(xs[i], ys[i], means[i], ns[i], ws[i], mus[i], vs[i])
        assert abs(float(val) - float(out["mll"][i])) < MLL_TOL
        for p in ("w", "mu", "v"):
            assert _rel(out[f"g_{p}"][i].reshape(-1), gr[p].reshape(-1)) < GRAD_RTOL


def test_notebook_recorded_output(dev, golden_dir):
    """HIP path at the end point of the reference-driven re-run of the comparison notebook's "pgmuvi 1D"
    fit (tests/golden/make_notebook_pin.py): equal to the oracle to round-off, and within the trajectory's
    print precision of the loss the reference's notebook recorded (-1.562)."""
    p = _load(golden_dir, "notebook_pin_1d.npz")
    x, y, noise = (torch.as_tensor(p[k], dtype=D) for k in ("x", "y", "noise"))
    w, mu, v = (torch.as_tensor(p[k], dtype=D) for k in ("final_w", "final_mu", "final_v"))
    c = float(p["final_c"])
    out = _hip_eval(dev, x, y, c, noise, w, mu.reshape(2, 1), v.reshape(2, 1))
    val, gr = orc.mll_value_grad_closed_form(x.reshape(-1, 1), y, torch.full_like(y, c), noise, w, mu.reshape(2, 1), v.reshape(2, 1), 0, 0.0)
    assert int(out["info"]) == 0
    assert abs(float(out["mll"]) - float(val)) < MLL_TOL
    for k in ("w", "mu", "v"):
        assert _rel(out[f"g_{k}"].reshape(-1), gr[k].reshape(-1)) < GRAD_RTOL
    assert abs(-float(out["mll"]) - float(p["nb_final_loss"])) < 6e-3


def test_notebook_fit_on_the_gpu_lands_on_the_recorded_result(dev, golden_dir):
    """The whole fit of that notebook cell on the HIP path (our mirror of trainers.train, the reference's
    default constraints for this light curve are not active at the optimum): 1000 AdamW iterations (the default of fit()) at lr 0.05
    from the notebook's printed initial values reach the recorded loss and frequencies.  At this step size the optimiser
    keeps bouncing around the optimum (the loss swings by 0.03 between iterations 700 and 1000) and two runs whose gradients
    differ in the last bit part ways after ~200 iterations, so the value at one fixed iteration depends on the rounding of
    the gradient sums (it changes with the tile split of the inverse pass); what is asserted is the optimum the run reaches:
    the best loss of the last 300 iterations and the frequencies there."""
    from pgmuvi_amd.trainers import train
    p = _load(golden_dir, "notebook_pin_1d.npz")
    x, y, noise = (torch.as_tensor(p[k], dtype=D).to(dev) for k in ("x", "y", "noise"))
    lik = g.likelihoods.FixedNoiseGaussianLikelihood(noise)
    model = _make_model(dev, x, y, lik, 2)
    model.initialize(**{"mean_module.constant": torch.tensor(float(p["nb_init_constant"]), dtype=D, device=dev),
                        "covar_module.mixture_weights": torch.as_tensor(p["nb_init_weights"], dtype=D).to(dev),
                        "covar_module.mixture_means": torch.as_tensor(p["nb_init_means"], dtype=D).reshape(2, 1, 1).to(dev),
                        "covar_module.mixture_scales": torch.as_tensor(p["nb_init_scales"], dtype=D).reshape(2, 1, 1).to(dev)})
    res = train(model=model, likelihood=lik, train_x=x, train_y=y, maxiter=1000, lr=0.05, optim="AdamW", progress=False)
    loss = np.array([float(v) for v in res["loss"]])
    assert loss.shape == (1000,) and np.all(np.isfinite(loss))
    best = 700 + int(np.argmin(loss[700:]))
    assert abs(loss[best] - float(p["nb_final_loss"])) < 1e-2, (best, loss[best])
    raw = torch.as_tensor(np.asarray(res["covar_module.raw_mixture_means"][best]), dtype=D)
    f = np.sort(model.covar_module.raw_mixture_means_constraint.transform(raw).numpy().reshape(-1))
    assert np.all(np.abs(f / np.sort(p["nb_final_freqs"]) - 1) < 5e-3), f
    assert abs(np.median(loss[900:]) - float(p["nb_final_loss"])) < 5e-2


def _notebook_2d_model(dev, p):
    """The state ``Lightcurve.fit(model='2D', num_mixtures=2)`` trains from in the notebook's 2-D cell: float32 parameters at
    raw = 0 on float64 data, ``set_default_constraints``' Interval constraints (bounds from the fixture)."""
    x, y, noise = (torch.as_tensor(p[k], dtype=D).to(dev) for k in ("x", "y", "noise"))
    lik = g.likelihoods.FixedNoiseGaussianLikelihood(noise)
    assert str(p["param_dtype"]) == "torch.float32"
    model = _make_model(dev, x, y, lik, 2, d=2, dtype=torch.float32)
    model.mean_module.register_constraint("raw_constant", g.constraints.Interval(float(p["constant_bounds"][0]), float(p["constant_bounds"][1])))
    model.covar_module.register_constraint("raw_mixture_means", g.constraints.Interval(float(p["means_bounds"][0]), float(p["means_bounds"][1])))
    return model, lik, x, y, noise


@pytest.mark.parametrize("order", [0, 1])
def test_notebook_2d_recorded_output(dev, golden_dir, order):
    """HIP path at the end points of the reference-driven re-runs of the comparison notebook's "pgmuvi 2D" cell
    (tests/golden/make_notebook_pin.py, one per kernel form): equal to the oracle to round-off; with GPyTorch's
    prod_d sum_q form (dim_order 0) the loss is the one the notebook recorded (0.904), with sum_q prod_d it is not."""
    p = _load(golden_dir, "notebook_pin_2d.npz")
    x, y, noise = (torch.as_tensor(p[k], dtype=D) for k in ("x", "y", "noise"))
    w, mu, v = (torch.as_tensor(p[f"order{order}_final_{k}"], dtype=D) for k in ("w", "mu", "v"))
    c = float(p[f"order{order}_final_c"])
    out = _hip_eval(dev, x, y, c, noise, w, mu, v, order=order)
    val, gr = orc.mll_value_grad_closed_form(x, y, torch.full_like(y, c), noise, w, mu, v, order, 0.0)
    assert int(out["info"]) == 0
    assert abs(float(out["mll"]) - float(val)) < MLL_TOL
    for k in ("w", "mu", "v"):
        assert _rel(out[f"g_{k}"].reshape(-1), gr[k].reshape(-1)) < GRAD_RTOL
    # the stored loss is the one AFTER which the last optimiser step was taken: one step of a converged fit apart
    assert abs(-float(out["mll"]) - float(p[f"order{order}_loss"][-1])) < 1e-4
    if order == 0:
        assert round(-float(out["mll"]), 3) == float(p["nb_final_loss"])
    else:
        assert abs(-float(out["mll"]) - float(p["nb_final_loss"])) > 0.02


def _hip_evaluation_for_cpu_models(dev):
    """``_hip.mll_value_grad`` with host tensors shipped to the MI355X and the results shipped back: lets a model whose
    float32 parameters, constraint transforms and optimiser live on the CPU (bit-for-bit the arithmetic of the reference's
    recorded CPU run) take its value and gradients from the HIP path."""
    real = _hip.mll_value_grad

    def evaluate(x, y, mean, noise, noise_scalar, w, mu, v, dim_order=0, jitter=0.0, need_grad=True, workspace=None):
        up = lambda t: t.to(dev) if torch.is_tensor(t) else t
        out = real(up(x), up(y), up(mean), up(noise), up(noise_scalar), up(w), up(mu), up(v), dim_order, jitter, need_grad)
        return {k: (val.cpu() if torch.is_tensor(val) else val) for k, val in out.items() if k != "_keep"}
    return evaluate


def test_notebook_2d_fit_with_hip_evaluations_lands_on_the_recorded_result(dev, golden_dir, monkeypatch):
    """The whole 2-D fit of that notebook cell with every value and gradient from the HIP path (our mirror of
    trainers.train; fit()'s defaults AdamW, stop 1e-5 over 30 losses, plus the cell's lr 0.05 / miniter 50 / 1000 iterations).
    The start is deterministic and the run must end where the reference's recorded output says it ended -- early stop around
    loop index 348, loss 0.904, time frequencies 13.8426 -- following the trajectory of the reference's own
    ``Lightcurve.fit`` stored in the fixture.  The float32 parameters, their sigmoid/softplus transforms and AdamW stay on
    the CPU here, as in the recorded run: this trajectory amplifies last-bit differences of float32 arithmetic (the same fit
    with parameters and torch's AdamW on the GPU drifts off after ~100 iterations and stops in a neighbouring optimum, next
    test), so only the evaluation -- the thing under test -- is moved to the GPU."""
    from pgmuvi_amd.trainers import train
    p = _load(golden_dir, "notebook_pin_2d.npz")
    cpu = torch.device("cpu")
    model, lik, x, y, _ = _notebook_2d_model(cpu, p)
    init = model.covar_module.mixture_means.detach().numpy().reshape(-1)
    assert np.allclose(init, p["nb_init_means"], atol=5e-5)                 # 9.4067 = midpoint of the Interval at raw 0
    assert abs(float(model.mean_module.constant) - float(p["nb_init_constant"])) < 1e-7
    monkeypatch.setattr(_hip, "mll_value_grad", _hip_evaluation_for_cpu_models(dev))
    res = train(model=model, likelihood=lik, train_x=x, train_y=y, maxiter=1000, miniter=50, stop=1e-5, stopavg=30, lr=0.05,
                optim="AdamW", progress=False)
    loss = np.asarray([float(v) for v in res["loss"]])
    f = model.covar_module.mixture_means.detach().numpy()[:, 0, 0]
    ref = p["order0_loss"]
    m = min(len(ref), len(loss))
    # Same trajectory: bit for bit (one float32 ulp of the loss) over the first 200 iterations, back within 2e-5 at the end; in
    # between a steep stretch of the descent may be entered an iteration apart (1e-2 in the loss at a given index).  The
    # HIP path sums in another order than LAPACK (1e-14 in the value, up to 1e-9 relative in a gradient component that is
    # itself a near-cancelling sum close to the optimum); Adam's normalised steps turn that into last-digit differences of the
    # float32 parameters, so the stop rule -- std of 30 losses that differ in the 6th digit against 1e-5 -- fires a few
    # iterations from the recorded 348 (the oracle, like GPyTorch on LAPACK, hits 348 exactly: tests/test_dropin_reference.py).
    assert np.max(np.abs(ref[:200] - loss[:200])) < 2e-6 and np.max(np.abs(ref[:m] - loss[:m])) < 0.05
    assert np.max(np.abs(ref[m - 20:m] - loss[m - 20:m])) < 2e-5
    assert abs((len(loss) - 1) - int(p["nb_progress_bar_stop"])) <= 25, len(loss)
    assert round(float(loss[-1]), 3) == float(p["nb_final_loss"]), loss[-1]
    assert np.all(np.abs(f / p["nb_final_time_freqs"] - 1) < 2e-4), f


def test_notebook_2d_fit_all_on_the_gpu(dev, golden_dir):
    """Same fit with model, transforms and torch's AdamW on the GPU: the first 50 iterations follow the recorded-run
    trajectory (to float32 rounding of the parameters), the fit converges by the stop rule to a loss no worse than the
    recorded one."""
    from pgmuvi_amd.trainers import train
    p = _load(golden_dir, "notebook_pin_2d.npz")
    model, lik, x, y, _ = _notebook_2d_model(dev, p)
    res = train(model=model, likelihood=lik, train_x=x, train_y=y, maxiter=1000, miniter=50, stop=1e-5, stopavg=30, lr=0.05,
                optim="AdamW", progress=False)
    loss = np.asarray([float(v) for v in res["loss"]])
    assert np.max(np.abs(loss[:50] - p["order0_loss"][:50])) < 1e-4
    assert len(loss) < 1000 and np.isfinite(loss).all() and loss[-1] < float(p["nb_final_loss"]) + 1e-3


def test_device_resident_training_loop_equals_the_host_loop(dev):
    """``train_device`` (one captured iteration replayed, losses and
    parameters logged on the device) follows the same trajectory as the reference-shaped ``train``."""
    from pgmuvi_amd.trainers import train, train_device
    t, y, e = syn.cfg2(n_obs=300)
    x, yy, nz = t.double().to(dev), y.double().to(dev), (e.double() ** 2).to(dev)
    h = syn.cfg_hypers(2, y.double())

    def build():
        lik = g.likelihoods.FixedNoiseGaussianLikelihood(nz)
        m = _make_model(dev, x, yy, lik, 4)
        m.initialize(**{"covar_module.mixture_weights": h["w"].to(dev) * 0.7, "covar_module.mixture_means": h["mu"].to(dev) * 1.02,
                        "covar_module.mixture_scales": h["v"].to(dev) * 1.5})
        return m, lik

    for optim in ("AdamW", "SGD"):
        m1, l1 = build(); m2, l2 = build()
        r1 = train(model=m1, likelihood=l1, train_x=x, train_y=yy, maxiter=45, lr=0.01, optim=optim, progress=False)
        r2 = train_device(model=m2, likelihood=l2, train_x=x, train_y=yy, maxiter=45, lr=0.01, optim=optim, check_every=20)
        assert len(r2["loss"]) == 45 and len(r2["delta_loss"]) == 44
        # torch's capturable AdamW keeps its step counter (and hence the bias corrections) in float32 on
        # the device: the two AdamW trajectories agree to fp32 round-off only; SGD is bit-for-bit the same arithmetic
        tol = 1e-9 if optim == "SGD" else 2e-5
        assert np.allclose(np.array(r1["loss"], dtype=float), np.array(r2["loss"], dtype=float), rtol=0, atol=tol), optim
        for (n1, p1), (n2, p2) in zip(m1.named_parameters(), m2.named_parameters()):
            assert n1 == n2 and torch.allclose(p1, p2, rtol=tol * 10, atol=tol), n1
        k = "covar_module.raw_mixture_means"
        assert np.allclose(np.array(r1[k]), np.array(r2[k]), atol=tol * 10)
    # early stop evaluated on the same window as the reference rule
    m3, l3 = build()
    r3 = train_device(model=m3, likelihood=l3, train_x=x, train_y=yy, maxiter=400, miniter=10, stop=1e-2, lr=1e-5, optim="SGD", check_every=16)
    m4, l4 = build()
    r4 = train(model=m4, likelihood=l4, train_x=x, train_y=yy, maxiter=400, miniter=10, stop=1e-2, lr=1e-5, optim="SGD", progress=False)
    assert len(r3["loss"]) == len(r4["loss"]) < 400


def test_nuts_potential_and_short_run_vs_oracle(dev):
    """the posterior potential (total log marginal likelihood + default priors
    + log-Jacobians) and its gradient on the HIP path against the oracle with autograd, each chain on its own light
    curve; then the same short NUTS run driven by the HIP path and by the oracle stand-in visits the same states."""
    import _oracle_backend as ob
    from pgmuvi_amd import mcmc
    C, n, Q = 4, 200, 2
    xs, ys, ns = [], [], []
    for c in range(C):
        (t, y, e), _ = syn.cfg3_lightcurve(5000 + c, n_obs=n)
        xs.append(t.double().reshape(n, 1)); ys.append(y.double()); ns.append(e.double() ** 2)
    x, y, nz = torch.stack(xs), torch.stack(ys), torch.stack(ns)
    for learn_noise in (False, True):
        pot = mcmc.SMPotential(x.to(dev), y.to(dev), None if learn_noise else nz.to(dev), num_mixtures=Q)
        rng = np.random.default_rng(8)
        z = rng.normal(0, 0.4, (C, pot.P))
        z[:, 1 + Q:1 + 2 * Q] += np.log(1 / 150.0)
        z[:, 1 + 2 * Q:1 + 3 * Q] += np.log(1 / 1500.0)
        if learn_noise:
            z[:, -1] = np.log(0.01)
        U, G = pot(z)
        for c in range(C):
            zt = torch.tensor(z[c], dtype=D, requires_grad=True)
            s = 1e-4 * y[c].std()
            ref = orc.nuts_potential(zt, x[c], y[c], None if learn_noise else nz[c], Q, 1, y[c].mean(), y[c].std() / 10,
                                     noise_loc=torch.log(s), noise_scale=s)
            ref.backward()
            assert abs(U[c] - float(ref.detach())) < 1e-9 * n * max(1.0, abs(float(ref.detach())) / n)
            assert np.allclose(G[c], zt.grad.numpy(), rtol=1e-7, atol=1e-7)
    init = {"mean_module.mean_prior": np.array(0.0), "covar_module.mixture_weights_prior": np.array([0.5, 0.1]),
            "covar_module.mixture_means_prior": np.array([1 / 150.0, 1 / 67.0]).reshape(2, 1, 1),
            "covar_module.mixture_scales_prior": np.array([1 / 1500.0, 1 / 700.0]).reshape(2, 1, 1)}
    kw = dict(num_mixtures=Q, num_samples=4, warmup_steps=4, seed=3, group_by_chain=True, max_tree_depth=3, initial_values=init)
    a = mcmc.run_mcmc(x.to(dev), y.to(dev), nz.to(dev), **kw)
    b = mcmc.run_mcmc(x, y, nz, compute=ob.mll_value_grad, **kw)
    assert np.array_equal(a["_diagnostics"]["n_leapfrog"], b["_diagnostics"]["n_leapfrog"])
    for k in ("covar_module.mixture_means_prior", "covar_module.mixture_weights_prior", "mean_module.mean_prior"):
        assert np.allclose(a[k], b[k], rtol=1e-6, atol=1e-9), k


def test_native_potential_equals_the_python_one(dev):
    """``pgm_pot_*`` (round 5: z -> theta, priors, Jacobians and the chain rule on the device, positions and results through
    host-mapped memory, one graph replay per tick) against the same potential assembled on the host from the plain batched
    evaluation: equal to round-off for fixed and learned noise, 1-D and 2-D, tick after tick on one handle; a position whose
    matrix cannot be factored gives U = +inf and a zero gradient in both."""
    from pgmuvi_amd import mcmc
    rng = np.random.default_rng(21)
    for (C, n, Q, d, learn) in ((3, 300, 2, 1, False), (2, 700, 3, 1, True), (2, 260, 2, 2, False), (1, 2048, 4, 1, False)):
        x = torch.sort(torch.rand(C, n, generator=torch.Generator().manual_seed(n), dtype=D) * 900, dim=1)[0]
        xx = x.unsqueeze(-1) if d == 1 else torch.stack([x, torch.rand(C, n, generator=torch.Generator().manual_seed(n + 1), dtype=D) * 2], dim=-1)
        y = torch.randn(C, n, generator=torch.Generator().manual_seed(n + 2), dtype=D)
        nz = 0.02 + 0.05 * torch.rand(C, n, generator=torch.Generator().manual_seed(n + 3), dtype=D)
        native = mcmc.SMPotential(xx.to(dev), y.to(dev), None if learn else nz.to(dev), num_mixtures=Q)
        host = mcmc.SMPotential(xx.to(dev), y.to(dev), None if learn else nz.to(dev), num_mixtures=Q)
        host._use_native = False
        assert native._use_native
        for tick in range(3):
            z = rng.normal(0, 0.3, (C, native.P))
            z[:, 1 + Q:1 + Q + Q * d] += np.log(1 / 120.0)
            z[:, 1 + Q + Q * d:1 + Q + 2 * Q * d] += np.log(1 / 1200.0)
            if learn:
                z[:, -1] = np.log(0.02)
            Un, Gn = native(z)
            Uh, Gh = host(z)
            assert np.isfinite(Un).all() and np.allclose(Un, Uh, rtol=1e-12, atol=1e-9), (C, n, tick, Un, Uh)
            assert np.allclose(Gn, Gh, rtol=1e-9, atol=1e-9 * n)
        if not learn:
            z[0, 1 + Q + Q * d:1 + Q + 2 * Q * d] = np.log(1e-9)          # (all but constant mixtures, tiny noise relative to them: not positive definite in fp64)
            z[0, 1:1 + Q] = np.log(1e8)
            Un, Gn = native(z)
            Uh, Gh = host(z)
            assert np.isinf(Un[0]) == np.isinf(Uh[0]) and np.array_equal(Gn[0] == 0.0, Gh[0] == 0.0)
            assert np.allclose(Un[1:], Uh[1:], rtol=1e-12, atol=1e-9)


def test_config5_at_its_stated_size(dev):
    """BASELINE config 5 as stated -- 8 chains x (N=2048, Q=4), default priors (``pgmuvi/lightcurve.py:3235-3330``) -- in one
    batched HIP evaluation per tick: potential and gradient of the first and the last chain against the oracle's autograd, then
    a short NUTS run of all 8 chains whose chains 0 and 1 must visit the states of the same run driven by the oracle stand-in
    (a chain's random stream depends on the seed and its id only)."""
    import _oracle_backend as ob
    from pgmuvi_amd import mcmc
    C, n, Q = 8, 2048, 4
    xs, ys, ns = [], [], []
    for c in range(C):
        (t, y, e), _ = syn.cfg3_lightcurve(5000 + c, n_obs=n)
        xs.append(t.double().reshape(n, 1)); ys.append(y.double()); ns.append(e.double() ** 2)
    x, y, nz = torch.stack(xs), torch.stack(ys), torch.stack(ns)
    pot = mcmc.SMPotential(x.to(dev), y.to(dev), nz.to(dev), num_mixtures=Q)
    rng = np.random.default_rng(9)
    z = rng.normal(0, 0.3, (C, pot.P))
    z[:, 1 + Q:1 + 2 * Q] += np.log(1.0 / np.array([150.0, 67.0, 400.0, 31.0]))
    z[:, 1 + 2 * Q:1 + 3 * Q] += np.log(0.1 / np.array([150.0, 67.0, 400.0, 31.0]))
    U, G = pot(z)
    assert np.isfinite(U).all() and np.isfinite(G).all()
    for c in (0, C - 1):
        zt = torch.tensor(z[c], dtype=D, requires_grad=True)
        ref = orc.nuts_potential(zt, x[c], y[c], nz[c], Q, 1, y[c].mean(), y[c].std() / 10)
        ref.backward()
        assert abs(U[c] - float(ref.detach())) < 1e-9 * n * max(1.0, abs(float(ref.detach())) / n)
        assert np.allclose(G[c], zt.grad.numpy(), rtol=1e-7, atol=1e-6)
    init = {"mean_module.mean_prior": np.array(0.0), "covar_module.mixture_weights_prior": np.array([0.5, 0.12, 0.05, 0.02]),
            "covar_module.mixture_means_prior": (1.0 / np.array([150.0, 67.0, 400.0, 31.0])).reshape(Q, 1, 1),
            "covar_module.mixture_scales_prior": (0.1 / np.array([150.0, 67.0, 400.0, 31.0])).reshape(Q, 1, 1)}
    kw = dict(num_mixtures=Q, num_samples=2, warmup_steps=1, seed=11, group_by_chain=True, max_tree_depth=1, initial_values=init)
    a = mcmc.run_mcmc(x.to(dev), y.to(dev), nz.to(dev), **kw)
    b = mcmc.run_mcmc(x[:2], y[:2], nz[:2], compute=ob.mll_value_grad, **kw)
    assert a["covar_module.mixture_means_prior"].shape[0] == C
    assert np.array_equal(a["_diagnostics"]["n_leapfrog"][:2], b["_diagnostics"]["n_leapfrog"])
    for k in ("covar_module.mixture_means_prior", "covar_module.mixture_weights_prior", "covar_module.mixture_scales_prior", "mean_module.mean_prior"):
        assert np.allclose(a[k][:2], b[k], rtol=1e-6, atol=1e-9), k


def test_config3_per_gpu_shard_of_64(dev):
    """BASELINE config 3's per-GPU shard at 8 GPUs -- 64 light curves x N=2048, Q=4 -- through the code path of
    ``bench.py --total-batch`` (``make_shard`` + ``sharded_batch_step``): first and last against the oracle, every value finite,
    and the same numbers whether the shard goes in one launch set or in chunks of 24 with a ragged tail."""
    from pgmuvi_amd.batch import make_shard, sharded_batch_step
    B, n = 64, 2048
    shard = make_shard(B, 0, 1, n, "cfg3", dev)
    out, ll = sharded_batch_step(shard, B, 64)
    out24, ll24 = sharded_batch_step(shard, B, 24)
    torch.cuda.synchronize()
    assert ll.shape == (B,) and int(out["info"].abs().max()) == 0 and bool(torch.isfinite(ll).all())
    # (the value does not depend on the launch set; the gradient sums of a set of 64 and more are formed per whole tile, those
    #  of the sets of 24 per k-split work item: equal to rounding)
    assert torch.equal(ll, ll24) and _rel(out["g_mu"].reshape(-1), out24["g_mu"].reshape(-1)) < 1e-10
    cpu = {k: v.cpu() for k, v in shard.items()}
    for i in (0, B - 1):
        val, gr = orc.mll_value_grad_closed_form(cpu["x"][i], cpu["y"][i], cpu["mean"][i], cpu["noise"][i], cpu["w"][i], cpu["mu"][i], cpu["v"][i])
        assert abs(float(val) - float(ll[i])) < MLL_TOL
        for p_ in ("w", "mu", "v"):
            assert _rel(out[f"g_{p_}"][i].reshape(-1), gr[p_].reshape(-1)) < GRAD_RTOL
        # (batches take diag(A^-1), i.e. the noise gradient, from the accumulators of the (j, j) tiles of the inverse pass)
        assert _rel(out["g_noise"][i], gr["noise"]) < GRAD_RTOL and _rel(out["g_mean"][i], gr["mean"]) < GRAD_RTOL
    assert torch.allclose(out["g_noise"], out24["g_noise"], rtol=1e-9, atol=1e-13)


@pytest.mark.parametrize("n", [8320, 16384])
def test_beyond_64_block_rows_vs_the_oracle_fixture(dev, golden_dir, n):
    """include/pgmuvi_hip.h's size contract above N = 8192: 65 block rows (the first size the fused sweep's 64-bit plans do
    not cover: the panel sweep on one matrix) and 128 block rows (pgm_max_n()), config 2's recipe, value and every gradient
    against the committed oracle fixtures; one point more than pgm_max_n() is refused."""
    import hashlib
    exp = _load(golden_dir, f"expect_cfg2_n{n}.npz")
    t, y, e = syn.cfg2(n_obs=n)
    sha = hashlib.sha256()
    for a in (t, y, e):
        sha.update(np.ascontiguousarray(a.numpy()).tobytes())
    assert sha.digest() == exp["inputs_sha256"].tobytes()
    w, mu, v = (torch.as_tensor(exp[f"{p}_0"]) for p in ("w", "mu", "v"))
    out = _hip_eval(dev, t.double().reshape(n, 1), y.double(), torch.as_tensor(exp["meanc_0"]), e.double() ** 2, w, mu, v)
    assert int(out["info"]) == 0
    assert abs(float(out["mll"]) - float(exp["mll_0"])) < MLL_TOL
    for p in ("w", "mu", "v", "noise", "mean"):
        assert _rel(out[f"g_{p}"].reshape(-1), torch.as_tensor(exp[f"g_{p}_0"]).reshape(-1)) < GRAD_RTOL, p
    out0 = _hip_eval(dev, t.double().reshape(n, 1), y.double(), torch.as_tensor(exp["meanc_0"]), e.double() ** 2, w, mu, v, need_grad=False)
    assert float(out0["mll"]) == float(out["mll"])
    _hip.release_workspaces()
    if n == _hip.max_n_limit():
        with pytest.raises(RuntimeError, match="at most 16384"):
            _hip.Workspace(dev, n + 1, 4, 1, 1)


def test_config3_at_its_stated_size_vs_the_oracle_fixture(dev, golden_dir):
    """BASELINE config 3 whole: 512 light curves x N=2048, Q=4 through ``make_shard`` + ``sharded_batch_step`` (the code path of
    ``bench.py --total-batch 512 --npoints 2048``) against the oracle's value of EVERY member and every gradient of members 0, 255
    and 511 (``tests/golden/expect_cfg3_b512_n2048.npz``, ``make_golden.py --fullsize``).  The fixture also carries the SHA-256
    of the 512 light curves as the reference's generator helpers made them: ``make_shard`` must produce exactly those."""
    import hashlib
    from pgmuvi_amd.batch import make_shard, sharded_batch_step
    exp = _load(golden_dir, "expect_cfg3_b512_n2048.npz")
    B, n = 512, 2048
    sha = hashlib.sha256()
    for i in range(B):
        (t, y, e), per = syn.cfg3_lightcurve(i, n_obs=n)
        assert per == float(exp["lead_period"][i])
        for a in (t, y, e):
            sha.update(np.ascontiguousarray(a.numpy()).tobytes())
    assert sha.digest() == exp["inputs_sha256"].tobytes()
    shard = make_shard(B, 0, 1, n, "cfg3", dev)
    out, ll = sharded_batch_step(shard, B)
    torch.cuda.synchronize()
    assert ll.shape == (B,) and int(out["info"].abs().max()) == 0
    dev_ = (ll.cpu() - torch.as_tensor(exp["mll"])).abs()
    assert float(dev_.max()) < MLL_TOL, (int(dev_.argmax()), float(dev_.max()))
    for i in (0, 255, 511):
        for p_ in ("w", "mu", "v", "noise", "mean"):
            assert _rel(out[f"g_{p_}"][i].reshape(-1), torch.as_tensor(exp[f"g_{p_}_{i}"]).reshape(-1)) < GRAD_RTOL, (p_, i)
    _hip.release_workspaces()


def test_lomb_scargle_kernel_vs_oracle(dev, golden_dir):
    """the periodogram kernel behind the astropy-shaped ``LombScargle`` against the numpy
    oracle -- the notebook's 89-point light curve (whose two highest peaks are the initial frequencies the reference
    printed, 0.0067 and 0.0154), config 2 at full size with error bars, a batch, unit weights."""
    from oracle import ls_oracle as lso
    from pgmuvi_amd import lombscargle as L
    p = _load(golden_dir, "notebook_pin_1d.npz")
    t, y, dy = p["x"], p["y"], np.sqrt(p["noise"])
    ls = L.LombScargle(t, y, dy)
    f = ls.autofrequency(nyquist_factor=5)
    pw = ls.power(f, assume_regular_frequency=True)             # what pgmuvi calls: astropy's 'auto' = the FFT approximation here
    assert f.size > 200 and np.allclose(pw, lso.power_fast(t, y, dy, f[0], f[1] - f[0], f.size), rtol=1e-9, atol=1e-11)
    assert np.allclose(ls.power(f, method="slow"), lso.power(t, y, dy, f), rtol=1e-9, atol=1e-12)
    assert np.allclose(ls.power(f[:150]), lso.power(t, y, dy, f[:150]), rtol=1e-9, atol=1e-12)   # short grid: exact sums
    from scipy.signal import find_peaks
    pk, _ = find_peaks(pw, distance=5)
    pk = pk[np.argsort(pw[pk])][::-1]
    assert [round(float(v), 4) for v in f[pk[:2]]] == [round(float(v), 4) for v in p["nb_init_means"]]
    # full size, with error bars and without
    tt, yy, ee = syn.cfg2(n_obs=4096)
    tt, yy, ee = tt.double(), yy.double(), ee.double()
    grid = torch.as_tensor(lso.autofrequency(tt.numpy()), dtype=D)
    sub = grid[::37]                                          # the oracle is O(N Nf) in numpy: a slice of the 51k grid
    for dyv in (ee, None):
        out = L.periodogram_batched(tt.to(dev).reshape(1, -1), yy.to(dev).reshape(1, -1), None if dyv is None else dyv.to(dev).reshape(1, -1), sub.to(dev))
        ref = lso.power(tt.numpy(), yy.numpy(), None if dyv is None else dyv.numpy(), sub.numpy())
        assert np.allclose(out[0].cpu().numpy(), ref, rtol=1e-8, atol=1e-12)
    fast = L.periodogram_batched(tt.to(dev).reshape(1, -1), yy.to(dev).reshape(1, -1), ee.to(dev).reshape(1, -1), grid.to(dev), method="auto")
    gnp = grid.numpy()
    assert np.allclose(fast[0].cpu().numpy(), lso.power_fast(tt.numpy(), yy.numpy(), ee.numpy(), gnp[0], gnp[1] - gnp[0], gnp.size), rtol=1e-8, atol=1e-10)
    full = L.periodogram_batched(tt.to(dev).reshape(1, -1), yy.to(dev).reshape(1, -1), ee.to(dev).reshape(1, -1), grid.to(dev))
    assert full.shape == (1, grid.numel()) and abs(float(grid[int(full[0].argmax())]) - 1 / 150.0) < 2 * float(grid[1] - grid[0])
    # batch of different light curves on one grid == singles
    ts, ys, es = [], [], []
    for i in range(4):
        (a, b, c), _ = syn.cfg3_lightcurve(i, n_obs=300)
        ts.append(a.double()); ys.append(b.double()); es.append(c.double())
    T, Y, E = torch.stack(ts).to(dev), torch.stack(ys).to(dev), torch.stack(es).to(dev)
    g2 = torch.linspace(0.001, 0.1, 777, dtype=D).to(dev)
    PB = L.periodogram_batched(T, Y, E, g2)
    for i in range(4):
        assert torch.equal(PB[i], L.periodogram_batched(T[i:i + 1], Y[i:i + 1], E[i:i + 1], g2)[0])
        assert np.allclose(PB[i].cpu().numpy(), lso.power(ts[i].numpy(), ys[i].numpy(), es[i].numpy(), g2.cpu().numpy()), rtol=1e-9, atol=1e-12)
    freqs, pows, grid3 = L.seed_frequencies(T, Y, E, num_peaks=3)
    assert freqs.shape == (4, 3) and np.isfinite(freqs).all()


def test_lomb_scargle_notebook_recorded_peaks(dev, golden_dir):
    """The HIP periodogram on the light curves of the reference's Lomb-Scargle notebook (tests/golden/make_ls_notebook_pin.py):
    equal to the oracle's restatement of astropy's FFT approximation (what ``method='auto'`` resolves to on this grid), and
    the five strongest peaks of band 0 are the five frequencies the notebook recorded, in the recorded order (``fit_LS``'s
    peak rule: ``find_peaks(power, distance=5)`` by decreasing power; with the exact sums the 4th and 5th, whose powers
    differ by 4e-4, swap); the best-band periodogram of the three-band curve peaks at the recorded period/height."""
    from scipy.signal import find_peaks
    from pgmuvi_amd import lombscargle as L
    from oracle import ls_oracle as lso
    p = _load(golden_dir, "ls_notebook_pin.npz")
    t, wl, y, dy = (p["one_" + k] for k in ("t", "wavelength", "y", "dy"))
    m = wl == np.unique(wl)[0]
    assert int(m.sum()) == int(p["nb1d_n_points"])
    ls = L.LombScargle(t[m], y[m], dy[m])
    f = ls.autofrequency(nyquist_factor=5)
    assert len(f) == int(p["nb1d_grid_length"])
    pw = ls.power(f, assume_regular_frequency=True)             # the call of fit_LS: astropy's 'auto' rule -> the FFT approximation
    assert np.allclose(pw, lso.power_fast(t[m], y[m], dy[m], f[0], f[1] - f[0], f.size), rtol=1e-9, atol=1e-11)
    assert np.allclose(ls.power(f, method="slow"), lso.power(t[m], y[m], dy[m], f), rtol=1e-9, atol=1e-12)
    pk_all, _ = find_peaks(pw, distance=5)
    pk_all = pk_all[np.argsort(pw[pk_all])][::-1]
    pk = pk_all[:5]
    got = [round(float(v), 6) for v in f[pk]]
    rec = [round(float(v), 6) for v in p["nb1d_peak_freqs"]]
    assert got == rec, got                                      # all five, in the recorded order
    assert abs(1.0 / f[pk[0]] - float(p["nbmb_best_band"][0])) < 1e-5 and abs(pw[pk[0]] - float(p["nbmb_best_band"][1])) < 2e-6
    # (the recorded significance flags come from the reference's own phase-scramble bootstrap over this periodogram -- a band
    #  selected from a 2-D light curve keeps the multiband code path -- and are covered by tests/test_dropin_reference.py)
    assert float(ls.false_alarm_probability(pw.max(), method="davies")) < 0.05
    # the multiband periodogram (per-band HIP periodograms, weighted as astropy's 'fast' method weights them) equals the
    # oracle's restatement of it ...
    from scipy.signal import peak_prominences
    mb = L.LombScargleMultiband(t, y, wl, dy)
    fm = mb.autofrequency(nyquist_factor=5)
    pm = mb.power(fm, method="fast")
    assert np.allclose(pm, lso.multiband_fast(t, y, wl, dy, fm, sb_auto=True), rtol=1e-9, atol=1e-11)
    assert np.allclose(mb.power(fm, method="fast", sb_method="slow"), lso.multiband_fast(t, y, wl, dy, fm), rtol=1e-9, atol=1e-12)
    # ... and IS the reference's recorded default multiband periodogram (notebook cell 20: peak period 149.170715, height
    # 0.909449, prominence 0.579050) -- the HIP kernel behind the number the reference holds
    pkm, _ = find_peaks(pm)
    km = pkm[np.argmax(pm[pkm])]
    prom = peak_prominences(pm, pkm)[0][np.argmax(pm[pkm])]
    rec_mb = p["nbmb_default"]
    assert abs(1.0 / fm[km] - float(rec_mb[0])) < 1e-5 and abs(pm[km] - float(rec_mb[1])) < 2e-6 and abs(prom - float(rec_mb[2])) < 3e-6, \
        (1.0 / fm[km], pm[km], prom)
    # the two-period light curve with its densely sampled fourth band (cell 34): all eight recorded peaks, in the recorded order
    t2, wl2, y2, dy2 = (p["two_" + k] for k in ("t", "wavelength", "y", "dy"))
    mb2 = L.LombScargleMultiband(t2, y2, wl2, dy2)
    f2 = mb2.autofrequency(nyquist_factor=5)
    p2 = mb2.power(f2, method="fast")
    pk2, _ = find_peaks(p2, distance=5)
    pk2 = pk2[np.argsort(p2[pk2])][::-1][:8]
    assert [round(float(v), 6) for v in f2[pk2]] == [round(float(v), 6) for v in p["nbmb2_peak_freqs"]]


def test_dense_backend_vs_oracle(dev):
    """pgm_mll_dense_f64 / pgm_predict_dense_f64 -- the factorisation sweep on a caller-built
    matrix -- against a torch Cholesky with autograd, ragged and fused-sweep sizes, batched; then the shim's quasi-periodic
    model (ScaleKernel(Periodic * RBF), pgmuvi/gps.py:915-935) end to end: loss, every parameter gradient, prediction."""
    import _oracle_backend as ob
    for n in (5, 130, 700, 1500):
        gen = torch.Generator().manual_seed(n)
        x = torch.sort(torch.rand(n, generator=gen, dtype=D) * 300)[0]
        A = 1.3 * orc.matern(x, x, 20.0, 1.5) + torch.diag(0.02 + 0.05 * torch.rand(n, generator=gen, dtype=D))
        r = torch.randn(n, generator=gen, dtype=D)
        out = _hip.mll_dense(A.to(dev), r.to(dev))
        ref = ob.mll_dense(A, r)
        assert int(out["info"]) == 0 and abs(float(out["mll"]) - float(ref["mll"])) < MLL_TOL
        assert _rel(out["g_r"], ref["g_r"]) < GRAD_RTOL
        ga = out["g_a"].cpu()
        assert torch.equal(ga, ga.T) and _rel(ga, ref["g_a"]) < GRAD_RTOL
        out0 = _hip.mll_dense(A.to(dev), r.to(dev), need_grad=False)
        assert float(out0["mll"]) == float(out["mll"])
    # batch
    n, B = 300, 3
    gen = torch.Generator().manual_seed(1)
    xs = torch.sort(torch.rand(B, n, generator=gen, dtype=D) * 300, dim=1)[0]
    As = torch.stack([orc.rbf(xs[b], xs[b], 15.0) + 0.05 * torch.eye(n, dtype=D) for b in range(B)])
    rs = torch.randn(B, n, generator=gen, dtype=D)
    outb = _hip.mll_dense(As.to(dev), rs.to(dev))
    for b in range(B):
        ref = ob.mll_dense(As[b], rs[b])
        assert abs(float(outb["mll"][b]) - float(ref["mll"])) < MLL_TOL and _rel(outb["g_a"][b], ref["g_a"]) < GRAD_RTOL
    # non-PD is reported, not hidden
    bad = _hip.mll_dense((-torch.eye(6, dtype=D)).to(dev), torch.ones(6, dtype=D).to(dev))
    assert int(bad["info"]) > 0 and math.isnan(float(bad["mll"]))
    # ---- the surface: quasi-periodic model
    t, y, e = syn.cfg2(n_obs=400)
    x, yy, nz = t.double(), y.double(), e.double() ** 2
    K = g.kernels

    def build(device):
        lik = g.likelihoods.FixedNoiseGaussianLikelihood(nz.to(device))

        class M(g.models.ExactGP):
            def __init__(self):
                super().__init__(x.to(device), yy.to(device), lik)
                self.mean_module = g.means.ConstantMean()
                per, rbf = K.PeriodicKernel(), K.RBFKernel()
                per.period_length = 150.0
                rbf.lengthscale = 750.0
                self.covar_module = K.ScaleKernel(K.ProductKernel(per, rbf))

            def forward(self, xx):
                return g.distributions.MultivariateNormal(self.mean_module(xx), self.covar_module(xx))
        return M().double().to(device), lik

    m, lik = build(dev)
    m.train(); lik.train()
    loss = -g.mlls.ExactMarginalLogLikelihood(lik, m)(m(x.to(dev)), yy.to(dev))
    loss.backward()
    raw = {k: p.detach().cpu().clone().requires_grad_(True) for k, p in m.named_parameters()}
    ker = orc.positive(raw["covar_module.raw_outputscale"]) * orc.periodic(
        x, x, orc.positive(raw["covar_module.base_kernel.kernels.0.raw_period_length"]).reshape(()),
        orc.positive(raw["covar_module.base_kernel.kernels.0.raw_lengthscale"]).reshape(())) * orc.rbf(
        x, x, orc.positive(raw["covar_module.base_kernel.kernels.1.raw_lengthscale"]).reshape(()))
    ref = -orc.mll_dense(ker, yy, raw["mean_module.raw_constant"], nz)
    ref.backward()
    assert abs(float(loss.detach()) - float(ref.detach())) < 1e-9
    for k, p in m.named_parameters():
        assert _rel(p.grad.reshape(-1), raw[k].grad.reshape(-1)) < 1e-6, k
    m.eval(); lik.eval()
    xs = torch.linspace(float(x.min()), float(x.max()), 333, dtype=D)
    with torch.no_grad():
        pred = lik(m(xs.to(dev)))
        Kxs = m.covar_module(x.to(dev), xs.to(dev)).to_dense().cpu()
        Kxx = m.covar_module(x.to(dev)).to_dense().cpu()
        kss = torch.diagonal(m.covar_module(xs.to(dev)).to_dense()).cpu()
    pm, pv = orc.posterior_dense(Kxx, Kxs, kss, yy, m.mean_module.constant.detach().cpu(), nz, m.mean_module.constant.detach().cpu())
    assert torch.allclose(pred.mean.cpu(), pm, atol=1e-8) and torch.allclose(pred.variance.cpu(), pv, atol=1e-8)


def _kernel_zoo(d):
    """Composed stationary kernels of the reference's alternative models (pgmuvi/gps.py:915-1342) and the rest of the leaf set."""
    K = g.kernels
    def qp():
        per, rbf = K.PeriodicKernel(), K.RBFKernel()
        per.period_length = 37.0; per.lengthscale = 1.3; rbf.lengthscale = 180.0
        sk = K.ScaleKernel(K.ProductKernel(per, rbf)); sk.outputscale = 0.8
        return sk
    def scaled(base, ls, osc, **kw):
        base.lengthscale = ls
        for k_, v_ in kw.items():
            setattr(base, k_, v_)
        sk = K.ScaleKernel(base); sk.outputscale = osc
        return sk
    zoo = {
        "matern0.5": scaled(K.MaternKernel(nu=0.5), 25.0, 1.1),
        "matern1.5": scaled(K.MaternKernel(nu=1.5), 25.0, 0.9),
        "matern2.5": scaled(K.MaternKernel(nu=2.5), 40.0, 1.3),
        "rbf": scaled(K.RBFKernel(), 30.0, 0.7),
        "rq": scaled(K.RQKernel(), 30.0, 0.7, alpha=1.7),
        "quasi_periodic": qp(),
        "periodic_plus_stochastic": K.AdditiveKernel(qp(), scaled(K.RBFKernel(), 8.0, 0.3)),
    }
    cos = K.CosineKernel(); cos.period_length = 55.0
    zoo["cosine_times_rbf_plus_constant"] = K.AdditiveKernel(K.ProductKernel(cos, scaled(K.RBFKernel(), 90.0, 0.6)), K.ConstantKernel())
    lin = K.LinearKernel(); lin.variance = 2e-5
    zoo["linear_times_matern"] = K.ProductKernel(lin, scaled(K.MaternKernel(nu=1.5), 60.0, 0.5))
    if d == 2:
        tk, wk = scaled(K.MaternKernel(nu=1.5), 20.0, 0.9), scaled(K.RBFKernel(), 0.8, 1.2)
        tk.register_buffer("active_dims", torch.tensor([0])); wk.register_buffer("active_dims", torch.tensor([1]))
        zoo = {"separable": K.ProductKernel(tk, wk), "qp_both_dims": qp(), "rq_2d": scaled(K.RQKernel(), 9.0, 0.8, alpha=0.9)}
        ak, ck = qp(), K.ConstantKernel()
        ak.register_buffer("active_dims", torch.tensor([0])); ck.register_buffer("active_dims", torch.tensor([1]))
        zoo["achromatic"] = K.ProductKernel(ak, ck)
    return {k: v.double() for k, v in zoo.items()}


@pytest.mark.parametrize("d", [1, 2])
def test_generic_kernel_programs_vs_oracle(dev, d):
    """pgm_mll_kernel_value_grad_f64 -- build, sweep and gradient contraction fused for composed stationary kernels -- against
    the oracle's torch formulas with autograd: every leaf kind, sums / products / scales, active_dims in 2-D, ragged sizes,
    the fused sweep with the early inverse pass (N = 1500) and a failed factorisation."""
    import _oracle_backend as ob
    from pgmuvi_amd.gpytorch.kernels import compile_program
    for name, kern in _kernel_zoo(d).items():
        prog = compile_program(kern, d)
        assert prog is not None, name
        theta = prog.theta().detach()
        for n in ((5, 130, 700, 1500) if name in ("quasi_periodic", "separable", "periodic_plus_stochastic") else (300,)):
            gen = torch.Generator().manual_seed(n + d)
            x = torch.rand(n, d, generator=gen, dtype=D) * torch.tensor([300.0, 2.0][:d], dtype=D)
            x = x[torch.argsort(x[:, 0])]
            y = torch.randn(n, generator=gen, dtype=D)
            nz = 0.02 + 0.05 * torch.rand(n, generator=gen, dtype=D)
            mean = torch.full((n,), 0.1, dtype=D)
            out = _hip.mll_kernel_value_grad(x.to(dev), y.to(dev), mean.to(dev), nz.to(dev), torch.tensor(0.01, dtype=D, device=dev), prog, theta.to(dev))
            ref = ob.mll_kernel_value_grad(x, y, mean, nz, torch.tensor(0.01, dtype=D), prog, theta)
            assert int(out["info"]) == 0 and abs(float(out["mll"]) - float(ref["mll"])) < MLL_TOL, (name, n)
            assert _rel(out["g_theta"], ref["g_theta"]) < GRAD_RTOL, (name, n, out["g_theta"], ref["g_theta"])
            assert _rel(out["g_noise"], ref["g_noise"]) < GRAD_RTOL and _rel(out["g_mean"], ref["g_mean"]) < GRAD_RTOL, (name, n)
            val_only = _hip.mll_kernel_value_grad(x.to(dev), y.to(dev), mean.to(dev), nz.to(dev), torch.tensor(0.01, dtype=D, device=dev), prog,
                                                  theta.to(dev), need_grad=False)
            assert float(val_only["mll"]) == float(out["mll"])
    bad = _hip.mll_kernel_value_grad(x.to(dev), y.to(dev), mean.to(dev), torch.full((n,), -5.0, dtype=D, device=dev), None, prog, theta.to(dev))
    assert int(bad["info"]) > 0 and torch.isnan(bad["mll"]) and torch.isnan(bad["g_theta"]).all()


def test_generic_kernel_with_an_underflowing_lengthscale(dev):
    """A length scale so small that the scaled squared distance overflows (u = inf): every off-diagonal entry of the kernel
    matrix is exp(-inf) = 0 -- not NaN -- and the evaluation is that of a diagonal matrix (the unclamped exp is
    for the 1-D spectral-mixture passes only, whose argument is finite by construction)."""
    from pgmuvi_amd.gpytorch import kernels as K
    from pgmuvi_amd.gpytorch.kernels import compile_program
    for base in (K.RBFKernel(), K.MaternKernel(nu=1.5), K.RQKernel()):
        sk = K.ScaleKernel(base.double()).double(); sk.outputscale = 0.8
        prog = compile_program(sk, 1)
        theta = prog.theta().detach().clone()
        theta[0] = 1e-160          # the length scale, straight into the parameter vector (the softplus of the surface cannot produce it)
        theta[-1] = 0.8
        n = 300
        gen = torch.Generator().manual_seed(7)
        x = torch.sort(torch.rand(n, 1, generator=gen, dtype=D) * 300.0, dim=0).values
        y = torch.randn(n, generator=gen, dtype=D)
        nz = 0.02 + 0.05 * torch.rand(n, generator=gen, dtype=D)
        out = _hip.mll_kernel_value_grad(x.to(dev), y.to(dev), torch.zeros(n, dtype=D, device=dev), nz.to(dev), None, prog, theta.to(dev))
        diag = 0.8 + nz
        want = -0.5 * (float((y * y / diag).sum()) + float(torch.log(diag).sum()) + n * np.log(2 * np.pi)) / n
        assert int(out["info"]) == 0 and abs(float(out["mll"]) - want) < 1e-9, (type(base).__name__, float(out["mll"]), want)


def test_generic_kernel_models_through_the_surface(dev):
    """``model(x) -> mll -> backward`` of the reference's quasi-periodic and separable model shapes on the fused generic path:
    loss and every raw-parameter gradient equal the same model evaluated through the dense back-end (the matrix built by
    torch, differentiated by autograd); outside the program's reach (ARD lengthscales) the dense route is taken silently."""
    t, y, e = syn.cfg2(n_obs=600)
    x, yy, nz = t.double().to(dev), y.double().to(dev), (e.double() ** 2).to(dev)
    X4, Y4, E4 = syn.cfg4(n_per_band=60)
    cases = [("quasi_periodic", x, yy, nz, 1), ("periodic_plus_stochastic", x, yy, nz, 1),
             ("separable", X4.double().to(dev), Y4.double().to(dev), (E4.double() ** 2).to(dev), 2)]
    for name, xx, yv, nv, d in cases:
        grads = {}
        for route in ("fused", "dense"):
            kern = _kernel_zoo(d)[name]
            lik = g.likelihoods.FixedNoiseGaussianLikelihood(nv)

            class Model(g.models.ExactGP):
                def __init__(self):
                    super().__init__(xx, yv, lik)
                    self.mean_module = g.means.ConstantMean()
                    self.covar_module = kern

                def forward(self, xq):
                    return g.distributions.MultivariateNormal(self.mean_module(xq), self.covar_module(xq))

            m = Model().double().to(dev)
            m.train(); lik.train()
            mll = g.mlls.ExactMarginalLogLikelihood(lik, m)
            out = m(xx)
            c = out.lazy_covariance_matrix
            assert c.fused
            if route == "dense":
                _ = c.K                                           # materialise: the log-likelihood then takes the dense back-end
                assert not c.fused
            loss = -mll(out, yv)
            loss.backward()
            grads[route] = (float(loss.detach()), {n_: p.grad.detach().clone() for n_, p in m.named_parameters()})
        assert abs(grads["fused"][0] - grads["dense"][0]) < MLL_TOL, name
        for n_ in grads["fused"][1]:
            assert _rel(grads["fused"][1][n_], grads["dense"][1][n_]) < 1e-6, (name, n_)
    ard = g.kernels.ScaleKernel(g.kernels.RBFKernel(ard_num_dims=2)).double().to(dev)
    ard.train()
    assert not ard(X4.double().to(dev)).fused


def test_native_fit_loop_equals_the_host_loop(dev):
    """SURVEY.md section 8f row 2, all on the device (pgm_fit_*): transforms, evaluation, chain rule, optimiser step and log
    as one hipGraph replay per iteration follow the trajectory of the reference-shaped ``train`` (torch optimisers, autograd
    through the shim's constraints) for SGD, Adam and AdamW, with Interval / GreaterThan / Positive constraints, a fixed
    noise vector and a learned scalar noise, 1-D and 2-D."""
    from pgmuvi_amd.trainers import train, train_native
    t, y, e = syn.cfg2(n_obs=300)
    x, yy, nz = t.double().to(dev), y.double().to(dev), (e.double() ** 2).to(dev)
    h = syn.cfg_hypers(2, y.double())

    def build(learn_noise):
        lik = g.likelihoods.GaussianLikelihood().double().to(dev) if learn_noise else g.likelihoods.FixedNoiseGaussianLikelihood(nz)
        m = _make_model(dev, x, yy, lik, 4)
        m.mean_module.register_constraint("raw_constant", g.constraints.Interval(float(yy.min()), float(yy.max())))
        m.covar_module.register_constraint("raw_mixture_means", g.constraints.GreaterThan(1.0 / 3450.0))
        m.initialize(**{"covar_module.mixture_weights": h["w"].to(dev) * 0.7, "covar_module.mixture_means": h["mu"].to(dev) * 1.02,
                        "covar_module.mixture_scales": h["v"].to(dev) * 1.5, "mean_module.constant": torch.tensor(0.1, dtype=D, device=dev)})
        if learn_noise:
            lik.noise = torch.tensor(0.02, dtype=D, device=dev)
        return m, lik

    for optim, learn_noise in (("SGD", False), ("Adam", False), ("AdamW", False), ("AdamW", True)):
        m1, l1 = build(learn_noise); m2, l2 = build(learn_noise)
        r1 = train(model=m1, likelihood=l1, train_x=x, train_y=yy, maxiter=40, lr=0.01, optim=optim, progress=False)
        r2 = train_native(model=m2, likelihood=l2, train_x=x, train_y=yy, maxiter=40, lr=0.01, optim=optim, check_every=16)
        assert len(r2["loss"]) == 40 and len(r2["delta_loss"]) == 39
        assert np.allclose(np.array(r1["loss"], dtype=float), np.array(r2["loss"], dtype=float), rtol=0, atol=1e-9), (optim, learn_noise)
        for (n1, p1), (n2, p2) in zip(m1.named_parameters(), m2.named_parameters()):
            assert n1 == n2 and torch.allclose(p1, p2, rtol=1e-8, atol=1e-10), (optim, n1)
        k = "covar_module.raw_mixture_means"
        assert np.allclose(np.array(r1[k], dtype=float), np.array(r2[k], dtype=float), atol=1e-9)
    # early stop on the reference's rule; the model ends at the stop iteration's parameters
    m3, l3 = build(False); m4, l4 = build(False)
    r3 = train_native(model=m3, likelihood=l3, train_x=x, train_y=yy, maxiter=400, miniter=10, stop=1e-2, lr=1e-5, optim="SGD", check_every=16)
    r4 = train(model=m4, likelihood=l4, train_x=x, train_y=yy, maxiter=400, miniter=10, stop=1e-2, lr=1e-5, optim="SGD", progress=False)
    assert len(r3["loss"]) == len(r4["loss"]) < 400
    for (n1, p1), (n2, p2) in zip(m3.named_parameters(), m4.named_parameters()):
        assert torch.allclose(p1, p2, rtol=1e-8, atol=1e-10), n1
    # 2-D
    X, Y, E = syn.cfg4(n_per_band=40)
    X, Y, NZ = X.double().to(dev), Y.double().to(dev), (E.double() ** 2).to(dev)
    h4 = syn.cfg_hypers(4, Y.cpu())
    def build2():
        lik = g.likelihoods.FixedNoiseGaussianLikelihood(NZ)
        m = _make_model(dev, X, Y, lik, 3, d=2)
        m.initialize(**{"covar_module.mixture_weights": h4["w"].to(dev), "covar_module.mixture_means": h4["mu"].to(dev),
                        "covar_module.mixture_scales": h4["v"].to(dev)})
        return m, lik
    m5, l5 = build2(); m6, l6 = build2()
    r5 = train(model=m5, likelihood=l5, train_x=X, train_y=Y, maxiter=15, lr=0.01, optim="Adam", progress=False)
    r6 = train_native(model=m6, likelihood=l6, train_x=X, train_y=Y, maxiter=15, lr=0.01, optim="Adam")
    assert np.allclose(np.array(r5["loss"], dtype=float), np.array(r6["loss"], dtype=float), rtol=0, atol=1e-9)
    # linear mean (the "1DLinear" / "2DLinear" models of pgmuvi/gps.py:223-267, 321-371): weights and bias are parameters too
    for (xx, yv, nv, dd, Qm) in ((x, yy, nz, 1, 2), (X, Y, NZ, 2, 3)):
        def buildl():
            torch.manual_seed(5)                                  # LinearMean draws its initial weights
            lik = g.likelihoods.FixedNoiseGaussianLikelihood(nv)
            m = _make_model(dev, xx, yv, lik, Qm, d=dd, mean="linear")
            with torch.no_grad():
                m.mean_module.weights.mul_(1e-3); m.mean_module.bias.fill_(0.05)
            hh = syn.cfg_hypers(2 if dd == 1 else 4, yv.cpu())
            m.initialize(**{"covar_module.mixture_weights": hh["w"][:Qm].to(dev), "covar_module.mixture_means": hh["mu"][:Qm].to(dev),
                            "covar_module.mixture_scales": hh["v"][:Qm].to(dev)})
            return m, lik
        m7, l7 = buildl(); m8, l8 = buildl()
        r7 = train(model=m7, likelihood=l7, train_x=xx, train_y=yv, maxiter=12, lr=0.005, optim="AdamW", progress=False)
        r8 = train_native(model=m8, likelihood=l8, train_x=xx, train_y=yv, maxiter=12, lr=0.005, optim="AdamW")
        assert np.allclose(np.array(r7["loss"], dtype=float), np.array(r8["loss"], dtype=float), rtol=0, atol=1e-9), dd
        assert torch.allclose(m7.mean_module.weights, m8.mean_module.weights, rtol=1e-8, atol=1e-12)
        assert torch.allclose(m7.mean_module.bias, m8.mean_module.bias, rtol=1e-8, atol=1e-12)
    # outside its scope it says so
    lik = g.likelihoods.FixedNoiseGaussianLikelihood(nz)
    pm = _make_model(dev, x, yy, lik, 2)
    pm.covar_module.register_prior("mixture_means_prior", g.priors.UniformPrior(0.0, 1.0), "mixture_means")
    with pytest.raises(NotImplementedError):
        train_native(model=pm, likelihood=lik, train_x=x, train_y=yy, maxiter=3)


def test_native_fit_of_short_light_curves_is_one_launch_per_iteration_and_the_host_loops_trajectory(dev, monkeypatch):
    """The device-resident loop below 129 points -- one k_small launch per iteration with the optimiser step inside, 25 iterations
    per graph replay, the log in host-mapped memory -- follows the reference-shaped ``train`` for a 1-D light curve (89 points) and
    a three-band 2-D one (96 points), constant and linear means, a learned scalar noise; and it is the trajectory of the launch
    sequence (PGM_SMALL=0) to 1e-10.  Iteration counts that are no multiple of 25 (whole replays + plain launches)."""
    from pgmuvi_amd.trainers import train, train_native
    t, y, e = syn.cfg2(n_obs=89)
    X2, Y2, E2 = syn.chromatic_sinusoid_2d(32, period=12.5, wavelengths=[0.8, 1.2, 2.2], amplitude_slope=0.5, wl_ref=0.8, noise_level=0.15, t_span=100.0, seed=3)
    h1, h4 = syn.cfg_hypers(2, y.double()), syn.cfg_hypers(4, Y2.double())
    cases = [("1d", t.double().to(dev), y.double().to(dev), (e.double() ** 2).to(dev), 1, 4, h1, "constant", False),
             ("1d-linear-noise", t.double().to(dev), y.double().to(dev), (e.double() ** 2).to(dev), 1, 2, h1, "linear", True),
             ("2d", X2.double().to(dev), Y2.double().to(dev), (E2.double() ** 2).to(dev), 2, 3, h4, "constant", False),
             ("2d-linear", X2.double().to(dev), Y2.double().to(dev), (E2.double() ** 2).to(dev), 2, 3, h4, "linear", False)]
    for name, x, yy, nz, d, Q, h, mean, learn in cases:
        def build():
            torch.manual_seed(5)
            lik = g.likelihoods.GaussianLikelihood().double().to(dev) if learn else g.likelihoods.FixedNoiseGaussianLikelihood(nz)
            m = _make_model(dev, x, yy, lik, Q, d=d, mean=mean)
            if mean == "linear":
                with torch.no_grad():
                    m.mean_module.weights.mul_(1e-3); m.mean_module.bias.fill_(0.05)
            m.initialize(**{"covar_module.mixture_weights": h["w"][:Q].to(dev), "covar_module.mixture_means": h["mu"][:Q].to(dev),
                            "covar_module.mixture_scales": h["v"][:Q].to(dev)})
            if learn:
                lik.noise = torch.tensor(0.02, dtype=D, device=dev)
            return m, lik
        m1, l1 = build(); m2, l2 = build(); m3, l3 = build()
        r1 = train(model=m1, likelihood=l1, train_x=x, train_y=yy, maxiter=63, lr=0.01, optim="AdamW", progress=False)
        _hip.release_workspaces()
        monkeypatch.setenv("PGM_SMALL", "2")          # (the one launch whatever the shape: by default 96 points x 2-D x Q=3 is the launch sequence's)
        r2 = train_native(model=m2, likelihood=l2, train_x=x, train_y=yy, maxiter=63, lr=0.01, optim="AdamW", check_every=40)
        _hip.release_workspaces()
        monkeypatch.setenv("PGM_SMALL", "0")
        r3 = train_native(model=m3, likelihood=l3, train_x=x, train_y=yy, maxiter=63, lr=0.01, optim="AdamW", check_every=40)
        monkeypatch.delenv("PGM_SMALL")
        _hip.release_workspaces()
        assert len(r2["loss"]) == 63
        assert np.allclose(np.array(r1["loss"], dtype=float), np.array(r2["loss"], dtype=float), rtol=0, atol=1e-9), name
        assert np.allclose(np.array(r3["loss"], dtype=float), np.array(r2["loss"], dtype=float), rtol=0, atol=1e-10), name
        for (n1, p1), (n2, p2) in zip(m1.named_parameters(), m2.named_parameters()):
            assert n1 == n2 and torch.allclose(p1, p2, rtol=1e-8, atol=1e-10), (name, n1)


class _LightcurveLike:
    """What ``train(lightcurve=...)`` reads of a ``pgmuvi.lightcurve.Lightcurve`` (``pgmuvi/trainers.py:79-99, 162-166, 193-195``):
    model, likelihood, the transformed data, and ``get_parameters()`` -- constrained values under the names with ``raw_``
    stripped (``pgmuvi/lightcurve.py:8999-9077``, no data transforms)."""
    xtransform = None
    ytransform = None

    def __init__(self, model, likelihood, x, y):
        self.model, self.likelihood, self._xdata_transformed, self._ydata_transformed = model, likelihood, x, y

    def get_parameters(self):
        out = {}
        for name, p in self.model.named_parameters():
            comps = name.split(".")
            mod = self.model
            for c in comps[:-1]:
                mod = getattr(mod, c)
            out[".".join(c.replace("raw_", "") for c in comps)] = getattr(mod, comps[-1].replace("raw_", "")).data
        return out


def test_native_fit_loop_in_lightcurve_mode_logs_every_iteration(dev):
    """``install_native_trainer`` routes ``Lightcurve.fit()`` to the native loop: its ``results`` must have the shape the
    reference's loop produces -- per key of ``get_parameters()`` the initial value plus one entry per iteration (what
    ``plot_results`` / ``to_table`` read) -- with the same values."""
    from pgmuvi_amd.trainers import train, train_native
    t, y, e = syn.cfg2(n_obs=200)
    x, yy, nz = t.double().to(dev), y.double().to(dev), (e.double() ** 2).to(dev)
    h = syn.cfg_hypers(2, y.double())

    def build():
        lik = g.likelihoods.FixedNoiseGaussianLikelihood(nz)
        m = _make_model(dev, x, yy, lik, 4)
        m.mean_module.register_constraint("raw_constant", g.constraints.Interval(float(yy.min()), float(yy.max())))
        m.covar_module.register_constraint("raw_mixture_means", g.constraints.GreaterThan(1.0 / 3450.0))
        m.initialize(**{"covar_module.mixture_weights": h["w"].to(dev), "covar_module.mixture_means": h["mu"].to(dev),
                        "covar_module.mixture_scales": h["v"].to(dev), "mean_module.constant": torch.tensor(0.1, dtype=D, device=dev)})
        return _LightcurveLike(m, lik, x, yy)

    from pgmuvi_amd.trainers import train_device
    a, b, c = build(), build(), build()
    ra = train(lightcurve=a, maxiter=30, lr=0.02, optim="AdamW", progress=False)
    rb = train_native(lightcurve=b, maxiter=30, lr=0.02, optim="AdamW", check_every=8)
    rc = train_device(lightcurve=c, maxiter=30, lr=0.02, optim="AdamW", check_every=8)
    for rx, tol in ((rb, 1e-8), (rc, 2e-5)):          # (train_device steps with torch's capturable AdamW: same trajectory to ~1e-6)
        assert set(ra) == set(rx) and len(rx["loss"]) == 30
        for key in ra:
            if key in ("loss", "delta_loss"):
                continue
            assert len(ra[key]) == len(rx[key]) == 31, key
            for va, vb in zip(ra[key], rx[key]):
                assert va.shape == vb.shape and np.allclose(va, vb, rtol=tol, atol=0.1 * tol + 1e-10), key


def test_native_fit_loop_with_the_default_priors(dev):
    """MAP in the native loop: the priors ``Lightcurve.set_default_priors`` registers (``pgmuvi/lightcurve.py:3273-3322``:
    Normal on the mean constant, LogNormal(0, 1) on mixture means / scales / weights, LogNormal on a learned noise) are
    added on the device before the division by N; same trajectory as the reference-shaped ``train`` through the shim's
    ExactMarginalLogLikelihood, for a fixed-noise and a learned-noise likelihood."""
    from pgmuvi_amd.trainers import train, train_native
    t, y, e = syn.cfg2(n_obs=260)
    x, yy, nz = t.double().to(dev), y.double().to(dev), (e.double() ** 2).to(dev)
    h = syn.cfg_hypers(2, y.double())

    def build(learn_noise):
        lik = g.likelihoods.GaussianLikelihood().double().to(dev) if learn_noise else g.likelihoods.FixedNoiseGaussianLikelihood(nz)
        m = _make_model(dev, x, yy, lik, 3)
        m.covar_module.register_constraint("raw_mixture_means", g.constraints.GreaterThan(1.0 / 3450.0))
        m.initialize(**{"covar_module.mixture_weights": h["w"][:3].to(dev) * 0.8, "covar_module.mixture_means": h["mu"][:3].to(dev) * 1.03,
                        "covar_module.mixture_scales": h["v"][:3].to(dev) * 1.4, "mean_module.constant": torch.tensor(0.05, dtype=D, device=dev)})
        m.mean_module.register_prior("mean_prior", g.priors.NormalPrior(float(yy.mean()), float(yy.std()) / 10), "constant")
        m.covar_module.register_prior("mixture_means_prior", g.priors.LogNormalPrior(0.0, 1.0), "mixture_means")
        m.covar_module.register_prior("mixture_scales_prior", g.priors.LogNormalPrior(0.0, 1.0), "mixture_scales")
        m.covar_module.register_prior("mixture_weights_prior", g.priors.LogNormalPrior(0.0, 1.0), "mixture_weights")
        if learn_noise:
            lik.noise = torch.tensor(0.02, dtype=D, device=dev)
            lik.noise_covar.register_prior("noise_prior", g.priors.LogNormalPrior(float(np.log(0.02)), 0.5), "noise")
        return m.double().to(dev), lik

    for optim, learn_noise in (("Adam", False), ("AdamW", True), ("SGD", True)):
        m1, l1 = build(learn_noise); m2, l2 = build(learn_noise)
        r1 = train(model=m1, likelihood=l1, train_x=x, train_y=yy, maxiter=30, lr=0.01, optim=optim, progress=False)
        r2 = train_native(model=m2, likelihood=l2, train_x=x, train_y=yy, maxiter=30, lr=0.01, optim=optim, check_every=8)
        assert np.allclose(np.array(r1["loss"], dtype=float), np.array(r2["loss"], dtype=float), rtol=0, atol=1e-9), (optim, learn_noise)
        for (n1, p1), (n2, p2) in zip(m1.named_parameters(), m2.named_parameters()):
            assert n1 == n2 and torch.allclose(p1, p2, rtol=1e-8, atol=1e-10), (optim, n1)
    # and the prior terms are really there: the same model without priors follows a different path
    m3, l3 = build(False)
    for mod in (m3.mean_module, m3.covar_module):
        for name in list(mod._priors):
            del mod._priors[name]
    r3 = train_native(model=m3, likelihood=l3, train_x=x, train_y=yy, maxiter=30, lr=0.01, optim="Adam")
    m4, l4 = build(False)
    r4 = train_native(model=m4, likelihood=l4, train_x=x, train_y=yy, maxiter=30, lr=0.01, optim="Adam")
    assert abs(r3["loss"][0] - r4["loss"][0]) > 1e-4


@pytest.mark.parametrize("order", [0, 1])
def test_2d_in_the_fused_sweep_vs_oracle(dev, order):
    """A 2-D light curve of a size that takes the fused sweep (7 block rows), both readings of the product/sum order."""
    X, Y, E = syn.cfg4(n_per_band=100)
    X, Y, nz = X.double(), Y.double(), E.double() ** 2
    h = syn.cfg_hypers(4, Y)
    w, mu, v = h["w"], h["mu"].reshape(-1, 2), h["v"].reshape(-1, 2)
    val, gr = orc.mll_value_grad_closed_form(X, Y, h["mean"], nz, w, mu, v, order, 0.0)
    out = _hip_eval(dev, X, Y, h["mean"], nz, w, mu, v, order)
    assert int(out["info"]) == 0 and abs(float(out["mll"]) - float(val)) < MLL_TOL
    for p in ("w", "mu", "v", "noise", "mean"):
        assert _rel(out[f"g_{p}"].reshape(-1), gr[p].reshape(-1)) < GRAD_RTOL, p


@pytest.mark.parametrize("case", ["1d_2900", "2d_order0", "2d_order1"])
def test_early_inverse_pass_equals_the_plain_schedule(dev, monkeypatch, case):
    """Single light curves in the fused sweep start the inverse pass inside the late diagonal-block launches (spare
    workgroups sum V_pi^T V_pj over finished block rows into R).  Same value and gradients as the plain schedule
    (PGM_EARLY=0: the inverse pass all in its own launch), to rounding of the different summation split, and vs the oracle."""
    if case == "1d_2900":
        gen = torch.Generator().manual_seed(5)
        n = 2900
        x = torch.sort(torch.rand(n, generator=gen, dtype=D) * 900)[0]
        y = torch.randn(n, generator=gen, dtype=D)
        nz = 0.01 + 0.05 * torch.rand(n, generator=gen, dtype=D)
        w = torch.tensor([0.6, 0.3, 0.2], dtype=D); mu = torch.tensor([[0.02], [0.11], [0.3]], dtype=D)
        v = torch.tensor([[0.003], [0.01], [0.02]], dtype=D)
        mean, order = 0.1, 0
    else:
        X, Y, E = syn.cfg4(n_per_band=170)                     # 8 bands x 170 = 1360 points: 11 block rows
        x, y, nz = X.double(), Y.double(), E.double() ** 2
        h = syn.cfg_hypers(4, Y)
        w, mu, v, mean = h["w"], h["mu"].reshape(-1, 2), h["v"].reshape(-1, 2), h["mean"]
        order = 1 if case.endswith("1") else 0
    n, q, d = y.shape[0], w.shape[0], mu.shape[1]
    outs = {}
    for flag in ("0", "1"):
        _hip.release_workspaces()
        monkeypatch.setenv("PGM_EARLY", flag)
        outs[flag] = {k: t.clone() for k, t in _hip_eval(dev, x, y, mean, nz, w, mu, v, order).items() if torch.is_tensor(t)}
        moved = _hip.get_workspace(dev, n, q, d).early_inverse_products()
        assert (moved > 0) == (flag == "1"), moved
    _hip.release_workspaces()
    assert float(outs["0"]["mll"]) == float(outs["1"]["mll"])
    for p in ("w", "mu", "v", "noise", "mean"):
        assert _rel(outs["1"][f"g_{p}"].reshape(-1), outs["0"][f"g_{p}"].reshape(-1)) < 1e-11, p
    val, gr = orc.mll_value_grad_closed_form(x, y, mean, nz, w, mu, v, order, 0.0)
    assert abs(float(outs["1"]["mll"]) - float(val)) < MLL_TOL
    for p in ("w", "mu", "v", "noise", "mean"):
        assert _rel(outs["1"][f"g_{p}"].reshape(-1), gr[p].reshape(-1)) < GRAD_RTOL, p


@pytest.mark.parametrize("n,batch,need_grad", [(2900, 1, True), (1500, 3, True), (2900, 1, False), (640, 1, True)])
def test_look_ahead_is_bit_for_bit_the_three_launch_chain(dev, monkeypatch, n, batch, need_grad):
    """Fused sweep with look-ahead (the row-solve launch forms the next diagonal tile, the head launch leaves the chain;
    PGM_LOOKAHEAD = first block row that does it, default 0) against the three-launch chain (PGM_LOOKAHEAD=99) and a switch in
    mid-sweep: the look-ahead workgroups apply the very operation sequence of a trailing-update tile, so U, V, the value and the
    residual gradients are identical bit for bit; the spectral-mixture gradients only differ by the split of the inverse pass
    (the early products move with the schedule)."""
    gen = torch.Generator().manual_seed(n + batch)
    xs, ys, zs = [], [], []
    for _ in range(batch):
        xs.append(torch.sort(torch.rand(n, generator=gen, dtype=D) * 900)[0]); ys.append(torch.randn(n, generator=gen, dtype=D))
        zs.append(0.01 + 0.05 * torch.rand(n, generator=gen, dtype=D))
    w = torch.tensor([0.6, 0.3, 0.2], dtype=D); mu = torch.tensor([[0.02], [0.11], [0.3]], dtype=D); v = torch.tensor([[0.003], [0.01], [0.02]], dtype=D)
    outs = {}
    for la in ("0", "99", "7"):
        _hip.release_workspaces()
        monkeypatch.setenv("PGM_LOOKAHEAD", la)
        if batch == 1:
            o = _hip_eval(dev, xs[0], ys[0], 0.1, zs[0], w, mu, v, need_grad=need_grad)
        else:
            X, Y, Z = (torch.stack(a).to(dev) for a in (xs, ys, zs))
            o = _hip.mll_value_grad(X.unsqueeze(-1), Y, torch.full_like(Y, 0.1), Z, None, w.to(dev).expand(batch, -1).contiguous(),
                                    mu.to(dev).expand(batch, -1, -1).contiguous(), v.to(dev).expand(batch, -1, -1).contiguous(), 0, 0.0, need_grad)
            torch.cuda.synchronize()
        outs[la] = {k: t.clone().cpu() for k, t in o.items() if torch.is_tensor(t)}
    monkeypatch.delenv("PGM_LOOKAHEAD")
    _hip.release_workspaces()
    for la in ("99", "7"):
        assert torch.equal(outs["0"]["mll"], outs[la]["mll"]), la
        assert torch.equal(outs["0"]["info"], outs[la]["info"])
        if need_grad:
            assert torch.equal(outs["0"]["g_mean"], outs[la]["g_mean"]), la
            for p in ("w", "mu", "v", "noise"):
                assert _rel(outs["0"][f"g_{p}"].reshape(-1), outs[la][f"g_{p}"].reshape(-1)) < 1e-11, (p, la)
    val, gr = orc.mll_value_grad_closed_form(xs[0], ys[0], 0.1, zs[0], w, mu, v, 0, 0.0)
    assert abs(float(outs["0"]["mll"].reshape(-1)[0]) - float(val)) < MLL_TOL
    if need_grad:
        for p in ("w", "mu", "v"):
            assert _rel(outs["0"][f"g_{p}"].reshape(batch, -1)[0], gr[p].reshape(-1)) < GRAD_RTOL, p


@pytest.mark.parametrize("n,batch,need_grad", [(4000, 1, True), (4000, 1, False), (2040, 3, True), (3100, 2, True)])
def test_lazy_plan_of_the_update_tiles_is_bit_for_bit_the_eager_one(dev, monkeypatch, n, batch, need_grad):
    """Fused sweep with more update tiles than CUs: the default plan lets block rows fall two sources behind and take them in
    one pass (only the diagonal tile of the next row is kept up to date on its own), PGM_LAZY=0 updates every row as early as
    it can.  Every tile still receives its sources in ascending order from -C, so the factor, the value and the residual
    gradients agree bit for bit; the spectral-mixture gradients to the rounding of the differently split inverse pass."""
    gen = torch.Generator().manual_seed(7 * n + batch)
    xs, ys, zs = [], [], []
    for _ in range(batch):
        xs.append(torch.sort(torch.rand(n, generator=gen, dtype=D) * 1500)[0]); ys.append(torch.randn(n, generator=gen, dtype=D))
        zs.append(0.01 + 0.05 * torch.rand(n, generator=gen, dtype=D))
    w = torch.tensor([0.6, 0.3, 0.2], dtype=D); mu = torch.tensor([[0.02], [0.11], [0.3]], dtype=D); v = torch.tensor([[0.003], [0.01], [0.02]], dtype=D)
    outs = {}
    for lz in ("1", "0"):
        _hip.release_workspaces()
        monkeypatch.setenv("PGM_LAZY", lz)
        X, Y, Z = (torch.stack(a).to(dev) for a in (xs, ys, zs))
        o = _hip.mll_value_grad(X.unsqueeze(-1), Y, torch.full_like(Y, 0.1), Z, None, w.to(dev).expand(batch, -1).contiguous(),
                                mu.to(dev).expand(batch, -1, -1).contiguous(), v.to(dev).expand(batch, -1, -1).contiguous(), 0, 0.0, need_grad)
        torch.cuda.synchronize()
        outs[lz] = {k: t.clone().cpu() for k, t in o.items() if torch.is_tensor(t)}
    monkeypatch.delenv("PGM_LAZY")
    _hip.release_workspaces()
    assert int(outs["1"]["info"].abs().sum()) == 0 and torch.isfinite(outs["1"]["mll"]).all()
    assert torch.equal(outs["1"]["mll"], outs["0"]["mll"])
    if need_grad:
        assert torch.equal(outs["1"]["g_mean"], outs["0"]["g_mean"])
        for p in ("w", "mu", "v", "noise"):
            assert _rel(outs["1"][f"g_{p}"].reshape(-1), outs["0"][f"g_{p}"].reshape(-1)) < 1e-11, p


@pytest.mark.parametrize("n,batch", [(2048, 8), (2500, 5), (4000, 3)])
def test_windows_for_small_batches_are_bit_for_bit_the_plain_fused_sweep(dev, monkeypatch, n, batch):
    """A handful of light curves of 16 block rows and more run the fused sweep in windows of 8 block rows (the rows beyond a
    window take its eight sources in one deep visit); PGM_BATCH_WINDOW=0 is the plain fused sweep.  Same sources in the same
    order for every tile: value, status and residual gradients bit for bit, the spectral-mixture gradients to rounding."""
    gen = torch.Generator().manual_seed(11 * n + batch)
    X = torch.sort(torch.rand(batch, n, generator=gen, dtype=D) * 1500, dim=1)[0].to(dev)
    Y = torch.randn(batch, n, generator=gen, dtype=D).to(dev)
    Z = (0.01 + 0.05 * torch.rand(batch, n, generator=gen, dtype=D)).to(dev)
    w = torch.tensor([0.6, 0.3, 0.2], dtype=D); mu = torch.tensor([[0.02], [0.11], [0.3]], dtype=D); v = torch.tensor([[0.003], [0.01], [0.02]], dtype=D)
    outs = {}
    for bw in ("-1", "0"):
        _hip.release_workspaces()
        monkeypatch.setenv("PGM_BATCH_WINDOW", bw)
        o = _hip.mll_value_grad(X.unsqueeze(-1), Y, torch.full_like(Y, 0.1), Z, None, w.to(dev).expand(batch, -1).contiguous(),
                                mu.to(dev).expand(batch, -1, -1).contiguous(), v.to(dev).expand(batch, -1, -1).contiguous(), 0, 0.0, True)
        torch.cuda.synchronize()
        outs[bw] = {k: t.clone().cpu() for k, t in o.items() if torch.is_tensor(t)}
    monkeypatch.delenv("PGM_BATCH_WINDOW")
    _hip.release_workspaces()
    assert int(outs["-1"]["info"].abs().sum()) == 0 and torch.isfinite(outs["-1"]["mll"]).all()
    assert torch.equal(outs["-1"]["mll"], outs["0"]["mll"])
    assert torch.equal(outs["-1"]["g_mean"], outs["0"]["g_mean"])
    for p in ("w", "mu", "v", "noise"):
        assert _rel(outs["-1"][f"g_{p}"].reshape(-1), outs["0"][f"g_{p}"].reshape(-1)) < 1e-11, p
    val, gr = orc.mll_value_grad_closed_form(X[0].cpu(), Y[0].cpu(), 0.1, Z[0].cpu(), w, mu, v, 0, 0.0)
    assert abs(float(outs["-1"]["mll"][0]) - float(val)) < MLL_TOL
    for p in ("w", "mu", "v"):
        assert _rel(outs["-1"][f"g_{p}"].reshape(batch, -1)[0], gr[p].reshape(-1)) < GRAD_RTOL, p


@pytest.mark.parametrize("n,batch,need_grad", [(2048, 8, True), (1000, 4, True), (1100, 12, True), (640, 20, True), (2500, 5, False), (3000, 2, True)])
def test_row_solve_of_small_batches_is_bit_for_bit_the_slab_kernels(dev, monkeypatch, n, batch, need_grad):
    """k_trsm64 (round 5: a handful of light curves in the fused sweep -- 128 x 64 slabs, two 16 x 16 blocks per wavefront side by
    side, 6 look-ahead workgroups per light curve, the launch's update sub-tiles as a launch of their own) against the staged
    slab kernel k_trsm (PGM_TRSM64=0): every 16 x 16 block receives the same k-steps in the same order and the forward-
    substitution sums are formed in the slab kernel's order, so EVERY output is the same bit for bit -- one slab per workgroup
    (4 x N=1000, 2 x N=3000), whole blocks (8 x N=2048, 12 x N=1100, 20 x N=640), value only, a last block that ends in padding;
    and the first light curve against the oracle."""
    gen = torch.Generator().manual_seed(7 * n + batch)
    X = torch.sort(torch.rand(batch, n, generator=gen, dtype=D) * 1500, dim=1)[0].to(dev)
    Y = torch.randn(batch, n, generator=gen, dtype=D).to(dev)
    Z = (0.01 + 0.05 * torch.rand(batch, n, generator=gen, dtype=D)).to(dev)
    w = torch.tensor([0.6, 0.3, 0.2], dtype=D); mu = torch.tensor([[0.02], [0.11], [0.3]], dtype=D); v = torch.tensor([[0.003], [0.01], [0.02]], dtype=D)
    outs = {}
    for sw in ("1", "0"):                                     # (1: k_trsm64 whenever its launch fits one round; 0: never)
        _hip.release_workspaces()
        monkeypatch.setenv("PGM_TRSM64", sw)
        o = _hip.mll_value_grad(X.unsqueeze(-1), Y, torch.full_like(Y, 0.1), Z, None, w.to(dev).expand(batch, -1).contiguous(),
                                mu.to(dev).expand(batch, -1, -1).contiguous(), v.to(dev).expand(batch, -1, -1).contiguous(), 0, 0.0, need_grad)
        torch.cuda.synchronize()
        outs[sw] = {k: t.clone().cpu() for k, t in o.items() if torch.is_tensor(t)}
    monkeypatch.delenv("PGM_TRSM64")
    _hip.release_workspaces()
    assert int(outs["1"]["info"].abs().sum()) == 0 and torch.isfinite(outs["1"]["mll"]).all()
    for key in (("mll", "g_w", "g_mu", "g_v", "g_noise", "g_mean") if need_grad else ("mll",)):
        assert torch.equal(outs["1"][key], outs["0"][key]), key
    val, gr = orc.mll_value_grad_closed_form(X[0].cpu(), Y[0].cpu(), 0.1, Z[0].cpu(), w, mu, v, 0, 0.0)
    assert abs(float(outs["1"]["mll"][0]) - float(val)) < MLL_TOL
    if need_grad:
        for p in ("w", "mu", "v"):
            assert _rel(outs["1"][f"g_{p}"].reshape(batch, -1)[0], gr[p].reshape(-1)) < GRAD_RTOL, p


@pytest.mark.parametrize("n,q,batch,d,order", [(1, 1, 1, 1, 0), (2, 2, 1, 1, 0), (17, 2, 1, 1, 0), (89, 2, 1, 1, 0), (127, 4, 1, 1, 0), (128, 4, 1, 1, 0),
                                               (100, 16, 1, 1, 0), (96, 3, 5, 1, 0), (128, 4, 37, 1, 0),
                                               (3, 1, 1, 2, 0), (106, 3, 1, 2, 0), (128, 8, 1, 2, 0), (112, 8, 1, 2, 0), (90, 2, 1, 2, 1), (128, 3, 9, 2, 1), (64, 4, 6, 2, 0)])
def test_one_launch_for_at_most_128_points(dev, monkeypatch, n, q, batch, d, order):
    """Light curves of at most 128 points (the reference's one published workload is N = 89, ``/root/reference/paper/paper.md:113``;
    its Lomb-Scargle notebook's multiband light curve has 106 points in three bands) take ONE launch, k_small -- factors, the
    matrix's sub-blocks built in the registers they are factored in, the inverse, the gradient contraction and the results --
    where every other size takes the launch sequence (PGM_SMALL=0 gives that sequence here too).  Same factor, so the value, z
    and alpha agree to the last bits; the gradient sums are split over 16 x 16 sub-blocks instead of quarter tiles: 1e-12.
    Both against the oracle (1e-9 / 1e-7); value-only; every mixture count up to 16; one and two input dimensions, both
    dimension orders; batches (one workgroup per light curve)."""
    gen = torch.Generator().manual_seed(1000 * n + q + batch + 7 * d + order)
    X = torch.rand(batch, n, d, generator=gen, dtype=D) * 400
    X[:, :, 0] = torch.sort(X[:, :, 0], dim=1)[0]
    if d == 2:
        X[:, :, 1] = torch.randint(1, 4, (batch, n), generator=gen).double() * 0.6      # wavelengths of three bands
    Y = torch.randn(batch, n, generator=gen, dtype=D)
    Z = 0.01 + 0.05 * torch.rand(batch, n, generator=gen, dtype=D)
    W = 0.1 + torch.rand(batch, q, generator=gen, dtype=D)
    MU = 0.005 + 0.2 * torch.rand(batch, q, d, generator=gen, dtype=D)
    V = 0.002 + 0.02 * torch.rand(batch, q, d, generator=gen, dtype=D)
    if d == 2:
        MU[:, :, 1] = 0.3 + 0.4 * torch.rand(batch, q, generator=gen, dtype=D)
        V[:, :, 1] = 0.1 + 0.3 * torch.rand(batch, q, generator=gen, dtype=D)
    ME = 0.3 * torch.randn(batch, 1, generator=gen, dtype=D).expand(batch, n).contiguous()
    args = lambda: (X.to(dev), Y.to(dev), ME.to(dev), Z.to(dev), None, W.to(dev), MU.to(dev), V.to(dev), order, 0.0)
    outs = {}
    for sw in ("2", "0"):                     # (2: one launch whatever the shape -- by default ONE light curve of 113 .. 128 points with
        _hip.release_workspaces()            #  more than four (mixture, dimension) pairs is left to the launch sequence, which is faster there)
        monkeypatch.setenv("PGM_SMALL", sw)
        o = _hip.mll_value_grad(*args(), True)
        o0 = _hip.mll_value_grad(*args(), False)
        torch.cuda.synchronize()
        outs[sw] = {k: t.clone().cpu() for k, t in o.items() if torch.is_tensor(t)}
        assert torch.equal(o0["mll"].cpu(), outs[sw]["mll"])       # value-only: the same arithmetic
    monkeypatch.delenv("PGM_SMALL")
    _hip.release_workspaces()
    a, b_ = outs["2"], outs["0"]
    assert int(a["info"].abs().sum()) == 0 and int(b_["info"].abs().sum()) == 0
    assert float((a["mll"] - b_["mll"]).abs().max()) < 1e-13
    for key in ("g_w", "g_mu", "g_v", "g_noise", "g_mean"):
        assert _rel(a[key].reshape(-1), b_[key].reshape(-1)) < 1e-11, key
    for i in sorted({0, batch - 1}):
        val, gr = orc.mll_value_grad_closed_form(X[i], Y[i], ME[i], Z[i], W[i], MU[i], V[i], order)
        assert abs(float(a["mll"][i]) - float(val)) < MLL_TOL
        for p_ in ("w", "mu", "v", "noise", "mean"):
            assert _rel(a[f"g_{p_}"][i].reshape(-1), gr[p_].reshape(-1)) < GRAD_RTOL, (p_, i)


def test_one_launch_path_reports_a_failed_factorisation_and_learned_noise(dev):
    """k_small: a matrix that is not positive definite (repeated times, no noise, a negative jitter) -> info > 0 and NaN in every
    output, the next evaluation on the same workspace is clean; a learned scalar noise (``noise_scalar`` per light curve, no
    fixed noise) against the oracle."""
    n = 60
    gen = torch.Generator().manual_seed(5)
    x = torch.sort(torch.rand(n, generator=gen, dtype=D) * 100)[0]
    x[10] = x[9]
    y = torch.randn(n, generator=gen, dtype=D)
    w = torch.tensor([0.8], dtype=D); mu = torch.tensor([[0.05]], dtype=D); v = torch.tensor([[0.01]], dtype=D)
    out = _hip.mll_value_grad(x.reshape(n, 1).to(dev), y.to(dev), torch.zeros(n, dtype=D, device=dev), None, None, w.to(dev), mu.to(dev), v.to(dev),
                              0, -1e-3, True)
    torch.cuda.synchronize()
    assert int(out["info"]) > 0 and torch.isnan(out["mll"]).all() and torch.isnan(out["g_w"]).all() and torch.isnan(out["g_noise"]).all()
    ns = torch.tensor(0.07, dtype=D)
    out = _hip.mll_value_grad(x.reshape(n, 1).to(dev), y.to(dev), torch.zeros(n, dtype=D, device=dev), None, ns.to(dev), w.to(dev), mu.to(dev), v.to(dev),
                              0, 0.0, True)
    torch.cuda.synchronize()
    val, gr = orc.mll_value_grad_closed_form(x, y, 0.0, ns, w, mu, v)
    assert int(out["info"]) == 0 and abs(float(out["mll"]) - float(val)) < MLL_TOL
    assert abs(float(out["g_noise"].sum()) - float(gr["noise"])) < GRAD_RTOL * max(1.0, abs(float(gr["noise"])))
    for p_ in ("w", "mu", "v", "mean"):
        assert _rel(out[f"g_{p_}"].reshape(-1), gr[p_].reshape(-1)) < GRAD_RTOL, p_


@pytest.mark.parametrize("n", [130, 255, 383, 640, 897, 1409, 2049, 3970, 5120, 5130])
def test_every_schedule_switch_off_gives_the_same_light_curve(dev, monkeypatch, n):
    """Awkward lengths (one point into a new block, one short of a full one, the last fused size) through the default
    schedule and through the plainest one -- three-launch chain, eager plan of the update tiles (the default lets block rows
    fall two sources behind from 32 block rows on: n = 3970, the last fused size 5120, and inside the windows of 5130), whole-tile inverse pass, no
    early inverse products, the whole matrix built before diagonal block 0 instead of beside it: the same factor, hence the same value bit for bit; gradients to the rounding of their differently split sums; both against the
    oracle where it is quick."""
    gen = torch.Generator().manual_seed(n)
    x = torch.sort(torch.rand(n, generator=gen, dtype=D) * 1200)[0]
    y = torch.randn(n, generator=gen, dtype=D)
    nz = 0.01 + 0.05 * torch.rand(n, generator=gen, dtype=D)
    w = torch.tensor([0.7, 0.25], dtype=D); mu = torch.tensor([[0.013], [0.21]], dtype=D); v = torch.tensor([[0.004], [0.015]], dtype=D)
    outs = {}
    # ("slab8": the default schedule with the chain's row solve on the staged 8-wavefront kernel of rounds 1-3 instead of
    #  k_trsm16 -- every output bit for bit, the left-out MFMAs of the triangular solve add exact zeros)
    for name, env in (("default", {}), ("slab8", {"PGM_TRSM16": "0"}),
                      ("plain", {"PGM_LOOKAHEAD": "99", "PGM_LAUUM_SUB": "0", "PGM_EARLY": "0", "PGM_LAZY": "0", "PGM_BUILD_BESIDE": "0", "PGM_EARLY_T": "0", "PGM_TRSM16": "0", "PGM_PREBUILD": "0"})):
        _hip.release_workspaces()
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        outs[name] = {k: t.clone().cpu() for k, t in _hip_eval(dev, x, y, -0.2, nz, w, mu, v).items() if torch.is_tensor(t)}
        for k_ in env:
            monkeypatch.delenv(k_)
    _hip.release_workspaces()
    assert int(outs["default"]["info"]) == 0 and torch.equal(outs["default"]["mll"], outs["plain"]["mll"])
    assert torch.equal(outs["default"]["g_mean"], outs["plain"]["g_mean"])
    for key in ("mll", "g_w", "g_mu", "g_v", "g_noise", "g_mean"):
        assert torch.equal(outs["default"][key], outs["slab8"][key]), key
    for p in ("w", "mu", "v", "noise"):
        assert _rel(outs["default"][f"g_{p}"].reshape(-1), outs["plain"][f"g_{p}"].reshape(-1)) < 1e-11, p
    if n <= 1409:
        val, gr = orc.mll_value_grad_closed_form(x, y, -0.2, nz, w, mu, v, 0, 0.0)
        assert abs(float(outs["default"]["mll"]) - float(val)) < MLL_TOL
        for p in ("w", "mu", "v", "noise", "mean"):
            assert _rel(outs["default"][f"g_{p}"].reshape(-1), gr[p].reshape(-1)) < GRAD_RTOL, p


@pytest.mark.parametrize("need_grad", [True, False])
def test_batch_schedule_of_round_3_is_bit_for_bit_round_2s(dev, monkeypatch, need_grad):
    """A batch big enough for the panel sweep's round-3 kernels (32 x N=1100: 9 block rows each, 288 in all) -- the strip
    kernel for the row solve (U_kk^-1 resident in LDS, the zero half of the triangular factor skipped) and the left-looking
    updates inside a panel -- against the slab kernel and the one-row-at-a-time in-panel updates of round 2: the MFMAs that
    are left out add exact zeros, every tile still receives its sources in ascending order, the forward-substitution sums keep
    their order, so every output is the same bit for bit; and each light curve's value equals its single evaluation's."""
    B, n = 32, 1100
    xs, ys, ns, ws, mus, vs, means = [], [], [], [], [], [], []
    for i in range(B):
        (t, y, e), per = syn.cfg3_lightcurve(40 + i, n_obs=n)
        h = syn.cfg_hypers(3, y.double(), lead_period=per)
        xs.append(t.double().reshape(n, 1)); ys.append(y.double()); ns.append(e.double() ** 2)
        ws.append(h["w"]); mus.append(h["mu"].reshape(4, 1)); vs.append(h["v"].reshape(4, 1)); means.append(h["mean"].expand(n))
    st = lambda L: torch.stack(L).to(dev)
    args = (st(xs), st(ys), st(means), st(ns), st(ws), st(mus), st(vs))
    keys = ("mll", "g_w", "g_mu", "g_v", "g_noise", "g_mean") if need_grad else ("mll",)
    outs = {}
    for name, env in (("round3", {}), ("round2", {"PGM_STRIPS": "0", "PGM_INLEFT": "0"})):
        _hip.release_workspaces()
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        o = evaluate_batch(*args, need_grad=need_grad)
        torch.cuda.synchronize()
        outs[name] = {k: o[k].clone().cpu() for k in keys + ("info",)}
        for k_ in env:
            monkeypatch.delenv(k_)
    _hip.release_workspaces()
    assert int(outs["round3"]["info"].abs().max()) == 0
    for k in keys:
        assert torch.equal(outs["round3"][k], outs["round2"][k]), k
    for i in (0, 7, 31):
        single = _hip_eval(dev, xs[i], ys[i], means[i], ns[i], ws[i], mus[i], vs[i])
        assert float(single["mll"]) == float(outs["round3"]["mll"][i])
        val, gr = orc.mll_value_grad_closed_form(xs[i], ys[i], means[i], ns[i], ws[i], mus[i], vs[i])
        assert abs(float(val) - float(outs["round3"]["mll"][i])) < MLL_TOL
        if need_grad:
            for p_ in ("w", "mu", "v"):
                assert _rel(outs["round3"][f"g_{p_}"][i].reshape(-1), gr[p_].reshape(-1)) < GRAD_RTOL, p_


def test_windowed_sweep_of_big_single_curves_equals_plain_panels(dev, monkeypatch):
    """41..64 block rows, one light curve: the rows of the current window ride the fused chain (update tiles as fillers of
    the diagonal-block launches), the rows beyond it get one deep update per window.  Same factor, hence the same value bit
    for bit, as the plain panel schedule (PGM_WINDOW=0) and as the early inverse pass switched off; gradients to rounding
    of the different summation split; and the directional derivative along the gradient matches central differences."""
    t, y, e = syn.cfg2(n_obs=5300)                              # 42 block rows
    x, yy, nz = t.double(), y.double(), e.double() ** 2
    h = syn.cfg_hypers(2, yy)
    w, mu, v = h["w"], h["mu"].reshape(4, 1), h["v"].reshape(4, 1)
    outs = {}
    for name, env in (("windowed", {}), ("panels", {"PGM_WINDOW": "0"}), ("plain", {"PGM_WINDOW": "0", "PGM_EARLY": "0"})):
        _hip.release_workspaces()
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        outs[name] = {k: t_.clone() for k, t_ in _hip_eval(dev, x, yy, h["mean"], nz, w, mu, v).items() if torch.is_tensor(t_)}
        for k_ in env:
            monkeypatch.delenv(k_)
    assert int(outs["windowed"]["info"]) == 0
    assert float(outs["windowed"]["mll"]) == float(outs["panels"]["mll"]) == float(outs["plain"]["mll"])
    for p in ("w", "mu", "v", "noise", "mean"):
        for other in ("panels", "plain"):
            assert _rel(outs["windowed"][f"g_{p}"].reshape(-1), outs[other][f"g_{p}"].reshape(-1)) < 1e-10, (p, other)
    _hip.release_workspaces()
    base = outs["windowed"]
    g_ = {p: base[f"g_{p}"].cpu() for p in ("w", "mu", "v")}
    gn = math.sqrt(sum(float((t_ ** 2).sum()) for t_ in g_.values()))
    eps = 1e-6 / gn
    plus = _hip_eval(dev, x, yy, h["mean"], nz, w + eps * g_["w"], mu + eps * g_["mu"], v + eps * g_["v"], need_grad=False)
    minus = _hip_eval(dev, x, yy, h["mean"], nz, w - eps * g_["w"], mu - eps * g_["mu"], v - eps * g_["v"], need_grad=False)
    fd = (float(plus["mll"]) - float(minus["mll"])) / (2 * eps)
    assert abs(fd - gn * gn) < 1e-4 * gn * gn


def test_one_workspace_many_problem_sizes(dev):
    """A workspace serves problems of any size up to its maximum, each with its own captured launch sequence; the tables and
    the scratch of the early inverse pass are shared between them.  Alternating sizes must not disturb earlier graphs."""
    ws = _hip.Workspace(dev, 2304, 2, 1, 1)
    w = torch.tensor([0.6, 0.3], dtype=D); mu = torch.tensor([[0.02], [0.11]], dtype=D); v = torch.tensor([[0.003], [0.01]], dtype=D)

    def problem(n):
        gen = torch.Generator().manual_seed(n)
        x = torch.sort(torch.rand(n, generator=gen, dtype=D) * 700)[0]
        return x, torch.randn(n, generator=gen, dtype=D), 0.01 + 0.05 * torch.rand(n, generator=gen, dtype=D)

    def run(n):
        x, y, nz = problem(n)
        out = _hip.mll_value_grad(x.reshape(n, 1).to(dev), y.to(dev), torch.zeros(n, dtype=D, device=dev), nz.to(dev), None,
                                  w.to(dev), mu.to(dev), v.to(dev), 0, 0.0, True, workspace=ws)
        torch.cuda.synchronize()
        return {k: t.clone() for k, t in out.items() if torch.is_tensor(t)}

    first = {n: run(n) for n in (1100, 2300, 300, 1700)}
    for n in (300, 2300, 1100, 1700, 1100):
        again = run(n)
        for k in ("mll", "g_w", "g_mu", "g_v", "g_noise", "g_mean"):
            assert torch.equal(again[k], first[n][k]), (n, k)
    x, y, nz = problem(1100)
    val, gr = orc.mll_value_grad_closed_form(x, y, 0.0, nz, w, mu, v)
    assert abs(float(first[1100]["mll"]) - float(val)) < MLL_TOL
    assert _rel(first[1100]["g_mu"].reshape(-1), gr["mu"].reshape(-1)) < GRAD_RTOL
    ws.close()


def test_workspace_cache_reuses_covering_workspaces_and_never_frees_live_ones(dev):
    """``_hip.get_workspace`` serves a request from any cached workspace that covers it (alternating shapes, the tail chunk of
    a batch, the dense back-end do not reallocate), and a workspace that something still holds -- here a native fit with its
    captured iteration graph -- survives any number of other shapes passing through the cache."""
    from pgmuvi_amd.trainers import train_native
    _hip.release_workspaces()
    big = _hip.get_workspace(dev, 1500, 4, 1, 4)
    assert _hip.get_workspace(dev, 1500, 4, 1, 4) is big
    assert _hip.get_workspace(dev, 700, 2, 1, 1) is big and _hip.get_workspace(dev, 1536, 1, 1, 3) is big      # covered
    other = _hip.get_workspace(dev, 1537, 1, 1, 1)                                                             # not covered: one more
    assert other is not big and _hip.get_workspace(dev, 1500, 4, 1, 4) is big and len(_hip._workspaces) == 2
    # evaluate_batch with a ragged tail chunk stays on one workspace
    B, n = 10, 300
    xs, ys, ns = [], [], []
    for i in range(B):
        (a, b, c), _ = syn.cfg3_lightcurve(i, n_obs=n)
        xs.append(a.double()); ys.append(b.double()); ns.append(c.double() ** 2)
    x, y, nz = torch.stack(xs).to(dev), torch.stack(ys).to(dev), torch.stack(ns).to(dev)
    h = syn.cfg_hypers(2, ys[0])
    w, mu, v = (h[k].double().to(dev).expand(B, *h[k].shape).contiguous() for k in ("w", "mu", "v"))
    mean = torch.zeros(B, n, dtype=D, device=dev)
    before = set(id(wk) for wk in _hip._workspaces.values())
    out = evaluate_batch(x, y, mean, nz, w, mu, v, chunk=4)
    assert set(id(wk) for wk in _hip._workspaces.values()) == before                   # chunks of 4, 4, 2: no new workspace
    one = _hip.mll_value_grad(x[9], y[9], mean[9], nz[9], None, w[9], mu[9], v[9])
    assert abs(float(out["mll"][9]) - float(one["mll"])) < 1e-12
    # a native fit holds its workspace; squeeze the cache, run other shapes, then continue the fit
    (a, b, c), _ = syn.cfg3_lightcurve(77, n_obs=200)
    lik = g.likelihoods.FixedNoiseGaussianLikelihood((c.double() ** 2).to(dev))
    model = _make_model(dev, a.double().to(dev), b.double().to(dev), lik, 2)
    ref_model = _make_model(dev, a.double().to(dev), b.double().to(dev), lik, 2)
    ref_model.load_state_dict(model.state_dict())
    old_budget = _hip.WORKSPACE_BUDGET_BYTES
    try:
        _hip.WORKSPACE_BUDGET_BYTES = 1                                                   # every miss drops the rest of the cache
        _hip.release_workspaces()
        from pgmuvi_amd import trainers as tr
        fit, _, _ = tr._native_fit_handle(model, lik, a.double().to(dev), b.double().to(dev), maxiter=12, lr=0.05, optim="Adam")
        if True:
            fit.run(6)
            for nn in (900, 130, 2100):                                                  # other shapes: each allocates, the cache drops the fit's
                xx = torch.sort(torch.rand(nn, dtype=D) * 500)[0].to(dev)
                o = _hip.mll_value_grad(xx.reshape(nn, 1), torch.randn(nn, dtype=D, device=dev), torch.zeros(nn, dtype=D, device=dev),
                                        torch.full((nn,), 0.05, dtype=D, device=dev), None, w[0], mu[0], v[0])
                assert int(o["info"]) == 0
            assert fit.ws.handle is not None and all(wk is not fit.ws for wk in _hip._workspaces.values())
            fit.run(6)
            k, losses, _, _, info = fit.read()
            assert k == 12 and info == 0 and np.isfinite(losses).all()
            res = train_native(model=ref_model, likelihood=lik, train_x=a.double().to(dev), train_y=b.double().to(dev), maxiter=12, lr=0.05,
                               optim="Adam", stop=None)
            assert np.allclose(losses, np.asarray([float(t) for t in res["loss"]]), rtol=0, atol=1e-12)
            fit.close()
    finally:
        _hip.WORKSPACE_BUDGET_BYTES = old_budget
        _hip.release_workspaces()


def test_failed_factorisation_returns_nan_gradients(dev):
    """A non-positive pivot leaves NaN in the value AND in every gradient output (never the previous evaluation's numbers)."""
    n = 300
    x = torch.sort(torch.rand(n, dtype=D) * 100)[0].reshape(n, 1).to(dev)
    y = torch.randn(n, dtype=D, device=dev)
    w, mu, v = torch.tensor([1.0], dtype=D, device=dev), torch.tensor([[0.01]], dtype=D, device=dev), torch.tensor([[1e-4]], dtype=D, device=dev)
    zero = torch.zeros(n, dtype=D, device=dev)
    good = _hip.mll_value_grad(x, y, zero, torch.full((n,), 0.1, dtype=D, device=dev), None, w, mu, v)
    assert int(good["info"]) == 0 and torch.isfinite(good["g_noise"]).all()
    bad = _hip.mll_value_grad(x, y, zero, torch.full((n,), -5.0, dtype=D, device=dev), None, w, mu, v)
    assert int(bad["info"]) > 0 and torch.isnan(bad["mll"])
    for k in ("g_w", "g_mu", "g_v", "g_noise", "g_mean"):
        assert torch.isnan(bad[k]).all(), k


def test_host_visible_status_belongs_to_one_evaluation(dev):
    """The factorisation status the host polls for (the last diagonal block stamps it with the evaluation's number) is that of
    the evaluation asked about: right for a good and a failed evaluation, for single light curves of one and of several block
    rows and for a batch with one bad member; withheld (None) once another evaluation has been enqueued on the workspace."""
    gen = torch.Generator().manual_seed(5)
    w, mu, v = torch.tensor([1.0], dtype=D, device=dev), torch.tensor([[0.01]], dtype=D, device=dev), torch.tensor([[1e-4]], dtype=D, device=dev)
    for n in (60, 300, 1500):
        x = torch.sort(torch.rand(n, generator=gen, dtype=D) * 100)[0].reshape(n, 1).to(dev)
        y = torch.randn(n, generator=gen, dtype=D).to(dev)
        zero = torch.zeros(n, dtype=D, device=dev)
        good_noise, bad_noise = torch.full((n,), 0.1, dtype=D, device=dev), torch.full((n,), -5.0, dtype=D, device=dev)
        good = _hip.mll_value_grad(x, y, zero, good_noise, None, w, mu, v)
        ws = good["workspace"]
        assert good["evaluation"] == ws.last_evaluation()
        assert ws.factorisation_failed(1, good["evaluation"]) is False
        bad = _hip.mll_value_grad(x, y, zero, bad_noise, None, w, mu, v, workspace=ws)
        assert bad["evaluation"] == good["evaluation"] + 1
        assert ws.factorisation_failed(1, bad["evaluation"]) is True
        assert ws.factorisation_failed(1, good["evaluation"]) is None          # superseded: the caller reads its own info
        again = _hip.mll_value_grad(x, y, zero, good_noise, None, w, mu, v, workspace=ws)
        assert ws.factorisation_failed(1, bad["evaluation"]) is None and ws.factorisation_failed(1) is False
        torch.cuda.synchronize()
        assert int(good["info"]) == 0 and int(bad["info"]) > 0 and int(again["info"]) == 0
        assert float(again["mll"]) == float(good["mll"])
    # a batch of three, the middle one not positive definite
    n = 200
    x = torch.sort(torch.rand(3, n, generator=gen, dtype=D) * 100, dim=1)[0].reshape(3, n, 1).to(dev)
    y = torch.randn(3, n, generator=gen, dtype=D).to(dev)
    noise = torch.full((3, n), 0.1, dtype=D, device=dev)
    noise[1] = -5.0
    out = _hip.mll_value_grad(x, y, torch.zeros(3, n, dtype=D, device=dev), noise, None, w.expand(3, 1).contiguous(),
                              mu.expand(3, 1, 1).contiguous(), v.expand(3, 1, 1).contiguous())
    assert out["workspace"].factorisation_failed(3, out["evaluation"]) is True
    torch.cuda.synchronize()
    info = out["info"].cpu()
    assert int(info[0]) == 0 and int(info[1]) > 0 and int(info[2]) == 0


def test_multiband_lomb_scargle_vs_oracle(dev):
    """The multiband periodogram the reference's 2-D seeding asks for (``LombScargleMultiband(...).power(f, method='fast')``,
    ``pgmuvi/multiband_ls_significance.py:51-106``): per-band powers by the HIP kernel, combined with astropy's weights (each
    band's summed squared power; pinned by the notebook's recorded cell, ``test_lomb_scargle_notebook_recorded_peaks``); config 4's
    8 bands."""
    from oracle import ls_oracle as lso
    from pgmuvi_amd import lombscargle as L
    X, Y, E = syn.cfg4(n_per_band=120)
    t, bands, y, e = X[:, 0].double().numpy(), X[:, 1].numpy(), Y.double().numpy(), E.double().numpy()
    for dy in (e, None):
        mb = L.LombScargleMultiband(t, y, bands, dy=dy)
        f = mb.autofrequency(nyquist_factor=3)
        p = mb.power(f, method="fast", sb_method="slow")
        assert np.allclose(p, lso.multiband_fast(t, y, bands, dy, f), rtol=1e-9, atol=1e-12)
        assert 0.0 < p.max() <= 1.0
        pa = mb.power(f, method="fast")
        assert np.allclose(pa, lso.multiband_fast(t, y, bands, dy, f, sb_auto=True), rtol=1e-9, atol=1e-11)


def test_c_abi_from_a_c_caller(dev):
    """The ABI exercised from C, not only through ctypes: ``tools/selftest`` (built by ``__graft_entry__.build()``; plain C++ host
    code linking ``libpgmuvi_hip.so``, no torch, no oracle) checks every entry point against a naive CPU implementation of its
    own -- dense kernel, 1-D / 2-D, ragged sizes, batched == single, non-PD reporting -- and exits non-zero on any mismatch."""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "selftest")
    if not os.path.exists(exe):
        pytest.skip("tools/selftest not built (python -c 'import __graft_entry__ as g; g.build()')")
    r = subprocess.run([exe, "1024"], capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert "FAIL" not in r.stdout and r.stdout.count("OK") >= 10


def test_bench_collective_path_on_rccl(dev):
    """``bench.py`` as a one-rank RCCL job on this one-GPU box, started both ways the driver may start the multi-GPU runs:
    plainly (``python bench.py --gpus N``: the self-launcher of ``pgmuvi_amd.launch`` starts the ranks -- ``--spawn`` takes that
    path with N = 1) and under ``python -m torch.distributed.run``.  The process group is RCCL ("nccl"): barrier, the
    all_gather of the log-likelihoods of both modes and the max-over-ranks timing all go through the collective library."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    common = ["--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu"]
    plain = [sys.executable, os.path.join(root, "bench.py"), "--spawn"] + common
    torchrun = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                "--master-port", "29617", os.path.join(root, "bench.py")] + common
    for base, extra, scaling in ((plain, ["--npoints", "1024", "--no-extra"], "weak"),
                                 (torchrun, ["--npoints", "512", "--total-batch", "24", "--chunk", "8"], "strong")):
        r = subprocess.run(base + extra, capture_output=True, text=True, timeout=280, env=env, cwd=root)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-2500:]
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
        out = json.loads(line)
        assert out["scaling"] == scaling and out["n_gpus"] == 1 and out["value"] > 0 and out["roofline"]["frac"] > 0
        assert out["roofline_build"]["frac"] > 0 and out["roofline_build"]["valu_floor_us_at_measured_issue_rate"] > 0


def test_ragged_step_gathers_over_rccl(dev):
    """The ragged batch's collective on RCCL (a one-rank "nccl" group on this one-GPU box): ``make_ragged_shard`` +
    ``sharded_ragged_step`` -- evaluation through the ragged entry point, then ``gather_by_owner``'s all_gather of padded buffers --
    returns the batch's log-likelihoods in the batch's own order, equal to the local evaluation."""
    import datetime
    import torch.distributed as dist
    from pgmuvi_amd.batch import make_ragged_shard, sharded_ragged_step
    assert not dist.is_initialized()
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29641", rank=0, world_size=1, timeout=datetime.timedelta(seconds=120))
    try:
        shard = make_ragged_shard(12, 0, 1, 100, 700, device=dev)
        out, ll = sharded_ragged_step(shard, device=dev)
        torch.cuda.synchronize()
        assert ll.shape == (12,) and ll.is_cuda and torch.equal(ll, out["mll"]) and int(out["info"].abs().max()) == 0
        c, n = shard["curves"][5], shard["lengths"][5]
        single = _hip_eval(dev, c["x"].reshape(n, 1), c["y"], c["mean"], c["noise"], c["w"], c["mu"].reshape(4, 1), c["v"].reshape(4, 1))
        assert float(single["mll"]) == float(ll[5])
    finally:
        dist.destroy_process_group()


def test_performance_guards(dev):
    """Timing guards (1.5 x what one MI355X measures, best of three) for the shapes a change to one schedule can break without
    any parity test noticing: one N=4096 light curve, shards of 2048-point curves, thousands of short curves per call, config 4's size."""
    import time

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
        assert int(out["info"].abs().max()) == 0
        best = float("inf")
        for _ in range(3):                                       # best of three blocks: a busy host must not fail this
            t0 = time.perf_counter()
            for _ in range(reps):
                f()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / reps * 1e3)
        _hip.release_workspaces()
        return best

    # Limits = 1.5 x the best-of-three measured on one MI355X in round 4 (`measured`, ms per call through the Python binding:
    # the C entry point plus ~0.05 ms of wrapper): a schedule regression of the headline, of a config-3 shard or of config 4's
    # size no longer passes silently (round 3's limits were 4 x: 9.0 ms for a 1.9-ms evaluation).
    measured = {(1, 4096): 1.862, (1, 1024): 0.294, (16, 2048): 3.50, (64, 2048): 11.99,
                (1024, 256): 1.30, (2048, 89): 0.703, (1, 8192): 10.82}
    report = []
    for (B, n), ref in measured.items():
        ms = timed(B, n, 5)
        report.append(f"{B} x N={n}: {ms:.3f} ms (guard {1.5 * ref:.3f})")
        assert ms < 1.5 * ref, f"{B} x N={n}: {ms:.2f} ms per call (measured {ref} ms in round 4, guard {1.5 * ref:.2f} ms)"
    # ... and the ragged entry point: 64 light curves with N ~ U{1024..2048} as one trimmed launch set (8.2-8.4 ms measured; the
    # padded sets of round 4's first form take 10.2, the same light curves padded to 2048 points 12.0)
    from pgmuvi_amd.batch import ragged_lengths
    lens = ragged_lengths(64, 1024, 2048)
    padded, lens = pad_curves(_ragged_curves(lens), device=dev)
    f = lambda: evaluate_ragged(padded=padded, lengths=lens)
    out = f(); torch.cuda.synchronize()
    assert int(out["info"].abs().max()) == 0
    best = float("inf")
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 3 * 1e3)
    _hip.release_workspaces()
    report.append(f"ragged 64 x N ~ U{{1024..2048}}: {best:.3f} ms (guard {1.2 * 8.4:.3f})")
    assert best < 1.2 * 8.4, f"ragged 64 x N ~ U{{1024..2048}}: {best:.2f} ms per call (8.4 ms measured; the padded sets take 10.2)"
    print("performance guards: " + "; ".join(report))


@pytest.mark.parametrize("shape", [(1, 89, 2, 1), (1, 250, 3, 2), (3, 300, 3, 1), (1, 1450, 2, 1), (8, 2048, 4, 1), (18, 1450, 2, 1), (12, 100, 2, 1)])
def test_results_do_not_depend_on_what_a_new_workspaces_memory_holds(dev, monkeypatch, shape):
    """A new workspace's buffers hold whatever the allocator hands out: zero pages in a fresh process, the remains of freed
    tensors and workspaces in a long-running one.  ``PGM_POISON=1`` (read by ``pgm_workspace_create``) fills every buffer with
    0xFF bytes -- NaN as doubles, -1 as integers, an unmapped address as a pointer -- so a kernel that reads what nobody wrote
    shows: value and every gradient of the first and of later evaluations on a poisoned workspace must be the bits a normal
    workspace gives (one launch, sixteenth tiles, fused sweep with build-beside and windows, panel sweep with the 64-column row solve)."""
    B, n, q, d = shape
    gen = torch.Generator().manual_seed(B * 1000 + n)
    x = torch.sort(torch.rand(B, n, generator=gen, dtype=D) * 900, dim=1)[0].unsqueeze(-1)
    if d == 2:
        x = torch.cat([x, torch.randint(1, 4, (B, n, 1), generator=gen).double() * 0.5], dim=-1)
    y = torch.randn(B, n, generator=gen, dtype=D)
    nz = 0.01 + 0.05 * torch.rand(B, n, generator=gen, dtype=D)
    w = 0.1 + torch.rand(B, q, generator=gen, dtype=D)
    mu = 0.005 + 0.3 * torch.rand(B, q, d, generator=gen, dtype=D)
    v = 0.001 + 0.02 * torch.rand(B, q, d, generator=gen, dtype=D)
    a = [t.to(dev) for t in (x, y, torch.zeros(B, n, dtype=D), nz)] + [None] + [t.to(dev) for t in (w, mu, v)]
    if B == 1:
        a = [None if t is None else t[0] for t in a]
    _hip.release_workspaces()
    keys = ("mll", "g_w", "g_mu", "g_v", "g_noise", "g_mean", "info")
    plain = _hip.Workspace(dev, n, q, d, B)
    ref = _hip.mll_value_grad(*a, 0, 0.0, True, workspace=plain)
    torch.cuda.synchronize()
    ref = {k: ref[k].clone() for k in keys}
    plain.close()
    assert int(ref["info"].abs().max()) == 0
    monkeypatch.setenv("PGM_POISON", "1")
    poisoned = _hip.Workspace(dev, n, q, d, B)
    monkeypatch.delenv("PGM_POISON")
    for rep in range(3):
        out = _hip.mll_value_grad(*a, 0, 0.0, True, workspace=poisoned)
        torch.cuda.synchronize()
        for k in keys:
            assert torch.equal(out[k], ref[k]), (shape, rep, k)
    # prediction reads the factor, the inverse images and z of that workspace
    if B == 1:
        xs = torch.linspace(-5.0, 905.0, 130, dtype=D).reshape(-1, 1)
        if d == 2:
            xs = torch.cat([xs, torch.full((130, 1), 1.0, dtype=D)], dim=-1)
        pm, pv = _hip.predict(poisoned, xs.to(dev), torch.zeros(130, dtype=D, device=dev))
        torch.cuda.synchronize()
        assert bool(torch.isfinite(pm).all()) and bool(torch.isfinite(pv).all())
    poisoned.close()
