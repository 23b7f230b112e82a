"""Random light curves through the HIP path (C ABI) and the CPU oracle: lengths around every schedule boundary (one tile,
32 tiles = where the build moves beside diagonal block 0, 8 block rows = where the early inverse pass starts, the quarter-tile
inverse pass, ragged last blocks), 1..4 mixture components, 1-D and 2-D inputs in both dimension orders, fixed-noise vector
and/or learned scalar noise, value + every gradient.  PGM_FUZZ_CASES / PGM_FUZZ_SEED widen the run (profiles/r02_fuzz_parity.txt:
168 cases)."""
import os

import pytest
import torch

from pgmuvi_amd import _hip
from oracle import sm_mll_oracle as orc

pytestmark = pytest.mark.gpu
D = torch.float64
SIZES = [17, 89, 127, 128, 129, 255, 257, 383, 511, 640, 897, 1000, 1023, 1024, 1025, 1100, 1153, 1280, 1409, 1537]


def test_random_light_curves_against_the_oracle():
    if not torch.cuda.is_available():
        pytest.skip("-m gpu tests need the MI355X")
    dev = torch.device("cuda:0")
    cases = int(os.environ.get("PGM_FUZZ_CASES", "36"))
    gen = torch.Generator().manual_seed(int(os.environ.get("PGM_FUZZ_SEED", "20261004")))
    worst_v, worst_g, report = 0.0, 0.0, []
    for c in range(cases):
        n = SIZES[int(torch.randint(len(SIZES), (1,), generator=gen))] if c % 3 else int(torch.randint(20, 1500, (1,), generator=gen))
        q = int(torch.randint(1, 5, (1,), generator=gen)); d = 1 if c % 4 else 2
        order = int(torch.randint(2, (1,), generator=gen)) if d == 2 else 0
        x = torch.rand(n, d, generator=gen, dtype=D) * 800.0
        if d == 1:
            x = torch.sort(x[:, 0])[0].reshape(n, 1)
        else:
            x[:, 1] = torch.randint(1, 4, (n,), generator=gen).double() * 0.5
        y = torch.randn(n, generator=gen, dtype=D)
        use_vec = c % 5 != 0
        noise = (0.01 + 0.05 * torch.rand(n, generator=gen, dtype=D)) if use_vec else None
        ns = None if (use_vec and c % 2) else float(0.02 + 0.1 * torch.rand(1, generator=gen))
        w = 0.1 + torch.rand(q, generator=gen, dtype=D)
        mu = 0.005 + 0.3 * torch.rand(q, d, generator=gen, dtype=D)
        v = 0.001 + 0.02 * torch.rand(q, d, generator=gen, dtype=D)
        mean = float(torch.randn(1, generator=gen)) * 0.3
        out = _hip.mll_value_grad(x.to(dev), y.to(dev), torch.full((n,), mean, dtype=D, device=dev), None if noise is None else noise.to(dev),
                                  None if ns is None else torch.tensor(ns, dtype=D, device=dev), w.to(dev), mu.to(dev), v.to(dev), order, 0.0, True)
        torch.cuda.synchronize()
        total = (noise if noise is not None else torch.zeros(n, dtype=D)) + (0.0 if ns is None else ns)
        val, gr = orc.mll_value_grad_closed_form(x if d == 2 else x[:, 0], y, mean, total, w, mu, v, order, 0.0)
        dv = abs(float(out["mll"]) - float(val))
        dg = 0.0
        for p in ("w", "mu", "v", "mean", "noise"):
            a, b = out[f"g_{p}"].detach().cpu().double().reshape(-1), gr[p].reshape(-1).double()
            dg = max(dg, float((a - b).abs().max() / (b.abs().max() + 1e-300)))
        line = (f"case {c:3d}: n={n:5d} q={q} d={d} order={order} noise={'vector' if use_vec else 'none'}{'+scalar' if ns is not None else ''}: "
                f"|d mll| {dv:.2e}  grad rel {dg:.2e}")
        print(line, flush=True)
        if not (int(out["info"]) == 0 and dv < 1e-9 and dg < 1e-7):
            report.append(line)
        worst_v, worst_g = max(worst_v, dv), max(worst_g, dg)
        _hip.release_workspaces()
    print(f"{cases} cases: worst |d mll| {worst_v:.2e}, worst gradient deviation {worst_g:.2e}, {len(report)} outside tolerance")
    assert not report, report


@pytest.mark.parametrize("case", ["duplicate_times", "noise_over_eight_decades", "constant_series", "one_dominant_component"])
def test_collisions_and_extremes_against_the_oracle(case):
    """Inputs the reference's own tests worry about: repeated observation times (two rows of K identical up to the noise),
    noise variances from 1e-6 to 1e2 in one light curve, a constant series, one mixture weight dwarfing the others."""
    if not torch.cuda.is_available():
        pytest.skip("-m gpu tests need the MI355X")
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(len(case))
    n = 700
    x = torch.sort(torch.rand(n, generator=gen, dtype=D) * 400.0)[0]
    y = torch.randn(n, generator=gen, dtype=D)
    noise = 0.01 + 0.05 * torch.rand(n, generator=gen, dtype=D)
    w = torch.tensor([0.6, 0.3, 0.2], dtype=D); mu = torch.tensor([[0.02], [0.11], [0.3]], dtype=D); v = torch.tensor([[0.003], [0.01], [0.02]], dtype=D)
    if case == "duplicate_times":
        x[100:200] = x[300:400]                                 # 100 exact collisions (unsorted on purpose: order is free)
    elif case == "noise_over_eight_decades":
        noise = 10.0 ** (-6.0 + 8.0 * torch.rand(n, generator=gen, dtype=D))
    elif case == "constant_series":
        y = torch.full((n,), 3.25, dtype=D)
    elif case == "one_dominant_component":
        w = torch.tensor([50.0, 1e-6, 1e-3], dtype=D)
    out = _hip.mll_value_grad(x.reshape(n, 1).to(dev), y.to(dev), torch.full((n,), 0.4, dtype=D, device=dev), noise.to(dev), None,
                              w.to(dev), mu.to(dev), v.to(dev), 0, 0.0, True)
    torch.cuda.synchronize()
    val, gr = orc.mll_value_grad_closed_form(x, y, 0.4, noise, w, mu, v, 0, 0.0)
    assert int(out["info"]) == 0
    assert abs(float(out["mll"]) - float(val)) < 1e-9 * max(1.0, abs(float(val)))
    for p in ("w", "mu", "v", "mean", "noise"):
        a, b = out[f"g_{p}"].detach().cpu().double().reshape(-1), gr[p].reshape(-1).double()
        assert float((a - b).abs().max() / (b.abs().max() + 1e-300)) < 1e-7, (case, p)
    _hip.release_workspaces()


def _hyper_dev(got, ref, w, mu, v):
    """Deviation of the hyper-parameter gradient in the metric the optimiser works in: d mll / d log(theta) = theta * d mll / d theta
    (what the softplus-transformed raw parameters see, up to a factor near one), all of (w, mu, v) as ONE vector, largest entry as
    the scale.  Component by component relative to itself is not a test of the kernels: with one mixture at 0.2 cycles/day on white
    noise over 1160 points the frequency derivative is the rest of a 3e9-fold cancellation (sum of |terms| 1.5e7, gradient -4.7e-3:
    ragged fuzz seed 77, call 117) and the oracle's own two routes agree on it to 2e-8 only, while d mll / d log v is 0.37."""
    def vec(g):
        parts = []
        for p, th in (("w", w), ("mu", mu), ("v", v)):
            try:
                t = g[f"g_{p}"]                     # (the binding's outputs)
            except KeyError:
                t = g[p]                            # (the oracle's dictionary)
            parts.append(t.detach().cpu().double().reshape(-1) * th.detach().cpu().double().reshape(-1))
        return torch.cat(parts)
    a, r = vec(got), vec(ref)
    return float((a - r).abs().max() / (r.abs().max() + 1e-300))


def test_random_ragged_batches_against_their_single_evaluations():
    """Random ragged batches through ``pgm_mll_value_grad_ragged_f64``: 12 .. 70 light curves per call (trimmed launch sets of the
    panel sweep, and a few calls below the threshold: padded sets), lengths from 20 to 1400 in different mixes (uniform, many
    equal ones, a few long stragglers), 1 .. 4 mixture components, 1-D and 2-D inputs in both dimension orders, noise vector
    and / or one learned scalar per light curve, with and without the gradient.  Every value must be the light curve's own
    single evaluation bit for bit, every gradient equal to it to rounding; one member per call also goes to the CPU oracle."""
    if not torch.cuda.is_available():
        pytest.skip("-m gpu tests need the MI355X")
    dev = torch.device("cuda:0")
    cases = int(os.environ.get("PGM_FUZZ_RAGGED_CASES", "8"))
    gen = torch.Generator().manual_seed(int(os.environ.get("PGM_FUZZ_SEED", "20261004")) + 1)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=gen))
    for c in range(cases):
        B = (ri(28, 70) if c % 3 == 1 else ri(12, 70)) if c % 4 else ri(3, 11)     # (2-D cases: enough members for a trimmed set)
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
        out = _hip.mll_value_grad_ragged(x.to(dev), y.to(dev), mean.to(dev), nz.to(dev) if use_vec else None, ns.to(dev) if use_scalar else None,
                                         lengths, w.to(dev), mu.to(dev), v.to(dev), order, 0.0, need_grad)
        torch.cuda.synchronize()
        res = {k: t.clone() for k, t in out.items() if torch.is_tensor(t)}
        assert int(res["info"].abs().max()) == 0, (c, res["info"])
        set_of, nbs = _hip.ragged_plan(lengths, B)
        worst = 0.0
        pick = ri(0, B - 1)
        for b, n in enumerate(lengths):
            single = _hip.mll_value_grad(x[b, :n].to(dev), y[b, :n].to(dev), mean[b, :n].to(dev), nz[b, :n].to(dev) if use_vec else None,
                                         ns[b].to(dev) if use_scalar else None, w[b].to(dev), mu[b].to(dev), v[b].to(dev), order, 0.0, need_grad)
            torch.cuda.synchronize()
            assert float(single["mll"]) == float(res["mll"][b]), (c, b, n, float(single["mll"]), float(res["mll"][b]))
            if need_grad:
                worst = max(worst, _hyper_dev({f"g_{p}": res[f"g_{p}"][b] for p in ("w", "mu", "v")}, single, w[b], mu[b], v[b]))
                for p in ("noise", "mean"):
                    a, r = res[f"g_{p}"][b, :n], single[f"g_{p}"].reshape(-1)
                    worst = max(worst, float((a - r).abs().max() / (r.abs().max() + 1e-300)))
                    assert float(res[f"g_{p}"][b, n:].abs().sum()) == 0.0
            if b == pick:
                total = (nz[b, :n] if use_vec else torch.zeros(n, dtype=D)) + (float(ns[b]) if use_scalar else 0.0)
                val, gr = orc.mll_value_grad_closed_form(x[b, :n] if d == 2 else x[b, :n, 0], y[b, :n], float(mean[b, 0]), total, w[b], mu[b], v[b], order, 0.0)
                assert abs(float(val) - float(res["mll"][b])) < 1e-9, (c, b, n)
                if need_grad:
                    assert _hyper_dev({f"g_{p}": res[f"g_{p}"][b] for p in ("w", "mu", "v")}, gr, w[b], mu[b], v[b]) < 1e-7, (c, b, n)
        print(f"ragged case {c}: {B} light curves, lengths {min(lengths)}..{max(lengths)}, q={q} d={d} order={order} "
              f"noise={'vector' if use_vec else ''}{'+scalar' if use_scalar else ''} grad={need_grad}: sets {nbs}, worst gradient deviation from the singles {worst:.2e}", flush=True)
        assert worst < 1e-10, (c, worst)
        _hip.release_workspaces()


def test_one_launch_value_is_the_launch_sequences_bit_for_bit(monkeypatch):
    """The one launch (k_small, forced for every shape: PGM_SMALL=2) against the launch sequence (PGM_SMALL=0) on random light curves
    of 1 .. 128 points, 1 .. 8 mixtures, one and two input dimensions in both orders, vector and / or scalar noise: the VALUE as
    bits -- a light curve must not change its value with the path a call's other members, or a cost rule, send it down (a member of
    a ragged launch set takes the sequence, the same light curve alone the one launch) -- and the gradients to the rounding of their
    differently split sums.  (Round 6: the compiler had contracted k_small's z^2 products into the first adds of its value sums,
    which k_finalize keeps apart; one light curve in several hundred differed in the last bit.  `tools/lab/ragged_repro.py 31 17 13`.)"""
    if not torch.cuda.is_available():
        pytest.skip("-m gpu tests need the MI355X")
    dev = torch.device("cuda:0")
    cases = int(os.environ.get("PGM_FUZZ_SMALL_CASES", "3000"))
    gen = torch.Generator().manual_seed(int(os.environ.get("PGM_FUZZ_SEED", "20261004")) + 2)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=gen))
    _hip.release_workspaces()
    tables = {"2": {}, "0": {}}

    def ws_of(sw, q, d):
        if (q, d) not in tables[sw]:
            monkeypatch.setenv("PGM_SMALL", sw)          # (read when a workspace is made)
            tables[sw][(q, d)] = _hip.Workspace(dev, 128, q, d, 1)
        return tables[sw][(q, d)]

    differ, worst = [], 0.0
    for c in range(cases):
        n = ri(1, 128) if c % 4 else ri(97, 128)
        d = 1 + (c % 3 == 1); q = ri(1, 8 if d == 1 else 6); order = ri(0, 1) if d == 2 else 0
        x = torch.rand(n, d, generator=gen, dtype=D) * 600.0
        if d == 1:
            x = torch.sort(x[:, 0])[0].reshape(n, 1)
        else:
            x[:, 1] = torch.randint(1, 4, (n,), generator=gen).double() * 0.5
        y = torch.randn(n, generator=gen, dtype=D)
        nz = 0.01 + 0.05 * torch.rand(n, generator=gen, dtype=D)
        use_vec = c % 5 != 4
        ns = None if (use_vec and c % 2) else 0.02 + 0.1 * torch.rand((), generator=gen, dtype=D)
        w = 0.1 + torch.rand(q, generator=gen, dtype=D)
        mu = 0.005 + 0.3 * torch.rand(q, d, generator=gen, dtype=D)
        v = 0.001 + 0.02 * torch.rand(q, d, generator=gen, dtype=D)
        mean = torch.full((n,), float(torch.randn((), generator=gen, dtype=D)) * 0.3, dtype=D)
        a = (x.to(dev), y.to(dev), mean.to(dev), nz.to(dev) if use_vec else None, None if ns is None else ns.to(dev), w.to(dev), mu.to(dev),
             v.to(dev), order, 0.0, True)
        one = _hip.mll_value_grad(*a, workspace=ws_of("2", q, d))
        seq = _hip.mll_value_grad(*a, workspace=ws_of("0", q, d))
        torch.cuda.synchronize()
        assert int(one["info"]) == 0 and int(seq["info"]) == 0, (c, n, q, d)
        if float(one["mll"]) != float(seq["mll"]):
            differ.append((c, n, q, d, order, float(one["mll"]), float(seq["mll"])))
        for k in ("g_w", "g_mu", "g_v", "g_mean", "g_noise"):
            worst = max(worst, float((one[k] - seq[k]).abs().max()) / (float(seq[k].abs().max()) + 1e-300))
    for table in tables.values():
        for ws in table.values():
            ws.close()
    monkeypatch.delenv("PGM_SMALL", raising=False)
    _hip.release_workspaces()
    print(f"{cases} light curves: {len(differ)} values differ in their bits; worst gradient deviation {worst:.2e}")
    assert not differ, differ[:5]
    assert worst < 1e-9


def test_default_schedule_value_is_the_plainest_schedules_bit_for_bit(monkeypatch):
    """Random light curves of 129 .. 1600 points (1-D and 2-D, 1 .. 4 mixtures) through the default schedule -- factors and matrix
    in one launch (k_prebuild) or the matrix beside diagonal block 0, split build tiles, look-ahead, lazy plan, early inverse
    products, quarter / sixteenth tiles in the inverse pass -- and through the plainest one (every switch off: k_precompute +
    k_build, the three-launch chain, whole tiles): separately compiled kernels evaluating the same expressions must give the
    same factor, hence the same VALUE as bits; gradients to rounding."""
    if not torch.cuda.is_available():
        pytest.skip("-m gpu tests need the MI355X")
    dev = torch.device("cuda:0")
    cases = int(os.environ.get("PGM_FUZZ_SWITCH_CASES", "240"))
    gen = torch.Generator().manual_seed(int(os.environ.get("PGM_FUZZ_SEED", "20261004")) + 3)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=gen))
    plain = {"PGM_LOOKAHEAD": "99", "PGM_LAUUM_SUB": "0", "PGM_EARLY": "0", "PGM_LAZY": "0", "PGM_BUILD_BESIDE": "0", "PGM_EARLY_T": "0",
             "PGM_TRSM16": "0", "PGM_PREBUILD": "0"}
    _hip.release_workspaces()
    tables = {"default": {}, "plain": {}}

    def ws_of(name, q, d):
        if (q, d) not in tables[name]:
            for k_, v_ in plain.items():
                if name == "plain":
                    monkeypatch.setenv(k_, v_)
                else:
                    monkeypatch.delenv(k_, raising=False)
            tables[name][(q, d)] = _hip.Workspace(dev, 1664, q, d, 1)
            for k_ in plain:
                monkeypatch.delenv(k_, raising=False)
        return tables[name][(q, d)]

    differ, worst = [], 0.0
    for c in range(cases):
        n = ri(129, 1600) if c % 3 else ri(129, 520)
        d = 1 + (c % 4 == 1); q = ri(1, 4); order = ri(0, 1) if d == 2 else 0
        x = torch.rand(n, d, generator=gen, dtype=D) * 900.0
        if d == 1:
            x = torch.sort(x[:, 0])[0].reshape(n, 1)
        else:
            x[:, 1] = torch.randint(1, 4, (n,), generator=gen).double() * 0.5
        y = torch.randn(n, generator=gen, dtype=D)
        nz = 0.01 + 0.05 * torch.rand(n, generator=gen, dtype=D)
        w = 0.1 + torch.rand(q, generator=gen, dtype=D)
        mu = 0.005 + 0.3 * torch.rand(q, d, generator=gen, dtype=D)
        v = 0.001 + 0.02 * torch.rand(q, d, generator=gen, dtype=D)
        mean = torch.full((n,), 0.1, dtype=D)
        a = (x.to(dev), y.to(dev), mean.to(dev), nz.to(dev), None, w.to(dev), mu.to(dev), v.to(dev), order, 0.0, True)
        one = _hip.mll_value_grad(*a, workspace=ws_of("default", q, d))
        two = _hip.mll_value_grad(*a, workspace=ws_of("plain", q, d))
        torch.cuda.synchronize()
        assert int(one["info"]) == 0 and int(two["info"]) == 0, (c, n, q, d)
        if float(one["mll"]) != float(two["mll"]) or not torch.equal(one["g_mean"], two["g_mean"]):
            differ.append((c, n, q, d, order, float(one["mll"]), float(two["mll"])))
        for k in ("g_w", "g_mu", "g_v", "g_noise"):
            worst = max(worst, float((one[k] - two[k]).abs().max()) / (float(two[k].abs().max()) + 1e-300))
    for table in tables.values():
        for ws in table.values():
            ws.close()
    _hip.release_workspaces()
    print(f"{cases} light curves: {len(differ)} values (or alpha vectors) differ in their bits; worst gradient deviation {worst:.2e}")
    assert not differ, differ[:5]
    assert worst < 1e-9
