"""CPU tests of the NUTS/HMC layer.

* the sampler itself on analytic targets (no GP, no device);
* the GP potential (``pgmuvi_amd.mcmc.SMPotential``) against the oracle's restatement with autograd, the one HIP
  call replaced by the oracle stand-in (test only; the GPU tests hold HIP == oracle);
* chains sharded over two ``gloo`` ranks == the same chains in one process.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import sm_mll_oracle as orc  # noqa: E402
from pgmuvi_amd import mcmc, synthetic as syn  # noqa: E402

D = torch.float64


def _gauss_target(P=5, seed=0):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((P, P))
    scal = np.array([0.1, 1.0, 3.0, 10.0, 0.5])[:P]
    S = (A @ A.T + 0.5 * np.eye(P)) * np.outer(scal, scal)
    Si, m = np.linalg.inv(S), np.arange(P) * 1.0

    def pot(z):                                    # row by row: a chain's numbers must not depend on its batch mates
        G = np.stack([Si @ (row - m) for row in z])
        return 0.5 * ((z - m) * G).sum(-1), G

    return pot, m, S


def test_nuts_recovers_a_correlated_gaussian():
    pot, m, S = _gauss_target()
    r = mcmc.sample(pot, np.random.default_rng(5).standard_normal((4, 5)), num_samples=1200, warmup_steps=400, seed=1)
    x = r["samples"]
    assert x.shape == (4, 1200, 5)
    sd = np.sqrt(np.diag(S))
    assert np.all(np.abs(x.mean((0, 1)) - m) < 0.25 * sd)
    assert np.all(np.abs(x.reshape(-1, 5).var(0) / np.diag(S) - 1) < 0.2)
    assert np.all(mcmc.split_rhat(x) < 1.05)
    assert r["stats"]["divergent"].sum() == 0
    assert 0.7 < r["stats"]["accept_prob"].mean() <= 1.0
    # the diagonal metric adapted towards the marginal variances (10^4 dynamic range)
    assert np.all(np.abs(np.log(r["inverse_mass"].mean(0) / np.diag(S))) < 1.0)
    # trees double: every recorded trajectory has 2^depth - 1 (or, on a rejected last doubling, fewer than 2^(depth+1)) steps
    n, dep = r["stats"]["n_leapfrog"], r["stats"]["depth"]
    assert np.all(n >= 2 ** dep - 1) and np.all(n < 2 ** (dep + 1))


def test_hmc_and_reproducibility_and_batch_independence():
    pot, m, S = _gauss_target()
    z0 = np.random.default_rng(2).standard_normal((3, 5))
    a = mcmc.sample(pot, z0, num_samples=300, warmup_steps=150, seed=3, sampler="HMC", trajectory_length=3.0)
    assert np.all(np.abs(a["samples"].mean((0, 1)) - m) < 0.6 * np.sqrt(np.diag(S)))
    # same seed -> same draws; a chain's stream depends on (seed, chain id) only, not on its batch mates
    b = mcmc.sample(pot, z0, num_samples=50, warmup_steps=30, seed=9)
    c = mcmc.sample(pot, z0, num_samples=50, warmup_steps=30, seed=9)
    assert np.array_equal(b["samples"], c["samples"])
    solo = mcmc.sample(pot, z0[1:2], num_samples=50, warmup_steps=30, seed=9, chain_ids=[1])
    assert np.array_equal(solo["samples"][0], b["samples"][1])
    with pytest.raises(ValueError):
        mcmc.sample(pot, z0, sampler="Gibbs")
    # metric initialised from the curvature at the start: exact for a Gaussian with the diagonal of the precision
    pot2, m2, S2 = _gauss_target()
    d = mcmc.sample(pot2, m2[None, :] + 0.0, num_samples=5, warmup_steps=0, seed=1, init_metric="curvature")
    assert np.allclose(d["inverse_mass"][0], 1.0 / np.diag(np.linalg.inv(S2)), rtol=1e-6)


def test_divergent_or_infinite_potential_is_rejected_not_propagated():
    def pot(z):                                    # standard normal inside |z|<3, wall outside
        U = 0.5 * (z ** 2).sum(-1)
        bad = np.abs(z).max(-1) > 3.0
        return np.where(bad, np.inf, U), np.where(bad[:, None], 0.0, z)
    r = mcmc.sample(pot, np.zeros((2, 2)), num_samples=300, warmup_steps=100, seed=0)
    assert np.isfinite(r["samples"]).all() and np.abs(r["samples"]).max() <= 3.0
    assert np.isfinite(r["stats"]["potential_energy"]).all()


def _lightcurves(C, n):
    xs, ys, ns = [], [], []
    for c in range(C):
        (t, y, e), _per = syn.cfg3_lightcurve(5000 + c, n_obs=n)
        xs.append(t.double().reshape(n, 1)); ys.append(y.double()); ns.append(e.double() ** 2)
    return torch.stack(xs), torch.stack(ys), torch.stack(ns)


@pytest.mark.parametrize("learn_noise", [False, True])
def test_potential_and_gradient_equal_the_oracle(learn_noise):
    import _oracle_backend as ob
    C, n, Q = 3, 48, 2
    x, y, nz = _lightcurves(C, n)
    pot = mcmc.SMPotential(x, y, None if learn_noise else nz, num_mixtures=Q, compute=ob.mll_value_grad)
    assert pot.P == 1 + 3 * Q + (1 if learn_noise else 0)
    assert [s.name for s in pot.sites][:4] == ["mean_module.mean_prior", "covar_module.mixture_weights_prior",
                                               "covar_module.mixture_means_prior", "covar_module.mixture_scales_prior"]
    rng = np.random.default_rng(4)
    z = rng.normal(0, 0.5, (C, pot.P))
    z[:, 1 + Q:1 + 2 * Q] += np.log(1 / 150.0)
    z[:, 1 + 2 * Q:1 + 3 * Q] += np.log(1 / 1500.0)
    if learn_noise:
        z[:, -1] = np.log(0.01) + rng.normal(0, 0.3, C)
    U, G = pot(z)
    for c in range(C):
        zt = torch.tensor(z[c], dtype=D, requires_grad=True)
        s = 1e-4 * y[c].std()
        ref = orc.nuts_potential(zt, x[c], y[c], None if learn_noise else nz[c], Q, 1, y[c].mean(), y[c].std() / 10,
                                 noise_loc=torch.log(s), noise_scale=s)
        ref.backward()
        assert abs(U[c] - float(ref.detach())) < 1e-8 * max(1.0, abs(float(ref.detach())))
        assert np.allclose(G[c], zt.grad.numpy(), rtol=1e-8, atol=1e-8)
    # constrain/unconstrain round trip and pyro-style site shapes
    vals = pot.constrain(z)
    assert vals["covar_module.mixture_means_prior"].shape == (C, Q, 1, 1) and vals["mean_module.mean_prior"].shape == (C,)
    assert np.allclose(pot.unconstrain(vals), z)


def test_short_nuts_run_on_a_small_light_curve_finds_the_period():
    import _oracle_backend as ob
    t, y, e = syn.cfg1()
    sel = slice(0, 256, 4)
    x, yy, nz = t.double()[sel], y.double()[sel], e.double()[sel] ** 2
    init = {"mean_module.mean_prior": np.array(float(yy.mean())), "covar_module.mixture_weights_prior": np.array([0.5]),
            "covar_module.mixture_means_prior": np.array([1 / 140.0]).reshape(1, 1, 1),
            "covar_module.mixture_scales_prior": np.array([1 / 1500.0]).reshape(1, 1, 1)}
    out = mcmc.run_mcmc(x, yy, nz, num_mixtures=1, num_samples=60, warmup_steps=60, num_chains=2, seed=0,
                        initial_values=init, compute=ob.mll_value_grad, group_by_chain=True, max_tree_depth=6)
    f = out["covar_module.mixture_means_prior"]
    assert f.shape == (2, 60, 1, 1, 1)
    # cfg 1 is a P=150 sinusoid; 64 points over 2.2 periods and the LogNormal(0,1) prior on the frequency
    # (mode at 1 cycle/day) leave a posterior of about 141 +- 15 days
    assert 120.0 < np.median(1.0 / f) < 165.0
    dg = out["_diagnostics"]
    assert dg["accept_prob"].mean() > 0.5 and dg["divergent"].mean() < 0.2
    flat = mcmc.run_mcmc(x, yy, nz, num_mixtures=1, num_samples=5, warmup_steps=5, num_chains=2, seed=0, initial_values=init,
                         compute=ob.mll_value_grad, max_tree_depth=4)
    assert flat["covar_module.mixture_weights_prior"].shape == (10, 1)


def _worker(rank, world, rendezvous, q):
    import _oracle_backend as ob
    dist.init_process_group("gloo", init_method="file://" + rendezvous, rank=rank, world_size=world)
    x, y, nz = _lightcurves(3, 32)
    out = mcmc.run_mcmc(x, y, nz, num_mixtures=1, num_samples=6, warmup_steps=6, seed=11, compute=ob.mll_value_grad,
                        group_by_chain=True, max_tree_depth=3, initial_values=_INIT)
    q.put((rank, out["covar_module.mixture_means_prior"], out["_diagnostics"]["n_leapfrog"]))
    dist.barrier()
    dist.destroy_process_group()


_INIT = {"mean_module.mean_prior": np.array(0.0), "covar_module.mixture_weights_prior": np.array([0.5]),
         "covar_module.mixture_means_prior": np.array([1 / 150.0]).reshape(1, 1, 1),
         "covar_module.mixture_scales_prior": np.array([1 / 1500.0]).reshape(1, 1, 1)}


def test_chains_sharded_over_two_gloo_ranks_equal_one_process(tmp_path):
    """Config 5's layout: each chain has its own light curve, chains are block-partitioned over the ranks, one
    all_gather of the draws at the end; a chain's draws do not depend on where it ran."""
    import _oracle_backend as ob
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    rendezvous = str(tmp_path / "rendezvous")             # a file of this test's own: no port to lose between probe and bind
    procs = [ctx.Process(target=_worker, args=(r, 2, rendezvous, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    x, y, nz = _lightcurves(3, 32)
    one = mcmc.run_mcmc(x, y, nz, num_mixtures=1, num_samples=6, warmup_steps=6, seed=11, compute=ob.mll_value_grad,
                        group_by_chain=True, max_tree_depth=3, initial_values=_INIT)
    for rank, f, nl in got:
        assert f.shape == (3, 6, 1, 1, 1)
        assert np.allclose(f, one["covar_module.mixture_means_prior"], rtol=1e-12, atol=0)
        assert np.array_equal(nl, one["_diagnostics"]["n_leapfrog"])


def test_effective_sample_size_on_known_chains():
    """Independent draws count (almost) fully, an AR(1) chain with coefficient 0.9 counts (1 - 0.9) / (1 + 0.9) of its length."""
    import numpy as np
    from pgmuvi_amd import mcmc
    rng = np.random.default_rng(0)
    iid = rng.standard_normal((4, 1000, 2))
    ess = mcmc.effective_sample_size(iid)
    assert ess.shape == (2,) and (ess > 2500).all() and (ess < 6000).all()
    z = np.zeros((4, 1000))
    for c in range(4):
        for t in range(1, 1000):
            z[c, t] = 0.9 * z[c, t - 1] + rng.standard_normal()
    ar = float(mcmc.effective_sample_size(z[:, :, None])[0])
    assert 120 < ar < 350, ar                                     # 4000 * 0.0526 = 210
