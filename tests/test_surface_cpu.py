"""CPU tests of the host logic: the GPyTorch-shaped operator surface, the autograd node,
the trainer mirror and the no-fallback rule.  Where an evaluation is needed the HIP call
is replaced -- in the test only -- by the oracle stand-in ``tests/_oracle_backend.py``."""
import math
import os
import re
import warnings
from unittest import mock

import numpy as np
import pytest
import torch

import _oracle_backend as ob
from oracle import sm_mll_oracle as orc
from pgmuvi_amd import _hip, gpytorch as g, synthetic as syn
from pgmuvi_amd.trainers import train

D = torch.float64
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model(x, y, lik, Q=4, d=1, mean="constant"):
    class Model(g.models.ExactGP):
        def __init__(self):
            super().__init__(x, y, lik)
            self.mean_module = g.means.ConstantMean() if mean == "constant" else g.means.LinearMean(input_size=d)
            self.covar_module = g.kernels.SpectralMixtureKernel(num_mixtures=Q, ard_num_dims=d)
            self.sci_kernel = self.covar_module

        def forward(self, xx):
            return g.distributions.MultivariateNormal(self.mean_module(xx), self.covar_module(xx))

    return Model().double()


@pytest.fixture()
def small():
    t, y, e = syn.cfg2(n_obs=48)
    return t.double(), y.double(), e.double() ** 2


def test_product_has_no_cpu_fallback_and_never_touches_the_oracle(small):
    x, y, noise = small
    lik = g.likelihoods.FixedNoiseGaussianLikelihood(noise)
    m = _model(x, y, lik)
    m.train()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        g.mlls.ExactMarginalLogLikelihood(lik, m)(m(x), y)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.covar_module(x).to_dense()
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|['\"]oracle['\"]|oracle/", re.M)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "pgmuvi_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".inc")):
                assert not pat.search(open(os.path.join(dirpath, f)).read()), f"{f} references oracle/"


def test_missing_library_fails_loudly(tmp_path):
    with mock.patch.object(_hip, "_lib", None), mock.patch.object(_hip, "_LIB_PATH", str(tmp_path / "nope.so")):
        with pytest.raises(_hip.HipLibraryMissing):
            _hip.load()


def test_parameter_names_shapes_and_constraints(small):
    x, y, noise = small
    lik = g.likelihoods.GaussianLikelihood()
    m = _model(x, y, lik)
    names = [n for n, _ in m.named_parameters()]
    assert names == ["likelihood.noise_covar.raw_noise", "mean_module.raw_constant", "covar_module.raw_mixture_weights",
                     "covar_module.raw_mixture_means", "covar_module.raw_mixture_scales"]      # pgmuvi/trainers.py:173
    k = m.covar_module
    assert k.raw_mixture_weights.shape == (4,) and k.raw_mixture_means.shape == (4, 1, 1) and k.num_mixtures == 4
    assert abs(float(k.mixture_weights[0]) - math.log(2.0)) < 1e-12          # softplus(0) = 0.6931 (reference notebook)
    assert float(lik.noise) > 1e-4                                           # GreaterThan(1e-4) default
    m.mean_module.register_constraint("raw_constant", g.constraints.Interval(-2.0, 3.0))
    assert "raw_constant_constraint" in m.mean_module._constraints           # reference tests/test_constraint_sets.py:95
    assert {n for n, _ in m.named_constraints()} >= {"mean_module.raw_constant_constraint", "covar_module.raw_mixture_means_constraint"}
    c = m.mean_module._constraints["raw_constant_constraint"]
    c.lower_bound = torch.tensor(-5.0)                                        # pgmuvi mutates bounds (lightcurve.py:3140)
    assert float(c.lower_bound) == -5.0 and -5.0 < float(m.mean_module.constant) < 3.0
    with pytest.raises(RuntimeError, match="nonexistent"):
        m.mean_module.register_constraint("raw_nothing", g.constraints.Positive())
    with pytest.raises(ValueError):
        g.constraints.Interval(1.0, 0.0)
    with pytest.raises(ValueError):
        g.constraints.Interval(0.0, math.inf)


def test_initialize_dotted_names_and_bounds(small):
    x, y, noise = small
    m = _model(x, y, g.likelihoods.FixedNoiseGaussianLikelihood(noise))
    h = syn.cfg_hypers(2, y)
    m.initialize(**{"covar_module.mixture_means": h["mu"], "covar_module.mixture_scales": h["v"],
                    "covar_module.mixture_weights": h["w"], "mean_module.constant": 0.25})        # lightcurve.py:4156
    assert torch.allclose(m.covar_module.mixture_means, h["mu"], rtol=1e-12)
    assert abs(float(m.mean_module.constant) - 0.25) < 1e-12
    m.covar_module.register_constraint("raw_mixture_means", g.constraints.GreaterThan(1.0 / 3450.0))
    m.initialize(**{"covar_module.mixture_means": h["mu"]})
    assert torch.allclose(m.covar_module.mixture_means, h["mu"], rtol=1e-6)
    with pytest.raises(AttributeError):
        m.initialize(**{"covar_module.no_such_thing": 1.0})
    m.covar_module.initialize_from_data(x, y)                                                    # gps.py:209
    assert torch.all(m.covar_module.mixture_means > 0) and torch.all(torch.isfinite(m.covar_module.mixture_scales))
    assert torch.allclose(m.covar_module.mixture_weights, (y.std() / 4).expand(4))


def test_out_of_scope_names_import_but_refuse_to_run():
    from pgmuvi_amd.gpytorch.kernels import GridInterpolationKernel, ScaleKernel, RBFKernel, MaternKernel  # noqa: F401
    from pgmuvi_amd.gpytorch.variational import CholeskyVariationalDistribution, VariationalStrategy      # noqa: F401
    with pytest.raises(NotImplementedError):
        GridInterpolationKernel(RBFKernel(), grid_size=10)
    with pytest.raises(NotImplementedError):
        g.models.ApproximateGP(None)
    for ctx in (g.settings.max_cg_iterations(10000), g.settings.fast_pred_var(), g.settings.fast_computations(False, False, False)):
        with ctx:
            pass
    with g.settings.fast_computations(False, False, False):
        assert g.settings.fast_computations.log_prob.off()
    assert g.settings.fast_computations.log_prob.on()


def test_mll_backward_train_loop_with_oracle_standin(small):
    x, y, noise = small
    with mock.patch.object(_hip, "mll_value_grad", ob.mll_value_grad):
        lik = g.likelihoods.FixedNoiseGaussianLikelihood(noise)
        m = _model(x, y, lik)
        h = syn.cfg_hypers(2, y)
        m.initialize(**{"covar_module.mixture_means": h["mu"], "covar_module.mixture_scales": h["v"],
                        "covar_module.mixture_weights": h["w"], "mean_module.constant": h["mean"]})
        m.train(); lik.train()
        mll = g.mlls.ExactMarginalLogLikelihood(lik, m)
        assert isinstance(mll, g.mlls.marginal_log_likelihood.MarginalLogLikelihood)
        loss = -mll(m(x), y)
        assert loss.dim() == 0
        loss.backward()
        ref = orc.mll(x, y, h["mean"], noise, h["w"], h["mu"].reshape(4, 1), h["v"].reshape(4, 1))
        assert abs(float(loss.detach()) + float(ref)) < 1e-12
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
        with pytest.raises(RuntimeError, match="train on the training inputs"):
            m(x + 1.0)
        res = train(model=m, likelihood=lik, train_x=x, train_y=y, maxiter=12, miniter=2, stop=1e-9, lr=0.01,
                    optim="AdamW", stopavg=3, progress=False)
        assert set(res) >= {"loss", "delta_loss", "covar_module.raw_mixture_means", "raw_mixture_means"}
        assert len(res["loss"]) == 12 and len(res["delta_loss"]) == 11 and res["loss"][-1] < res["loss"][0]
        res2 = train(model=m, likelihood=lik, train_x=x, train_y=y, maxiter=50, miniter=2, stop=1e3, lr=1e-6,
                     optim="SGD", stopavg=3, progress=False)
        assert len(res2["loss"]) == 4                          # early stop: first i > miniter with std(loss[-3:]) < stop
    for bad in (dict(optim="LBFGS"), dict(lossfn="nope"), dict(optim="NUTS")):
        with pytest.raises((ValueError, NotImplementedError)):
            train(model=m, likelihood=lik, train_x=x, train_y=y, maxiter=1, progress=False, **bad)


def test_jitter_retry_policy_with_oracle_standin():
    """psd_safe_cholesky semantics: retry with 1e-8, 1e-7, 1e-6 (fp64), warn each time, then NotPSDError."""
    x = torch.linspace(0, 1, 12, dtype=D)
    y = torch.zeros(12, dtype=D)
    calls = []

    def flaky(*a, **k):
        jit = a[9] if len(a) > 9 else k.get("jitter", 0.0)
        calls.append(jit)
        out = ob.mll_value_grad(*a, **k)
        if jit < 5e-8:
            out["info"] = torch.ones_like(out["info"])
        return out

    lik = g.likelihoods.FixedNoiseGaussianLikelihood(torch.full((12,), 0.1, dtype=D))
    m = _model(x, y, lik, Q=1)
    m.train()
    mll = g.mlls.ExactMarginalLogLikelihood(lik, m)
    with mock.patch.object(_hip, "mll_value_grad", flaky), warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        val = mll(m(x), y)
    assert calls == [0.0, 1e-8, 1e-7] and torch.isfinite(val)
    assert sum("added jitter" in str(w.message) for w in rec) == 2
    always = lambda *a, **k: {**ob.mll_value_grad(*a, **k), "info": torch.tensor(3, dtype=torch.int32)}
    with mock.patch.object(_hip, "mll_value_grad", always), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with pytest.raises(g.utils.errors.NotPSDError):
            mll(m(x), y)


def test_fixed_noise_likelihood_semantics():
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        lik = g.likelihoods.FixedNoiseGaussianLikelihood(torch.tensor([1e-9, 0.1], dtype=D))
    assert any("Very small noise" in str(w.message) for w in rec) and float(lik.noise[0]) == 1e-6
    lik2 = g.likelihoods.FixedNoiseGaussianLikelihood(torch.full((5,), 0.1), learn_additional_noise=True)
    assert [n for n, _ in lik2.named_parameters()] == ["second_noise_covar.raw_noise"]
    with pytest.raises(RuntimeError):
        g.likelihoods.FixedNoiseGaussianLikelihood(torch.full((5,), 0.1)).second_noise = 0.1
    with pytest.raises(RuntimeError):
        g.models.ExactGP(torch.zeros(3), torch.zeros(3), likelihood=torch.nn.Identity())


def test_native_fit_host_side(small):
    """The constraint descriptors handed to pgm_fit_create (kinds and bounds exactly as the shim's transforms use them) and
    the scope checks of train_native; the loop itself needs the GPU (tests/test_gpu_parity.py)."""
    from pgmuvi_amd.trainers import _constraint_descriptor, train_native
    x, y, noise = small
    lik = g.likelihoods.FixedNoiseGaussianLikelihood(noise)
    m = _model(x, y, lik, Q=2)
    m.mean_module.register_constraint("raw_constant", g.constraints.Interval(-1.5, 2.5))
    m.covar_module.register_constraint("raw_mixture_means", g.constraints.GreaterThan(1.0 / 3450.0))
    m.covar_module.register_constraint("raw_mixture_scales", g.constraints.LessThan(0.5))
    assert _constraint_descriptor(m.mean_module, "raw_constant", 1) == [(3, -1.5, 4.0)]
    k1, a1, b1 = _constraint_descriptor(m.covar_module, "raw_mixture_means", 2)[1]
    assert k1 == 1 and a1 == float(torch.tensor(1.0 / 3450.0).float()) and b1 == 0.0          # float32 bound, as stored
    assert _constraint_descriptor(m.covar_module, "raw_mixture_scales", 2)[0] == (2, 0.5, 0.0)
    assert _constraint_descriptor(m.covar_module, "raw_mixture_weights", 2) == [(1, 0.0, 0.0)] * 2        # Positive
    # prior tables for pgm_fit_set_priors: plain Normal / LogNormal priors on the loop's parameters, in raw-vector order
    from pgmuvi_amd.trainers import _prior_descriptor
    glik = g.likelihoods.GaussianLikelihood()
    pri = _model(x, y, glik, Q=2)
    pieces = [(pri.mean_module, "raw_constant"), (pri.covar_module, "raw_mixture_weights"), (pri.covar_module, "raw_mixture_means"),
              (pri.covar_module, "raw_mixture_scales"), (glik.noise_covar, "raw_noise")]
    assert _prior_descriptor(pri, glik, pieces, glik.noise_covar) is None
    pri.mean_module.register_prior("mean_prior", g.priors.NormalPrior(0.3, 0.1), "constant")
    pri.covar_module.register_prior("mixture_means_prior", g.priors.LogNormalPrior(0.0, 1.0), "mixture_means")
    glik.register_prior("noise_prior", g.priors.LogNormalPrior(-4.0, 0.5), "noise")       # on the likelihood, as pgmuvi/test_script.py:235
    kind, loc, scale = _prior_descriptor(pri, glik, pieces, glik.noise_covar)
    assert kind == [1, 0, 0, 2, 2, 0, 0, 2]
    assert loc[0] == pytest.approx(0.3) and scale[0] == pytest.approx(0.1) and loc[7] == -4.0 and scale[7] == 0.5 and loc[3:5] == [0.0, 0.0]
    pri.covar_module.register_prior("w_prior", g.priors.UniformPrior(0.0, 1.0), "mixture_weights")
    with pytest.raises(NotImplementedError):                     # other prior families: left to train_device
        _prior_descriptor(pri, glik, pieces, glik.noise_covar)
    with pytest.raises(NotImplementedError):
        train_native(model=pri, likelihood=glik, train_x=x, train_y=y, maxiter=2)
    fx = _model(x, y, lik, Q=2)
    lik.register_prior("noise_prior", g.priors.LogNormalPrior(-4.0, 0.5), "noise")         # a prior on a FIXED noise: not a loop parameter
    with pytest.raises(NotImplementedError):
        train_native(model=fx, likelihood=lik, train_x=x, train_y=y, maxiter=2)
    del lik._priors["noise_prior"]
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        train_native(model=m, likelihood=lik, train_x=x, train_y=y, maxiter=2)
    with pytest.raises(ValueError):
        train_native(model=m, likelihood=lik)


def test_train_picks_the_fused_optimiser_step_only_for_gpu_parameters():
    """``train()`` makes torch's fused Adam/AdamW when every parameter sits on the GPU (one launch per step: the loop is
    host-bound at the sizes pgmuvi's users have); on the CPU, for SGD, and for the captured loop (capturable) it is
    torch's default."""
    from pgmuvi_amd import trainers
    seen = {}

    class Fake:
        def __init__(self, cuda):
            self.is_cuda = cuda

    def spy(params, lr, eps, **kw):
        seen["kw"] = kw
        return "optimiser"

    with mock.patch.dict(trainers._OPTIMISERS, {"AdamW": spy, "Adam": spy, "SGD": spy}), \
            mock.patch.object(torch, "is_floating_point", lambda p: True):
        assert trainers._optimiser("AdamW", [Fake(True), Fake(True)], 1e-3, 1e-8) == "optimiser"
        assert seen["kw"] == {"fused": True}
        trainers._optimiser("Adam", [Fake(True), Fake(False)], 1e-3, 1e-8)
        assert seen["kw"] == {}
        trainers._optimiser("AdamW", [Fake(False)], 1e-3, 1e-8)
        assert seen["kw"] == {}
        trainers._optimiser("SGD", [Fake(True)], 1e-3, 1e-8)
        assert seen["kw"] == {}
        trainers._optimiser("AdamW", [Fake(True)], 1e-3, 1e-8, capturable=True)
        assert seen["kw"] == {"capturable": True}
    p = torch.nn.Parameter(torch.zeros(3, dtype=torch.float64))
    opt = trainers._optimiser("AdamW", [p], 1e-3, 1e-8)      # (CPU parameters: torch's default form, and it steps)
    p.grad = torch.ones(3, dtype=torch.float64)
    opt.step()
    assert float(p[0]) < 0.0


def test_launcher_counts_gpus_without_the_hip_runtime(tmp_path):
    """``launch.visible_gpu_count``: topology nodes with SIMDs whose render node this process may open, cut down by the
    *_VISIBLE_DEVICES lists -- no torch, no HIP call in the parent of ``bench.py --gpus N``."""
    from pgmuvi_amd import launch
    nodes, dri = tmp_path / "nodes", tmp_path / "dri"
    for i, (simd, minor) in enumerate([(0, -1), (1024, 128), (1024, 129), (1024, 130)]):
        (nodes / str(i)).mkdir(parents=True)
        (nodes / str(i) / "properties").write_text(f"cpu_cores_count 0\nsimd_count {simd}\ndrm_render_minor {minor}\n")
    dri.mkdir()
    for m in (128, 129):
        (dri / f"renderD{m}").write_text("")
    assert launch._kfd_gpu_nodes(str(nodes), str(dri)) == 2                       # (renderD130 is not ours)
    assert launch._kfd_gpu_nodes(str(nodes), str(tmp_path / "none")) == 3         # (no /dev/dri to check against)
    assert launch._kfd_gpu_nodes(str(tmp_path / "missing")) is None
    assert launch.visible_gpu_count({}, str(tmp_path / "missing")) is None
    assert launch.visible_gpu_count({"HIP_VISIBLE_DEVICES": "0,2"}, str(tmp_path / "missing")) == 2
    assert launch.visible_gpu_count({"HIP_VISIBLE_DEVICES": ""}, str(tmp_path / "missing")) == 0
    assert launch.visible_gpu_count({"ROCR_VISIBLE_DEVICES": "0,1,2", "HIP_VISIBLE_DEVICES": "1"}, str(tmp_path / "missing")) == 1
    assert launch.visible_gpu_count({"CUDA_VISIBLE_DEVICES": "0,-1,2"}, str(tmp_path / "missing")) == 1
    import ast, inspect
    imported = {a.name.split(".")[0] for node in ast.walk(ast.parse(inspect.getsource(launch))) if isinstance(node, (ast.Import, ast.ImportFrom))
                for a in (node.names if isinstance(node, ast.Import) else [ast.alias(name=node.module or "")])}
    assert "torch" not in imported


def test_deferred_check_keeps_a_failure_that_had_to_leave():
    """``settings.defer_cholesky_check``: more than LIMIT evaluations between two queries (an LBFGS closure with a line search)
    must not lose a failed factorisation among the evicted ones."""
    from pgmuvi_amd import mll_function as mf
    mf.drop_deferred()
    ok = {"info": torch.zeros(1, dtype=torch.int32)}
    bad = {"info": torch.tensor([3], dtype=torch.int32)}
    mf._defer(bad)
    for _ in range(mf._Deferred.LIMIT + 3):
        mf._defer(dict(ok))
    assert len(mf._deferred.items) == mf._Deferred.LIMIT
    assert mf.take_deferred_failure() is True
    assert mf.take_deferred_failure() is False                # (the flag is consumed with the query)
    for _ in range(mf._Deferred.LIMIT + 3):
        mf._defer(dict(ok))
    assert mf.take_deferred_failure() is False
    mf._defer(bad)
    for _ in range(mf._Deferred.LIMIT + 1):
        mf._defer(dict(ok))
    mf.drop_deferred()
    assert mf.take_deferred_failure() is False


def test_release_workspaces_reaches_the_workspaces_outside_the_cache():
    """``batch._two_streams`` keeps a pair of workspaces of its own; ``_hip.release_workspaces()`` hands them back too, and
    their bytes are weighed against the cache budget when they are made."""
    from pgmuvi_amd import _hip, batch
    batch._twin_workspaces[("fake",)] = [object(), object(), []]
    _hip.release_workspaces()
    assert not batch._twin_workspaces
    est = _hip.workspace_bytes_estimate(2048, 4, 1, 3)
    assert 3 * 8 * 2048 * 2048 <= est <= 3 * 8 * 2048 * 2048 * 1.2
