#!/usr/bin/env python3
"""bench.py -- marginal-log-likelihood evaluations/sec of the MI355X hot path.

    python bench.py --gpus N --steps K --warmup W [--batch B]                      # headline: weak scaling
    python bench.py --gpus N --steps K --warmup W --total-batch B [--npoints 2048] # strong scaling of one B-light-curve batch
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Started without a launcher and with ``--gpus N`` > 1 the process is only a parent: before anything touches a GPU it starts N
fresh children of its own command line, one rank per GPU (``pgmuvi_amd.launch.spawn_ranks``: RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* as torch.distributed.run sets them), relays rank 0's JSON line and exits with the children's worst status; with fewer
than N visible devices it exits non-zero with a one-line reason.

A *step* is one pass of the hot path -- SM kernel build + blocked Cholesky MLL + full
hyper-parameter gradient (pgmuvi/trainers.py:179-181 without the optimiser step) -- over
one batch of B synthetic config-2 light curves per GPU (N=4096 points, Q=4 mixtures,
fp64; default B=1 = BASELINE.json configs[1], "Single 1D Lightcurve").  Inputs are
resident in HBM before the timed region.  With N>1 GPUs every rank evaluates its own
light curves (weak scaling, no data-path collective) and one RCCL all_gather of the
log-likelihoods closes each step.  Rank 0 prints ONE JSON line.

``--total-batch B`` switches to STRONG scaling (north_star: "a 4096-lightcurve batch"; BASELINE configs[2]: 512 x N=2048):
a step is one evaluation of the WHOLE batch of B config-3 light curves -- block-partitioned over the ranks
(``pgmuvi_amd.batch.shard_bounds``), each shard in memory-bounded chunks through the batched entry point, ONE all_gather of
the B log-likelihoods -- and ``value`` = B * steps / elapsed.  The default invocation also appends a short strong-scaling
measurement (``strong_scaling``: configs[2] and a 4096 x N=4096 batch) to the weak-scaling line, so that the driver's own
N=1,2,4,8 runs record both.
"""
import argparse
import glob
import hashlib
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from pgmuvi_amd import _hip, launch, synthetic as syn  # noqa: E402
from pgmuvi_amd.batch import default_chunk, gather_logliks, make_shard, sharded_batch_step, shard_bounds  # noqa: E402

PMC_GLOB = "r*_pmc_hbm_traffic*.json"   # tools/pmc_traffic.py output; only a file measured on the CURRENT library is quoted
FP64_MATRIX_PEAK_TFLOPS = 78.6   # MI355X datasheet "FP64 matrix" (the guides list no fp64 MFMA figure)
NB = 128


def make_batch(n, B, rank, dev):
    xs, ys, ns, ws, mus, vs, ms = [], [], [], [], [], [], []
    for b in range(B):
        t, y, e = syn.cfg2(n_obs=n, seed=2 + 1000 * rank + b)        # rank 0, b 0 == the golden cfg-2 light curve
        h = syn.cfg_hypers(2, y.double())
        xs.append(t.double().reshape(n, 1)); ys.append(y.double()); ns.append(e.double() ** 2)
        ws.append(h["w"]); mus.append(h["mu"].reshape(-1, 1)); vs.append(h["v"].reshape(-1, 1)); ms.append(h["mean"].expand(n))
    st = lambda L: torch.stack(L).to(dev).contiguous()
    return dict(x=st(xs), y=st(ys), mean=st(ms), noise=st(ns), w=st(ws), mu=st(mus), v=st(vs))


def update_flops(n):
    """Algorithmic flops of the trailing_update launches of one evaluation: every 128^3
    tile product of  A_ij -= U_ki^T U_kj  (Cholesky) and  R_ij -= U_ki^T V_kj  (inverse
    factor); together 2 N^3 / 3 of the N^3 per evaluation."""
    nb = (n + NB - 1) // NB
    tiles = sum((nb - 1 - k) * (nb - k) // 2 + (nb - 1 - k) * (k + 1) for k in range(nb))
    return tiles * 2.0 * NB ** 3


def trsm_flops(n, need_grad=True):
    """Row solves of the sweep: block row k multiplies U_kk^-T into its nb-1-k (value only) or nb-1 (with the
    inverse factor) other 128x128 blocks."""
    nb = (n + NB - 1) // NB
    blocks = sum((nb - 1) if need_grad else (nb - 1 - k) for k in range(nb))
    return blocks * 2.0 * NB ** 3


def _host_threads():
    # the GPU box gives one GPU's share of the host (16 cores); never oversubscribe
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    return max(1, min(avail, 16))


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(n, reps, recipe="cfg2", threads=None, budget_s=None):
    """The oracle (torch-CPU restatement of the reference path, NOT GPyTorch) timed on the
    host cores: value + gradient by autograd through the dense graph, as loss.backward()
    does in the reference.  Median of ``reps`` (>= 5) after one warm-up.
    ``threads``: torch threads (default: one GPU's share of the host, at most 16); ``budget_s``: a time limit for the whole
    measurement -- repetitions stop once it is used up (never before three), and when the warm-up evaluation alone takes more
    than half of it that evaluation is the sample (counts and warm-ups are in ``sample``)."""
    from oracle import sm_mll_oracle as orc
    capped = threads is None
    torch.set_num_threads(_host_threads() if capped else max(1, int(threads)))
    if recipe == "cfg2":
        t, y, e = syn.cfg2(n_obs=n)
        h = syn.cfg_hypers(2, y.double())
        what = f"the same N={n} Q=4 light curve"
    else:
        (t, y, e), per = syn.cfg3_lightcurve(0, n_obs=n)
        h = syn.cfg_hypers(3, y.double(), lead_period=per)
        what = f"light curve 0 of the batch (N={n}, Q=4)"
    x64, y64, nz = t.double(), y.double(), e.double() ** 2
    mu, v = h["mu"].reshape(-1, 1), h["v"].reshape(-1, 1)
    times = []
    val = None
    t0 = time.perf_counter()
    val, _g = orc.mll_value_grad_autograd(x64, y64, h["mean"], nz, h["w"], mu, v)      # warm-up
    first = time.perf_counter() - t0
    if budget_s is not None and first > budget_s / 2:
        times.append(first)                                  # (no room for more within the budget: the warm-up IS the sample)
    else:
        for _ in range(reps):
            t0 = time.perf_counter()
            val, _g = orc.mll_value_grad_autograd(x64, y64, h["mean"], nz, h["w"], mu, v)
            times.append(time.perf_counter() - t0)
            if budget_s is not None and len(times) >= 3 and first + sum(times) > budget_s:
                break
    done = len(times)
    warm = 0 if (budget_s is not None and first > budget_s / 2) else 1
    times.sort()
    med = times[len(times) // 2]
    how = ("torch threads capped at 16 = one GPU's share of the host" if capped else
           f"torch.set_num_threads({torch.get_num_threads()}): BASELINE.md section 2's protocol, every logical CPU this process may be scheduled on"
           f" (CPU quota of the box's cgroup: {_cpu_quota()})")
    return dict(value=1.0 / med, unit="evals/s", cores=torch.get_num_threads(), kind="port", cpu_model=_cpu_model(),
                sample=f"{done} value+grad evaluation(s) ({warm} warm-up) of {what}, median {med * 1e3:.0f} ms; torch-CPU restatement of the "
                       f"reference path (oracle/), not GPyTorch; {how} ({os.cpu_count()} logical CPUs on the box)"), float(val)


def _cpu_quota():
    """CPUs' worth of time the cgroup grants this process ("16.0 CPUs"), or "none"."""
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                return "none" if txt[0] == "max" else f"{int(txt[0]) / int(txt[1]):.1f} CPUs"
            q = int(txt[0])
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            return "none" if q < 0 else f"{q / per:.1f} CPUs"
        except (OSError, ValueError, IndexError, ZeroDivisionError):
            continue
    return "unknown"


def _all_cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def reference_published_workload(dev, iters=1000):
    """The one workload the reference publishes a time for (paper/paper.md:113; comparison notebook "pgmuvi 1D" cell, recorded output
    :496-503): 89 points, Q=2, fp64, FixedNoise likelihood, 1000 AdamW iterations at lr 0.05 -- 11.19 s, 89.56 it/s on Google Colab
    (hardware unstated).  The same fit from the notebook's printed initial values (tests/golden/notebook_pin_1d.npz: the light curve
    and the start values, numbers only) through ``train()`` -- the reference-shaped loop on the Python surface -- and
    ``train_native`` -- the device-resident loop; each timed once after a short untimed run that creates workspace and graphs."""
    import numpy as np
    from pgmuvi_amd import gpytorch as g
    from pgmuvi_amd.trainers import train, train_native
    p = np.load(os.path.join(ROOT, "tests", "golden", "notebook_pin_1d.npz"), allow_pickle=False)
    D = torch.float64
    x, y, noise = (torch.as_tensor(p[k], dtype=D).to(dev) for k in ("x", "y", "noise"))

    def make():
        lik = g.likelihoods.FixedNoiseGaussianLikelihood(noise)

        class Model(g.models.ExactGP):
            def __init__(self):
                super().__init__(x, y, lik)
                self.mean_module = g.means.ConstantMean()
                self.covar_module = g.kernels.SpectralMixtureKernel(num_mixtures=2)

            def forward(self, xx):
                return g.distributions.MultivariateNormal(self.mean_module(xx), self.covar_module(xx))

        m = Model().to(D).to(dev)
        m.initialize(**{"mean_module.constant": torch.tensor(float(p["nb_init_constant"]), dtype=D, device=dev),
                        "covar_module.mixture_weights": torch.as_tensor(p["nb_init_weights"], dtype=D).to(dev),
                        "covar_module.mixture_means": torch.as_tensor(p["nb_init_means"], dtype=D).reshape(2, 1, 1).to(dev),
                        "covar_module.mixture_scales": torch.as_tensor(p["nb_init_scales"], dtype=D).reshape(2, 1, 1).to(dev)})
        return m, lik

    out = {"workload": f"N={int(x.numel())}, Q=2, fp64, FixedNoiseGaussianLikelihood, {iters} AdamW iterations at lr 0.05 (every iteration run: "
                       "no early stop), the comparison notebook's light curve and printed start values",
           "reference_published": {"wall_s": 11.19, "it_per_s": 89.56,
                                   "source": "/root/reference/paper/paper.md:113; docs/source/notebooks/PGMUVI_comparison_with_other_codes.ipynb:496-503",
                                   "hardware": "Google Colab, hardware unstated -- context, not a same-node comparison"},
           "recorded_final_loss": float(p["nb_final_loss"])}
    for name, fn, kw in (("train", train, dict(progress=False)), ("train_native", train_native, dict(check_every=250))):
        m, lik = make()
        fn(model=m, likelihood=lik, train_x=x, train_y=y, maxiter=20, lr=0.05, optim="AdamW", stop=None, **kw)     # untimed: workspace, graphs
        m, lik = make()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = fn(model=m, likelihood=lik, train_x=x, train_y=y, maxiter=iters, lr=0.05, optim="AdamW", stop=None, **kw)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        loss = [float(v) for v in res["loss"]]
        out[name] = {"wall_s": round(dt, 4), "it_per_s": round(len(loss) / dt, 1), "iterations": len(loss),
                     "best_loss_of_last_300": round(min(loss[-300:]), 4), "speedup_vs_published_it_per_s": round(len(loss) / dt / 89.56, 1)}
    return out


def lib_sha16():
    return hashlib.sha256(open(_hip.lib_path(), "rb").read()).hexdigest()[:16]


def _pmc_workload(pmc):
    """(n, light curves per launch) of the ``tools/evalloop n reps need_grad [q [batch]]`` run a traffic file was measured on."""
    a = str(pmc.get("_workload", "")).split()[1:]
    try:
        return int(a[0]), (int(a[4]) if len(a) > 4 else 1)
    except (IndexError, ValueError):
        return None


def measured_traffic(fused, n, nloc):
    """Memory-side bytes per sweep launch from the committed rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE in separate
    runs, FETCH x2 for gfx950; tools/pmc_traffic.py) -- quoted only when that file was measured on the library that is
    running now (the file records the library's sha256); a stale file gives ``None``."""
    sha = lib_sha16()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", PMC_GLOB)), reverse=True):
        pmc = json.load(open(path))
        if pmc.get("_lib_sha16") != sha or _pmc_workload(pmc) != (n, nloc):
            continue
        rows = [v for k, v in pmc.items() if isinstance(v, dict) and
                (k.startswith("k_update") or (fused and (k.startswith("k_diag") or k.startswith("k_trsm"))))]
        tot = sum(r["launches"] for r in rows)
        if tot:
            return (sum(r["hbm_bytes_per_launch_corrected"] * r["launches"] for r in rows) / tot,
                    f"profiles/{os.path.basename(path)} (rocprofv3 --pmc, separate FETCH_SIZE / WRITE_SIZE passes, same workload, "
                    f"same library sha256 {sha})")
    return None, f"no profiles/{PMC_GLOB} measured on this library (sha256 {sha}) for N={n} x {nloc}: see tools/profile.sh"


def sweep_roofline(prof, ws, n, nloc, steps):
    """``roofline`` object of the factorisation sweep's tile GEMM from the per-launch HIP-event times of ``steps`` evaluations
    of ``nloc`` light curves.  Single light curve (fused sweep): the tiles run in all three launch kinds of the chain -- filler
    workgroups of the k_diag launches, the tail of the k_trsm grids, k_update_rows -- so the achieved rate is taken over ALL
    sweep launches (the diagonal-block and row-solve work inside them included in the time); batches (panel sweep): over the
    k_update launches."""
    fused_ms, fused_launches = prof.get("diag_block+trailing_update", (0.0, 0))
    upd_ms, upd_launches = prof["trailing_update"]
    flops = update_flops(n) * nloc * steps
    launch_mix, early_products = None, 0
    if fused_launches:
        trsm_ms, trsm_launches = prof["row_solve"]
        diag_ms, diag_launches = prof["diag_block"]
        flops += trsm_flops(n, need_grad=True) * nloc * steps
        early_products = ws.early_inverse_products()          # (inverse-pass products that ran as fillers inside these launches)
        flops += early_products * 2.0 * NB ** 3 * steps
        launch_mix = {"k_diag": round((fused_ms + diag_ms) / max(fused_launches + diag_launches, 1) * 1e3, 2),
                      "k_trsm": round(trsm_ms / max(trsm_launches, 1) * 1e3, 2),
                      "k_update_rows": round(upd_ms / max(upd_launches, 1) * 1e3, 2)}
        upd_ms += fused_ms + diag_ms + trsm_ms
        upd_launches += fused_launches + diag_launches + trsm_launches
    achieved = flops / (upd_ms * 1e-3) / 1e12 if upd_ms > 0 else 0.0
    traffic, traffic_src = measured_traffic(bool(fused_launches), n, nloc)
    kname = ("factorisation sweep, all launches (k_diag: diagonal block + filler tiles incl. early inverse-pass products; k_trsm / k_trsm16: row solve + update tiles; "
             "k_update_rows): trailing-update + row-solve tile GEMM"
             if fused_launches else "trailing_update (k_update)") + ", v_mfma_f64_16x16x4_f64 TN"
    roofline = dict(bound="mfma", kernel=kname,
                    achieved=round(achieved, 3), peak=FP64_MATRIX_PEAK_TFLOPS, unit="TFLOP/s",
                    frac=round(achieved / FP64_MATRIX_PEAK_TFLOPS, 4), traffic=traffic, traffic_source=traffic_src,
                    avg_launch_us=round(upd_ms / max(upd_launches, 1) * 1e3, 2), launches_per_eval=upd_launches // max(steps, 1),
                    flops_per_eval=flops / max(steps, 1))
    if launch_mix:
        roofline["avg_launch_us_by_kernel"] = launch_mix
        roofline["early_inverse_products_per_eval"] = early_products
    return roofline


BUILD_PAIR_INSTRS = 24    # fp64 VALU instructions per (pair, mixture) of the 1-D build (18 of them the branch-free exp)
VALU_F64_CYCLES = 5.5     # MEASURED issue cost of an fp64 VALU instruction per wavefront and SIMD on this part: tools/lab/ratelab issues
                          # long dependent-free runs of each instruction of the exp (v_fma_f64, v_mul_f64, v_add_f64, v_max_f64, v_rndne_f64,
                          # v_cvt_i32_f64, v_ldexp_f64) at 1, 2 and 4 wavefronts per SIMD and divides clock ticks by instructions: 5.3-5.9
                          # cycles for every one of them (profiles/r04_ratelab.txt) -- not the 4 of "16 lanes per clock"


def build_roofline(prof, n, nloc, q=4, beside=False):
    """The HBM-bound stage of the path: the kernel build streams the upper block triangle of A out
    once -- 8 N^2 / 2 bytes at tile granularity -- from ``prof`` = per-launch HIP-event times of a run in which the WHOLE
    matrix is built by k_build (``whole_build_profile``).  Beside the achieved GB/s stands the floor the Q N^2 / 2 fp64 exp
    set (VALU issue: the build is bound by them, not by HBM)."""
    nbk = (n + NB - 1) // NB
    build_ms, build_launches = prof["sm_build"]
    ntiles = nbk * (nbk + 1) // 2
    build_bytes = 8.0 * NB * NB * ntiles * nloc
    us = build_ms / max(build_launches, 1) * 1e3
    gbs = build_bytes / (us * 1e-6) / 1e9 if us > 0 else 0.0
    # the floor of the build's arithmetic: its fp64 VALU instructions at the measured issue rate, 256 CUs x 4 SIMDs at 2.4 GHz
    valu_floor_us = ntiles * nloc * NB * NB * q * BUILD_PAIR_INSTRS * VALU_F64_CYCLES / (256 * 4 * 64 * 2.4e3)
    kernel = "k_build (spectral-mixture kernel matrix, upper block triangle, one launch)"
    if beside:
        kernel += ("; timed on a side workspace with PGM_BUILD_BESIDE=0 PGM_PREBUILD=0, outside the timed region -- in the production path of one "
                   "light curve only block row 0 is a launch of its own, the other tiles are built by the spare workgroups of "
                   "the first k_diag launch, hidden beside diagonal block 0 (up to 55 tiles: factors and matrix by one launch, k_prebuild)")
    return dict(bound="hbm", kernel=kernel, achieved=round(gbs, 1), peak=8000.0, unit="GB/s", frac=round(gbs / 8000.0, 4),
                avg_launch_us=round(us, 2), bytes_per_launch=build_bytes,
                instructions_per_pair_and_mixture=BUILD_PAIR_INSTRS, valu_cycles_per_instruction_measured=VALU_F64_CYCLES,
                valu_floor_us_at_measured_issue_rate=round(valu_floor_us, 2),
                frac_of_valu_floor_at_measured_issue_rate=round(valu_floor_us / us, 4) if us > 0 else None,
                note="the build is bound by the Q N^2 / 2 fp64 exp it evaluates (VALU issue), not by HBM: the floor beside the GB/s is that arithmetic "
                     "at the measured issue rate of the instructions (tools/lab/ratelab)")


def whole_build_profile(dev, data, n, reps=5):
    """Per-launch times of ``reps`` value-only evaluations of one light curve on a workspace created with
    PGM_BUILD_BESIDE=0 (the switch is read when a workspace is made): there the whole matrix is one k_build launch."""
    prev = {k: os.environ.get(k) for k in ("PGM_BUILD_BESIDE", "PGM_PREBUILD")}
    os.environ["PGM_BUILD_BESIDE"] = "0"
    os.environ["PGM_PREBUILD"] = "0"                        # (short light curves: factors and matrix otherwise come from one launch, k_prebuild)
    try:
        side = _hip.Workspace(dev, (n + NB - 1) // NB * NB, 4, 1, 1)
    finally:
        for k, v in prev.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    one = {k: v[0] for k, v in data.items()}
    f = lambda: _hip.mll_value_grad(one["x"], one["y"], one["mean"], one["noise"], None, one["w"], one["mu"], one["v"], 0, 0.0, False,
                                    workspace=side)
    f(); torch.cuda.synchronize()
    side.profile(True)
    for _ in range(reps):
        f()
    prof = side.profile_read()
    side.profile(False)
    side.close()
    return prof


class Harness:
    """Barrier + synchronize on both sides of exactly K timed steps, MAX over ranks."""

    def __init__(self, dev, world):
        self.dev, self.world = dev, world

    def fence(self):
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
            torch.cuda.synchronize()

    def time(self, step, steps, warmup):
        res = None
        for _ in range(warmup):
            res = step()
        self.fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            res = step()
        self.fence()
        on_host = dist.is_initialized() and dist.get_backend() == "gloo"
        tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cpu" if on_host else self.dev)
        if dist.is_initialized():
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item()), res


def strong_scaling_run(h, total, n, steps, warmup, rank, world, dev, chunk=None, profile=True):
    """One ``total``-light-curve batch (config-3 recipe) evaluated by all ranks together: block partition, chunked batched
    evaluation per rank, one all_gather of the log-likelihoods per step."""
    counts = [shard_bounds(total, r, world)[1] - shard_bounds(total, r, world)[0] for r in range(world)]
    nloc = counts[rank]
    chunk = chunk or default_chunk(n)
    shard = make_shard(total, rank, world, n, "cfg3", dev)
    ws = _hip.get_workspace(dev, n, 4, 1, max(1, min(chunk, nloc)))
    step = lambda: sharded_batch_step(shard, total, chunk)
    if warmup == 0 and nloc:
        # no untimed full pass (it would cost as much as the measurement): the launch sequences of the two chunk shapes are
        # captured on one chunk each instead
        for lo, hi in sorted({(0, min(chunk, nloc)), (nloc - ((nloc % chunk) or min(chunk, nloc)), nloc)}):
            sharded_batch_step({k: v[lo:hi] for k, v in shard.items()}, hi - lo, chunk, group=False)
    elapsed, (out, ll) = h.time(step, steps, warmup)
    assert nloc == 0 or int(out["info"].abs().max()) == 0, "factorisation failed inside the timed region"
    assert ll.numel() == total and bool(torch.isfinite(ll).all())
    tfl = float(n) ** 3 * total * steps / elapsed / world / 1e12
    res = dict(total_batch=total, n=n, steps=steps, warmup=warmup, evals_per_s=round(total * steps / elapsed, 3),
               ms_per_step=round(elapsed / steps * 1e3, 4), light_curves_per_gpu=counts, chunk=chunk,
               algorithmic_tflops_per_gpu=round(tfl, 3), algorithmic_frac_of_fp64_mfma_peak=round(tfl / FP64_MATRIX_PEAK_TFLOPS, 4),
               loglik_checksum=float(ll.sum()))
    res.update(fixture_deviation(ll, total, n))
    prof = None
    if profile and nloc:
        ws.profile(True)
        step()
        prof = ws.profile_read()
        ws.profile(False)
    return res, prof, ws, nloc, (float(ll[0]) if total else None)


def fixture_deviation(ll, total, n):
    """BASELINE configs[2] at its stated size: the gathered log-likelihoods of the 512 x N=2048 batch against the committed oracle
    values (tests/golden/expect_cfg3_b512_n2048.npz, made in the build container by tests/golden/make_golden.py --fullsize from
    inputs generated with the reference's helpers).  Numbers only are read: nothing under oracle/ is imported or run."""
    path = os.path.join(ROOT, "tests", "golden", f"expect_cfg3_b{total}_n{n}.npz")
    if not os.path.exists(path):
        return {}
    import numpy as np
    want = torch.as_tensor(np.load(path, allow_pickle=False)["mll"], dtype=torch.float64)
    dev_ = float((ll.detach().cpu().double() - want).abs().max())
    return {"max_abs_dev_vs_fixture": dev_, "fixture": f"tests/golden/{os.path.basename(path)} (oracle values of all {total} light curves)",
            "fixture_tolerance": 1e-9}


def ragged_run(h, total, n_lo, n_hi, steps, warmup, rank, world, dev):
    """A ``total``-light-curve batch with N ~ U{n_lo..n_hi} (what a real many-light-curve batch is: every pgmuvi ``Lightcurve``
    has its own N): the light curves are dealt to the ranks by their N^3 (``balanced_assignment``), each rank runs its own
    through the ragged entry point (launch sets that share a chain length), one all_gather of the log-likelihoods per step."""
    from pgmuvi_amd.batch import make_ragged_shard, sharded_ragged_step
    shard = make_ragged_shard(total, rank, world, n_lo, n_hi, device=dev)
    step = lambda: sharded_ragged_step(shard, device=dev)
    elapsed, (out, ll) = h.time(step, steps, warmup)
    assert not shard["index"] or int(out["info"].abs().max()) == 0, "factorisation failed inside the timed region"
    assert ll.numel() == total and bool(torch.isfinite(ll).all())
    work = sum(float(n) ** 3 for n in shard["lengths"])
    tfl = work * steps / elapsed / world / 1e12
    sets = list(out.get("launch_sets", [])) if shard["index"] else []     # (the sets the last timed step ran, from the call itself)
    return dict(total_batch=total, n_range=[n_lo, n_hi], steps=steps, warmup=warmup, evals_per_s=round(total * steps / elapsed, 3),
                ms_per_step=round(elapsed / steps * 1e3, 4), light_curves_per_gpu=[shard["owner"].count(r) for r in range(world)],
                launch_sets_rank0=len(sets), block_rows_of_the_sets_rank0=sets,
                algorithmic_tflops_per_gpu=round(tfl, 3), algorithmic_frac_of_fp64_mfma_peak=round(tfl / FP64_MATRIX_PEAK_TFLOPS, 4),
                sum_n3_over_padded=round(work / (total * float(n_hi) ** 3), 4), loglik_checksum=float(ll.sum()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1, help="weak scaling: light curves per GPU per step")
    ap.add_argument("--total-batch", type=int, default=0, help="strong scaling: light curves of the whole batch, sharded over the GPUs")
    ap.add_argument("--chunk", type=int, default=0, help="strong scaling: light curves per launch set (0: by memory, at most 64)")
    ap.add_argument("--n", "--npoints", dest="n", type=int, default=4096,
                    help="points per light curve (behind torch.distributed.run spell it --npoints: '--n' is an ambiguous prefix there)")
    ap.add_argument("--cpu-reps", type=int, default=5)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the batched / strong-scaling side measurements")
    ap.add_argument("--no-big-batch", action="store_true", help="skip the 4096 x N=4096 strong-scaling side measurement (about 16 s on one GPU)")
    ap.add_argument("--spawn", action="store_true", help="go through the self-launcher even with --gpus 1 (a one-rank RCCL job)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="TEST MODE for a one-GPU box (never a default; PGM_BENCH_SHARE_GPU=1 does the same): all --gpus N ranks use device 0 and talk "
                         "over gloo with the log-likelihoods staged through host memory -- the HIP path in N processes of one job; refused when "
                         "two or more devices are visible")
    args = ap.parse_args()
    share = args.share_gpu or os.environ.get("PGM_BENCH_SHARE_GPU") == "1" or os.environ.get(launch.SHARE_VAR) == "1"

    if not launch.under_a_launcher() and (args.gpus > 1 or args.spawn):
        # parent only: nothing here touches a GPU -- the devices are counted from the kernel driver's topology files and the
        # *_VISIBLE_DEVICES variables, not through the HIP runtime (None = unknown: the ranks then report what they find)
        sys.exit(launch.spawn_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus,
                                    visible_devices=launch.visible_gpu_count(), share_gpu=share))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (start it plainly, or with --nproc-per-node {args.gpus})")
    if share:
        if torch.cuda.device_count() != 1:
            sys.exit(f"bench.py: --share-gpu is the one-GPU test mode, {torch.cuda.device_count()} devices visible")
        local = 0                                                # every rank on the one device
    if not torch.cuda.is_available() or local >= torch.cuda.device_count():
        sys.exit(f"bench.py: rank {rank} needs GPU {local}, {torch.cuda.device_count()} visible (there is no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # one rank per GPU over RCCL ("nccl" is RCCL on ROCm) whenever a launcher made this process a rank -- also a job of one
    # rank, which lets a one-GPU box exercise the collective path (--spawn; PGM_BENCH_DIST=1 does the same without a launcher)
    if launch.under_a_launcher() or os.environ.get("PGM_BENCH_DIST") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if share:       # RCCL cannot put two ranks on one device: gloo, values staged through host memory (pgmuvi_amd.batch)
            dist.init_process_group("gloo", init_method=launch.init_method_of(), rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", init_method=launch.init_method_of(), device_id=dev, rank=rank, world_size=world)
    h = Harness(dev, world)
    par_note = (f"TEST MODE --share-gpu: {world} ranks on ONE device over gloo (host-staged gather) -- exercises the multi-process path, "
                f"not a scaling measurement; " if share else "")
    n = args.n
    metric = "marginal-log-likelihood evals/sec, N=4096 Q=4 SM kernel" if n == 4096 else f"marginal-log-likelihood evals/sec, N={n} Q=4 SM kernel"

    if args.total_batch > 0:
        # ---------------- strong scaling: one batch, all ranks
        total = args.total_batch
        res, prof, ws, nloc, first = strong_scaling_run(h, total, n, args.steps, args.warmup, rank, world, dev, args.chunk or None)
        if rank == 0:
            result = {
                "metric": metric, "value": res["evals_per_s"], "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                "dtype": "f64", "data": "synthetic",
                "config": {"workload": f"cfg3-style batch: {total} independent 1-D light curves N={n}, Q=4 spectral-mixture exact GP, fp64, "
                                       f"value + full gradient, the whole batch per step", "total_batch": total, "n": n, "q": 4, "d": 1,
                           "light_curves_per_gpu": res["light_curves_per_gpu"], "chunk": res["chunk"],
                           "parallelism": par_note + f"block partition of the batch over {world} GPU(s), one all_gather of the {total} log-liks per step"},
                "whole_evaluation": {"algorithmic_flops_per_eval": float(n) ** 3, "tflops_per_gpu": res["algorithmic_tflops_per_gpu"],
                                     "frac_of_fp64_mfma_peak": res["algorithmic_frac_of_fp64_mfma_peak"]},
            }
            if prof is not None:                                  # one profiled pass of rank 0's shard
                result["roofline"] = sweep_roofline(prof, ws, n, nloc, 1)
                result["roofline_build"] = build_roofline(prof, n, min(nloc, res["chunk"]))     # (batches build the whole matrix in k_build)
                result["phase_ms_per_step"] = {k: round(v[0], 4) for k, v in prof.items()}
            if world == 1 and not args.no_cpu:
                cb, cpu_val = cpu_baseline(n, args.cpu_reps, "cfg3")
                result["cpu_baseline"] = cb
                result["parity"] = {"abs_dmll_vs_cpu_oracle": abs(first - cpu_val), "tolerance": 1e-4, "mll": first}
                result["speedup_vs_cpu_baseline"] = round(res["evals_per_s"] / cb["value"], 1)
            print(json.dumps(result))
        if dist.is_initialized():
            dist.destroy_process_group()
        return

    # ---------------- headline: weak scaling, B light curves per GPU per step
    B = args.batch
    data = make_batch(n, B, rank, dev)
    ws = _hip.get_workspace(dev, n, 4, 1, B)

    def step():
        out = _hip.mll_value_grad(data["x"], data["y"], data["mean"], data["noise"], None, data["w"], data["mu"], data["v"],
                                  0, 0.0, True, workspace=ws)
        ll = gather_logliks(out["mll"], B * world)
        return out, ll

    elapsed, (out, ll) = h.time(step, args.steps, args.warmup)
    assert int(out["info"].abs().max()) == 0, "factorisation failed inside the timed region"
    assert ll.numel() == B * world and bool(torch.isfinite(ll).all())
    ms_per_step = elapsed / args.steps * 1e3
    value = B * world * args.steps / elapsed
    gpu_val = float(out["mll"].reshape(-1)[0])

    # ---- per-kernel device time (HIP events on the launch stream), same K steps again
    ws.profile(True)
    for _ in range(args.steps):
        step()
    prof = ws.profile_read()
    ws.profile(False)
    roofline = sweep_roofline(prof, ws, n, B, args.steps)
    nbk = (n + NB - 1) // NB
    # (one light curve never builds its matrix in one k_build launch in production: up to 55 tiles k_prebuild makes factors and
    #  matrix together, beyond 32 tiles the spare workgroups of the first k_diag launch build all but block row 0)
    beside = B == 1
    # the HBM-bound stage, whole: where the production path hides most of the build inside the first k_diag launch it is
    # timed once on a side workspace that builds the matrix in one launch (outside the timed region)
    roofline_build = build_roofline(whole_build_profile(dev, data, n) if beside else prof, n, 1 if beside else B, beside=beside)
    phases = {k: round(v[0] / args.steps, 4) for k, v in prof.items()}

    result = None
    if rank == 0:
        result = {
            "metric": metric,
            "value": round(value, 3), "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"cfg2: 1-D light curve N={n}, Q=4 spectral-mixture exact GP, fp64, value + full gradient, "
                                   f"{B} light curve(s) per GPU per step", "batch_per_gpu": B, "n": n, "q": 4, "d": 1,
                       "parallelism": par_note + (f"independent light curves per GPU x{world}, all_gather of log-liks per step" if world == 1 else
                                       f"ONE light curve per GPU x{world}: a ~2 ms evaluation then a blocking all_gather with nothing to overlap it -- at N>1 "
                                       f"this value measures collective latency per step; the multi-GPU figure is `strong_scaling` (top level of "
                                       f"this line): one 512 x N=2048 and one 4096 x N=4096 batch sharded over the {world} GPUs")},
            "roofline": roofline,
            "roofline_build": roofline_build,
            "phase_ms_per_step": phases,
        }
        if world > 1:
            result["note"] = ("weak scaling of ONE light curve per GPU: a step is a ~2 ms evaluation followed by a blocking all_gather with "
                              "nothing to overlap it, so the N>1 values of this line measure collective latency per step; the batch "
                              "measurement in `strong_scaling` is the meaningful multi-GPU figure (north_star: 4096-light-curve batch)")
            result["cpu_baseline"] = None
            result["cpu_baseline_note"] = "timed on rank 0 of the N=1 run only (the contract's rule); see that run's line"
    extra = {}
    if world == 1 and not args.no_extra:
        # measured fp64 MFMA issue rate on this device (context for the datasheet peak)
        extra["mfma_f64_probe_tflops"] = round(_hip.probe_mfma_f64(local), 2)
        # throughput mode: 8 light curves share every launch (batch on gridDim.z)
        B2 = 8
        d2 = make_batch(n, B2, rank, dev)
        ws2 = _hip.get_workspace(dev, n, 4, 1, B2)
        f2 = lambda: _hip.mll_value_grad(d2["x"], d2["y"], d2["mean"], d2["noise"], None, d2["w"], d2["mu"], d2["v"], 0, 0.0, True, workspace=ws2)
        f2(); torch.cuda.synchronize()
        k2 = max(2, args.steps // 4)
        t0 = time.perf_counter()
        for _ in range(k2):
            f2()
        torch.cuda.synchronize()
        bt = time.perf_counter() - t0
        ws2.profile(True)
        for _ in range(2):
            f2()
        p2 = ws2.profile_read()
        ws2.profile(False)
        u2 = p2["trailing_update"][0] + p2.get("diag_block+trailing_update", (0.0, 0))[0]
        extra["batched"] = {"batch_per_gpu": B2, "evals_per_s": round(B2 * k2 / bt, 3),
                            "trailing_update_tflops": round(update_flops(n) * B2 * 2 / (u2 * 1e-3) / 1e12, 2) if u2 > 0 else None,
                            "trailing_update_frac": round(update_flops(n) * B2 * 2 / (u2 * 1e-3) / 1e12 / FP64_MATRIX_PEAK_TFLOPS, 4) if u2 > 0 else None}
        del d2, ws2
    if not args.no_extra:
        # strong scaling beside the headline (at every world size, so that the driver's N=1,2,4,8 runs record it): BASELINE
        # configs[2] (512 x N=2048) and the north_star's 4096-light-curve batch at the headline size; the whole batch per step.
        # The headline's workspaces go first (the batches bring their own, 8.6 GB for 64 x N=4096).
        del ws, data, out, ll, step
        _hip.release_workspaces()
        torch.cuda.empty_cache()
        ss = {}
        cases = [("cfg3_512_x_n2048", 512, 2048, 3, 1)]
        if not args.no_big_batch:
            cases.append(("batch4096_x_n4096", 4096, 4096, 2, 0))     # (no untimed full pass: the two chunk shapes are captured on one chunk each)
        for tag, total, nn, k, w in cases:
            ss[tag] = strong_scaling_run(h, total, nn, k, w, rank, world, dev, None, profile=False)[0]
            _hip.release_workspaces()
        # the same 512 light curves as they come in practice: unequal lengths (ragged N)
        ss["ragged_512_x_n1024_2048"] = ragged_run(h, 512, 1024, 2048, 3, 1, rank, world, dev)
        _hip.release_workspaces()
        extra["strong_scaling"] = ss                             # (top level of the line: `result.update(extra)` below)
        if world == 1:
            extra["reference_published_workload"] = reference_published_workload(dev)
            _hip.release_workspaces()
    if rank == 0 and world == 1 and not args.no_cpu:
        cb, cpu_val = cpu_baseline(n, args.cpu_reps)
        result["cpu_baseline"] = cb
        result["parity"] = {"abs_dmll_vs_cpu_oracle": abs(gpu_val - cpu_val), "tolerance": 1e-4, "mll": gpu_val}
        result["speedup_vs_cpu_baseline"] = round(value / cb["value"], 1)
        # BASELINE.md section 2 / SURVEY.md section 8d word the protocol with torch.set_num_threads(os.cpu_count()): that figure
        # beside the 16-thread one (`cpu_baseline` stays the one-GPU share of the host, as in every earlier round's record)
        if _all_cores() != cb["cores"]:
            cb_all, val_all = cpu_baseline(n, args.cpu_reps, threads=_all_cores(), budget_s=14.0)
            cb_all["abs_dmll_vs_16_thread_run"] = abs(val_all - cpu_val)
            result["cpu_baseline_all_cores"] = cb_all
            result["speedup_vs_cpu_baseline_all_cores"] = round(value / cb_all["value"], 1)
        else:
            result["cpu_baseline_all_cores"] = dict(cb, note="this process may use no more CPUs than the 16-thread figure already does")
    if rank == 0:
        result.update(extra)
        if "strong_scaling" in extra:
            # the driver's parser keeps `config`: the batch figures (whole job, all ranks) stand there too, beside the caveat
            result["config"]["strong_scaling_evals_per_s"] = {k: v["evals_per_s"] for k, v in extra["strong_scaling"].items()}
            dv = extra["strong_scaling"].get("cfg3_512_x_n2048", {}).get("max_abs_dev_vs_fixture")
            if dv is not None:
                result["config"]["cfg3_512_x_n2048_max_abs_dev_vs_fixture"] = dv
        print(json.dumps(result))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
