"""SURVEY.md section 8e on hardware a one-GPU box has: the HIP path in TWO processes of one job.

RCCL cannot put two ranks on one device, so the ranks of these jobs all use device 0 and talk over gloo with the
log-likelihoods staged through host memory (``bench.py --share-gpu`` / ``launch.spawn_ranks(share_gpu=True)``: a test mode,
refused when two or more devices are visible).  Everything else is the multi-GPU path as the driver's 8-GPU run will take it:
the parent that never touches a GPU, one fresh child per rank, ``make_shard`` / ``make_ragged_shard`` per rank, the batched
and ragged HIP entry points, ONE gather per step, MAX-over-ranks timing, rank 0's JSON line relayed, a dying rank taking the
job down.  The expected numbers are this process's own one-rank HIP results: bit for bit.
"""
import io
import json
import os
import subprocess
import sys
import time

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "_gpu_rank_worker.py")

from pgmuvi_amd import launch  # noqa: E402
from pgmuvi_amd.batch import make_ragged_shard, make_shard, sharded_batch_step, sharded_ragged_step  # noqa: E402


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("-m gpu tests need the MI355X")
    if torch.cuda.device_count() != 1:
        pytest.skip("--share-gpu is the one-GPU test mode")
    return torch.device("cuda:0")


def _kfd_holders():
    """Processes (other than this one) that have the GPU's compute device open."""
    me, held = str(os.getpid()), []
    for pid in os.listdir("/proc"):
        if not pid.isdigit() or pid == me:
            continue
        try:
            for fd in os.listdir(f"/proc/{pid}/fd"):
                if os.readlink(f"/proc/{pid}/fd/{fd}") == "/dev/kfd":
                    held.append(int(pid))
                    break
        except OSError:
            continue
    return held


def test_two_ranks_on_one_gpu_equal_one_process_bit_for_bit(dev, tmp_path):
    total, n, chunk = 16, 512, 8
    rag = (14, 100, 700)
    before = _kfd_holders()
    buf, err = io.StringIO(), open(tmp_path / "stderr.txt", "w+")
    rc = launch.spawn_ranks([sys.executable, WORKER, "--gpus", "2", "--total-batch", str(total), "--npoints", str(n), "--chunk", str(chunk),
                             "--ragged", *map(str, rag), "--outdir", str(tmp_path)], 2, visible_devices=launch.visible_gpu_count(),
                            share_gpu=True, out=buf, err=err)
    err.seek(0)
    assert rc == 0, err.read()[-4000:]
    line = json.loads([ln for ln in buf.getvalue().splitlines() if ln.startswith("{")][-1])
    assert line["world"] == 2 and line["backend"] == "gloo" and line["lib"] == "libpgmuvi_hip.so"
    assert [r[0] for r in line["ranks"]] == [0, 1] and [r[2] for r in line["ranks"]] == [8, 8] and [r[3] for r in line["ranks"]] == [0, 0]
    assert len({r[1] for r in line["ranks"]} | {os.getpid()}) == 3            # two processes of their own, neither is this one
    assert _kfd_holders() == before                                             # nobody of the job still holds the device
    # ---- the one-process results, here
    whole = make_shard(total, 0, 1, n, "cfg3", dev)
    ref, ref_ll = sharded_batch_step(whole, total, chunk)         # the same launch sets of 8 the two ranks ran
    ref1, ref1_ll = sharded_batch_step(whole, total, None)        # ... and the whole batch as ONE launch set
    rwhole = make_ragged_shard(rag[0], 0, 1, rag[1], rag[2], device=dev)
    rref, rref_ll = sharded_ragged_step(rwhole, device=dev)
    torch.cuda.synchronize()
    assert torch.equal(ref_ll, ref1_ll)                           # (values do not depend on the launch set)
    owners = set()
    for r in range(2):
        got = torch.load(tmp_path / f"rank{r}.pt")
        assert int(got["info"].abs().max()) == 0 and got["nloc"] == 8
        assert torch.equal(got["ll"], ref_ll.cpu())               # every rank holds the WHOLE vector, bit for bit
        for k in ("g_w", "g_mu", "g_v", "g_noise", "g_mean"):
            assert torch.equal(got[k], ref[k][8 * r: 8 * r + 8].cpu()), k
        # ragged: dealt by N^3, gathered back into the batch's order; each member's value is its own single evaluation's
        assert torch.equal(got["ragged_ll"], rref_ll.cpu())
        assert got["ragged_index"] == make_ragged_shard(rag[0], r, 2, rag[1], rag[2])["index"]
        owners |= set(got["ragged_index"])
        assert len(got["ragged_index"]) >= 1
    assert owners == set(range(rag[0]))


def test_four_ranks_with_unequal_shards_on_one_gpu(dev, tmp_path):
    """Four ranks of one job on the one device (five processes on the card with this one: the box allows six): ten light curves
    dealt 3 + 3 + 2 + 2, a ragged batch of nine dealt by N^3 -- every rank ends with the whole vector, bit for bit the
    one-process result, and the gradients of its own block."""
    total, n, chunk = 10, 256, 4
    rag = (9, 50, 400)
    before = _kfd_holders()
    buf, err = io.StringIO(), open(tmp_path / "stderr.txt", "w+")
    rc = launch.spawn_ranks([sys.executable, WORKER, "--gpus", "4", "--total-batch", str(total), "--npoints", str(n), "--chunk", str(chunk),
                             "--ragged", *map(str, rag), "--outdir", str(tmp_path)], 4, visible_devices=launch.visible_gpu_count(),
                            share_gpu=True, out=buf, err=err)
    err.seek(0)
    assert rc == 0, err.read()[-4000:]
    line = json.loads([ln for ln in buf.getvalue().splitlines() if ln.startswith("{")][-1])
    assert line["world"] == 4 and [r[2] for r in line["ranks"]] == [3, 3, 2, 2]
    assert len({r[1] for r in line["ranks"]}) == 4 and _kfd_holders() == before
    whole = make_shard(total, 0, 1, n, "cfg3", dev)
    ref, ref_ll = sharded_batch_step(whole, total, None)
    rwhole = make_ragged_shard(rag[0], 0, 1, rag[1], rag[2], device=dev)
    rref, rref_ll = sharded_ragged_step(rwhole, device=dev)
    torch.cuda.synchronize()
    lo, owners = 0, []
    for r in range(4):
        got = torch.load(tmp_path / f"rank{r}.pt")
        k = got["nloc"]
        assert int(got["info"].abs().max()) == 0
        assert torch.equal(got["ll"], ref_ll.cpu())                           # values: the same bits whatever the launch set
        for key in ("g_w", "g_mu", "g_v"):                                    # gradient sums: split by the call's work items -> rounding
            assert torch.allclose(got[key], ref[key][lo: lo + k].cpu(), rtol=1e-10, atol=1e-12), key
        assert torch.equal(got["ragged_ll"], rref_ll.cpu())
        owners += got["ragged_index"]
        lo += k
    assert lo == total and sorted(owners) == list(range(rag[0]))


def test_chains_over_two_ranks_on_one_gpu_equal_one_process(dev, tmp_path):
    """Config 5's layout on the hardware a one-GPU box has: four chains, each on its own light curve, dealt two and two over the
    ranks of one job; each rank drives its chains through the native potential (``pgm_pot_*``) and ONE gather of the draws
    closes the run.  Both ranks end up with all four chains, and they are the chains this process samples alone: the same
    trees (leapfrog counts), the same draws (a chain's numbers depend on its id, not on where it ran or with whom)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _gpu_rank_worker as wk
    from pgmuvi_amd import mcmc
    C, n = 4, 300
    buf, err = io.StringIO(), open(tmp_path / "stderr.txt", "w+")
    rc = launch.spawn_ranks([sys.executable, WORKER, "--gpus", "2", "--chains", str(C), str(n), "--outdir", str(tmp_path)], 2,
                            visible_devices=launch.visible_gpu_count(), share_gpu=True, out=buf, err=err)
    err.seek(0)
    assert rc == 0, err.read()[-4000:]
    line = json.loads([ln for ln in buf.getvalue().splitlines() if ln.startswith("{")][-1])
    assert line["world"] == 2 and line["backend"] == "gloo" and line["chains"] == C
    assert len({r[1] for r in line["ranks"]} | {os.getpid()}) == 3
    x, y, nz, kw = wk.chain_problem(C, n)
    one = mcmc.run_mcmc(x.to(dev), y.to(dev), nz.to(dev), **kw)
    for r in range(2):
        got = torch.load(tmp_path / f"chains_rank{r}.pt", weights_only=False)
        assert np.array_equal(got["n_leapfrog"], one["_diagnostics"]["n_leapfrog"])
        assert got["covar_module.mixture_means_prior"].shape == (C, 5, 2, 1, 1)
        for k in ("covar_module.mixture_means_prior", "covar_module.mixture_weights_prior", "covar_module.mixture_scales_prior",
                  "mean_module.mean_prior"):
            assert np.allclose(got[k], one[k], rtol=1e-9, atol=0), k
        assert np.allclose(got["potential_energy"], one["_diagnostics"]["potential_energy"], rtol=1e-10, atol=1e-9)
        assert np.array_equal(got["step_size"], one["_diagnostics"]["step_size"]) or \
            np.allclose(got["step_size"], one["_diagnostics"]["step_size"], rtol=1e-9)


def test_bench_two_ranks_share_the_gpu(dev):
    """``python bench.py --gpus 2 --share-gpu`` as typed: strong scaling (16 x N=512 over two ranks) and the weak headline form;
    rank 0's line comes back through the parent, timing is the MAX over both ranks, the batch is split 8 + 8."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", launch.SHARE_VAR, launch.INIT_VAR):
        env.pop(k, None)
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--steps", "2", "--warmup", "1", "--no-cpu"]
    r = subprocess.run(base + ["--npoints", "512", "--total-batch", "16"], capture_output=True, text=True, timeout=280, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-2500:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["config"]["light_curves_per_gpu"] == [8, 8]
    assert "TEST MODE --share-gpu" in out["config"]["parallelism"] and out["value"] > 0
    r = subprocess.run(base + ["--npoints", "1024", "--no-extra"], capture_output=True, text=True, timeout=280, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-2500:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0 and out["roofline"]["frac"] > 0
    assert "collective latency" in out["config"]["parallelism"]
    # not without asking, and not where one rank per GPU is possible
    plain = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--no-cpu"], capture_output=True,
                           text=True, timeout=120, env=env, cwd=ROOT)
    assert plain.returncode == 2 and "needs 2 visible GPUs" in plain.stderr


def test_a_rank_dying_after_its_first_hip_call_takes_the_job_down(dev, tmp_path):
    """Rank 1 evaluates its shard on the GPU and then leaves with status 3; rank 0 is left waiting in the gather.  The launcher
    reports 3, stops rank 0 within the grace period and nothing of the job holds the device afterwards."""
    before = _kfd_holders()
    buf, err = io.StringIO(), open(tmp_path / "stderr.txt", "w+")
    t0 = time.time()
    rc = launch.spawn_ranks([sys.executable, WORKER, "--gpus", "2", "--total-batch", "8", "--npoints", "256", "--outdir", str(tmp_path),
                             "--fail-rank", "1"], 2, visible_devices=launch.visible_gpu_count(), share_gpu=True, grace_s=5.0,
                            out=buf, err=err)
    took = time.time() - t0
    err.seek(0)
    text = err.read()
    assert rc == 3, text[-3000:]
    assert "leaving with status 3 after 4 HIP evaluations" in text and "launch: rank 1 exited with status 3" in text
    assert "{" not in buf.getvalue()                              # rank 0 never got to its line
    assert not os.path.exists(tmp_path / "rank0.pt")
    assert took < 120
    time.sleep(0.5)
    assert _kfd_holders() == before, "gpu_holders_left"
    # the device is usable afterwards
    whole = make_shard(2, 0, 1, 256, "cfg3", dev)
    out, ll = sharded_batch_step(whole, 2, None)
    torch.cuda.synchronize()
    assert int(out["info"].abs().max()) == 0 and bool(torch.isfinite(ll).all())
