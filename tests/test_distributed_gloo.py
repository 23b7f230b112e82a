"""The N>1 path on CPU: world_size-2 ``gloo`` processes shard a batch of light curves,
evaluate their shards (oracle stand-in for the HIP call: no GPU here) and all_gather the
log-likelihoods; the result must equal the single-process evaluation of the whole batch."""
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from pgmuvi_amd import synthetic as syn  # noqa: E402
from pgmuvi_amd.batch import (balanced_assignment, default_chunk, evaluate_batch, evaluate_ragged, gather_logliks,  # noqa: E402
                              make_ragged_shard, make_shard, pad_curves, ragged_lengths, shard_bounds, sharded_batch_step,
                              sharded_ragged_step)

B, N = 5, 40


def _rank_main(target, rank, world, init, outdir, args):
    """Body of one gloo rank: stderr into a file of its own (shown by the parent when the rank fails), rendezvous through a
    file only this job knows (no port to lose to another process between finding it free and binding it), the result as a
    file -- nothing of the rank has to outlive it in a queue."""
    err = os.open(os.path.join(outdir, f"rank{rank}.err"), os.O_WRONLY | os.O_CREAT | os.O_TRUNC)
    os.dup2(err, 2)
    dist.init_process_group("gloo", init_method=init, rank=rank, world_size=world)
    res = target(rank, world, *args)
    torch.save(res, os.path.join(outdir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _run_ranks(target, world, *args, tmp_path):
    """``world`` fresh gloo processes running ``target(rank, world, *args)``; returns their results in rank order."""
    init = f"file://{tmp_path}/rendezvous"
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_rank_main, args=(target, r, world, init, str(tmp_path), args)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
    bad = [(r, p.exitcode) for r, p in enumerate(procs) if p.exitcode != 0]
    for p in procs:
        if p.is_alive():
            p.kill()
    if bad:
        logs = "\n".join(f"--- rank {r} (exit {rc}) stderr:\n" + open(os.path.join(str(tmp_path), f"rank{r}.err")).read()[-3000:] for r, rc in bad)
        raise AssertionError(f"ranks failed: {bad}\n{logs}")
    return [torch.load(os.path.join(str(tmp_path), f"rank{r}.pt")) for r in range(world)]


def _batch():
    xs, ys, ns, ws, mus, vs, ms = [], [], [], [], [], [], []
    for i in range(B):
        (t, y, e), per = syn.cfg3_lightcurve(i, n_obs=N)
        h = syn.cfg_hypers(3, y.double(), lead_period=per)
        xs.append(t.double().reshape(N, 1)); ys.append(y.double()); ns.append(e.double() ** 2)
        ws.append(h["w"]); mus.append(h["mu"].reshape(4, 1)); vs.append(h["v"].reshape(4, 1)); ms.append(h["mean"].expand(N))
    return tuple(torch.stack(L) for L in (xs, ys, ms, ns, ws, mus, vs))


def _worker(rank, world):
    import _oracle_backend as ob
    x, y, m, nz, w, mu, v = _batch()
    lo, hi = shard_bounds(B, rank, world)
    out = evaluate_batch(x[lo:hi], y[lo:hi], m[lo:hi], nz[lo:hi], w[lo:hi], mu[lo:hi], v[lo:hi], _compute=ob.mll_value_grad)
    ll = gather_logliks(out["mll"], B)
    gw = gather_logliks(out["g_w"], B)
    eq = gather_logliks(torch.full((2, 3), float(rank), dtype=torch.float64), 2 * world)   # equal shards: single-collective path
    assert eq.shape == (2 * world, 3) and eq[:, 0].tolist() == [float(r) for r in range(world) for _ in range(2)]
    return rank, ll, gw


def test_world_size_2_shard_and_gather_equals_single_process(tmp_path):
    import _oracle_backend as ob
    got = _run_ranks(_worker, 2, tmp_path=tmp_path)
    x, y, m, nz, w, mu, v = _batch()
    full = evaluate_batch(x, y, m, nz, w, mu, v, _compute=ob.mll_value_grad)
    for rank, ll, gw in got:
        assert ll.shape == (B,) and torch.equal(ll, full["mll"])       # every rank holds the whole vector
        assert torch.equal(gw, full["g_w"])


def _strong_worker(rank, world, total, n, chunk):
    """The code path of ``bench.py --total-batch`` (``make_shard`` + ``sharded_batch_step``) with the oracle stand-in."""
    import _oracle_backend as ob
    shard = make_shard(total, rank, world, n, "cfg3")
    out, ll = sharded_batch_step(shard, total, chunk, _compute=ob.mll_value_grad)
    return rank, shard["y"].shape[0], ll, out.get("g_mu")


@pytest.mark.parametrize("total,world", [(7, 2), (2, 3)])
def test_strong_scaling_step_is_independent_of_the_partition(total, world, tmp_path):
    """A ``total``-light-curve batch evaluated by ``world`` ranks (block partition, chunks of 2 with a ragged tail, one
    all_gather) gives every rank the same vector as one process evaluating the whole batch -- also when shards are unequal
    (7 over 2) or empty (2 over 3)."""
    import _oracle_backend as ob
    n = 48
    got = _run_ranks(_strong_worker, world, total, n, 2, tmp_path=tmp_path)
    whole = make_shard(total, 0, 1, n, "cfg3")
    ref, ref_ll = sharded_batch_step(whole, total, None, _compute=ob.mll_value_grad)       # single process: no collective
    assert torch.equal(ref_ll, ref["mll"]) and ref_ll.shape == (total,)
    assert [g[1] for g in got] == [shard_bounds(total, r, world)[1] - shard_bounds(total, r, world)[0] for r in range(world)]
    for rank, nloc, ll, gmu in got:
        assert torch.equal(ll, ref_ll)
        lo, hi = shard_bounds(total, rank, world)
        if nloc:
            assert torch.equal(gmu, ref["g_mu"][lo:hi])
    assert default_chunk(2048) == 512 and default_chunk(4096) == 256 and 1 <= default_chunk(16384) <= 16


def _ragged_worker(rank, world, total, n_lo, n_hi):
    """Ragged batch: light curves dealt to the ranks by N^3 (``make_ragged_shard``), each rank evaluates its own through the
    ragged entry point (oracle stand-in), one all_gather puts the values back into the batch's order."""
    import _oracle_backend as ob
    shard = make_ragged_shard(total, rank, world, n_lo, n_hi)
    out, ll = sharded_ragged_step(shard, chunk=3, _compute=ob.mll_value_grad_ragged)
    return rank, shard["index"], ll, out.get("g_w"), [g.clone() for g in out.get("g_noise", [])]


@pytest.mark.parametrize("total,world", [(7, 2), (2, 3)])
def test_ragged_batch_over_ranks_is_the_single_process_result(total, world, tmp_path):
    """Unequal N across ranks: ``balanced_assignment`` on N^3 decides who evaluates what, every rank
    ends with the same vector of log-likelihoods in the batch's order as one process evaluating everything -- also with a
    rank that owns nothing (2 light curves over 3 ranks)."""
    import _oracle_backend as ob
    n_lo, n_hi = 24, 70
    got = _run_ranks(_ragged_worker, world, total, n_lo, n_hi, tmp_path=tmp_path)
    whole = make_ragged_shard(total, 0, 1, n_lo, n_hi)
    assert whole["index"] == list(range(total)) and whole["lengths"] == ragged_lengths(total, n_lo, n_hi)
    assert len(set(whole["lengths"])) > 1                             # (ragged indeed)
    ref, ref_ll = sharded_ragged_step(whole, _compute=ob.mll_value_grad_ragged)
    assert ref_ll.shape == (total,)
    # the deal: every light curve exactly once, heaviest-first onto the lightest rank
    owner = balanced_assignment([float(n) ** 3 for n in whole["lengths"]], world)
    assert sorted(i for g in got for i in g[1]) == list(range(total))
    for rank, index, ll, gw, gnoise in got:
        assert index == [i for i in range(total) if owner[i] == rank]
        assert torch.equal(ll, ref_ll)
        for k, i in enumerate(index):
            assert torch.equal(gw[k], ref["g_w"][i])
            assert gnoise[k].shape == (whole["lengths"][i],) and torch.equal(gnoise[k], ref["g_noise"][i])
    # each value is the light curve's own (the equal-length evaluation of that light curve alone)
    c = whole["curves"][total - 1]
    n = whole["lengths"][total - 1]
    alone = ob.mll_value_grad(c["x"].reshape(n, 1), c["y"], c["mean"].expand(n), c["noise"], None, c["w"], c["mu"].reshape(4, 1),
                              c["v"].reshape(4, 1))
    assert torch.equal(alone["mll"], ref_ll[total - 1])


def test_ragged_launch_sets_follow_the_cost_model(monkeypatch):
    """``pgm_ragged_plan`` (host only): light curves are taken by block-row count, longest first.  Twelve and more of them
    (160 block rows in all) form ONE trimmed set -- every member stops at its own last block row -- except stragglers at the
    long end; fewer, or ``PGM_RAGGED_TRIM=0``: padded sets -- a short group joins the set above it only where padding is
    cheaper than a launch set of its own.  Sets never exceed the workspace's batch."""
    from pgmuvi_amd import _hip
    lengths = [1025 + 128 * k for k in range(8) for _ in range(57)] + [900] * 57          # 9 block-row counts, 57 light curves each
    set_of, nbs = _hip.ragged_plan(lengths, 1024)
    assert nbs == [16] and set(set_of) == {0}
    set_of, nbs = _hip.ragged_plan(lengths, 512)                 # 513 light curves: the last one does not fit the first set
    assert nbs == [16, 8] and set_of.count(0) == 512 and set_of.count(1) == 1
    set_of, nbs = _hip.ragged_plan([2048] * 10 + [1900] * 40, 512)
    assert nbs == [16]
    # stragglers: three light curves of 32 block rows among a hundred of 4 keep to themselves (a padded set, the fused sweep)
    set_of, nbs = _hip.ragged_plan([512] * 50 + [4096] * 3 + [500] * 50, 512)
    assert nbs == [32, 4] and [set_of.count(k) for k in range(2)] == [3, 100] and set_of[50:53] == [0, 0, 0]
    # ... but eight that share their upper half of block rows are company enough
    set_of, nbs = _hip.ragged_plan([512] * 50 + [4096] * 3 + [2500] * 5, 512)
    assert nbs == [32] and set(set_of) == {0}
    # fewer than 12 light curves: padded sets
    set_of, nbs = _hip.ragged_plan([2048] * 10 + [1900], 512)     # a lone light curve of 15 block rows rides with the 16-block-row set
    assert nbs == [16] and set(set_of) == {0}
    set_of, nbs = _hip.ragged_plan([200, 2300, 640, 2300, 130], 512)   # the caller's order is kept inside a set, the answer does not depend on it
    assert nbs[0] == 18 and set_of[1] == set_of[3] == 0
    set_of, nbs = _hip.ragged_plan([512] * 20, 8)                 # more members than the workspace holds: split
    assert nbs == [4, 4, 4] and [set_of.count(k) for k in range(3)] == [8, 8, 4]
    with pytest.raises(RuntimeError):
        _hip.ragged_plan([0, 5], 4)
    # the padded sets of many light curves (round 4's first form)
    monkeypatch.setenv("PGM_RAGGED_TRIM", "0")
    set_of, nbs = _hip.ragged_plan(lengths, 512)                 # every count its own set
    assert nbs == [16, 15, 14, 13, 12, 11, 10, 9, 8] and len(set(set_of)) == 9
    assert all(nbs[s2] == (n + 127) // 128 for s2, n in zip(set_of, lengths))
    set_of, nbs = _hip.ragged_plan([2048] * 10 + [1900] * 40, 512)     # 40 * (16^3 - 15^3) * 0.044 > 15 * 60
    assert nbs == [16, 15]


def test_partitioning_rules():
    assert [shard_bounds(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert [shard_bounds(3, r, 8) for r in range(8)][:4] == [(0, 1), (1, 2), (2, 3), (3, 3)]
    cover = sorted(i for r in range(8) for i in range(*shard_bounds(4096, r, 8)))
    assert cover == list(range(4096))
    with pytest.raises(ValueError):
        shard_bounds(4, 4, 4)
    owner = balanced_assignment([n ** 3 for n in (4096, 512, 512, 2048, 2048, 256)], 2)
    loads = [sum(c for c, o in zip([n ** 3 for n in (4096, 512, 512, 2048, 2048, 256)], owner) if o == r) for r in range(2)]
    assert max(loads) == 4096 ** 3            # the big one alone, the rest together
    assert gather_logliks(torch.arange(3.0), 3).tolist() == [0.0, 1.0, 2.0]     # single process: identity


def test_self_launcher_starts_one_rank_per_device_and_relays_rank_0():
    """``bench.py --gpus N`` without torch.distributed.run: ``pgmuvi_amd.launch.spawn_ranks`` starts N fresh children of the
    command line with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, relays rank 0's JSON line and returns the children's
    worst status.  Here: two gloo ranks on the CPU with the oracle stand-in (``tests/_launch_worker.py``), against one process
    evaluating the whole batch; a rank that dies before the rendezvous takes the job down with its status; too few devices
    are refused before anything is started."""
    import io
    import json
    import _oracle_backend as ob
    from pgmuvi_amd import launch
    worker = os.path.join(ROOT, "tests", "_launch_worker.py")
    total, n = 5, 40
    buf = io.StringIO()
    rc = launch.spawn_ranks([sys.executable, worker, "--gpus", "2", "--total-batch", str(total), "--npoints", str(n)], 2, out=buf)
    assert rc == 0
    got = json.loads([ln for ln in buf.getvalue().splitlines() if ln.startswith("{")][-1])
    assert got["world"] == 2 and got["gpus_arg"] == 2 and got["child_mark"] == "1" and got["master"][0] == "127.0.0.1"
    assert got["init"].startswith("file://") and not os.path.exists(os.path.dirname(got["init"][7:]))   # met through a file of the job's own, gone now
    assert got["ranks"] == [[0, 0, 3], [1, 1, 2]]                 # rank == local rank, block partition 3 + 2
    whole = make_shard(total, 0, 1, n, "cfg3")
    _, ref = sharded_batch_step(whole, total, None, _compute=ob.mll_value_grad)
    assert got["loglik"] == ref.tolist()
    # a rank that fails: its status comes back, the rank left waiting in the rendezvous is stopped
    buf = io.StringIO()
    rc = launch.spawn_ranks([sys.executable, worker, "--gpus", "2", "--fail-rank", "1"], 2, grace_s=2.0, out=buf)
    assert rc == 3 and "{" not in buf.getvalue()
    # fewer visible devices than ranks: refused, nothing started
    err = io.StringIO()
    assert launch.spawn_ranks([sys.executable, "-c", "raise SystemExit(0)"], 2, visible_devices=1, err=err) == 2
    assert "needs 2 visible GPUs" in err.getvalue()
    # the one-GPU test mode is refused where one rank per GPU is possible (and where there is no GPU at all)
    for shown in (2, 8, 0):
        err = io.StringIO()
        assert launch.spawn_ranks([sys.executable, "-c", "raise SystemExit(0)"], 2, visible_devices=shown, err=err, share_gpu=True) == 2
        assert "--share-gpu is the one-GPU test mode" in err.getvalue()
    env = launch.rank_environment(1, 2, 1, base={}, init_method="file:///x/y", share_gpu=True)
    assert env[launch.INIT_VAR] == "file:///x/y" and env[launch.SHARE_VAR] == "1" and launch.init_method_of(env) == "file:///x/y"
    assert launch.init_method_of({}) is None
    env = launch.rank_environment(3, 8, 12345, base={})
    assert (env["RANK"], env["LOCAL_RANK"], env["WORLD_SIZE"], env["MASTER_ADDR"], env["MASTER_PORT"]) == ("3", "3", "8", "127.0.0.1", "12345")
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and launch.under_a_launcher(env) and not launch.under_a_launcher({})


def test_bench_refuses_more_ranks_than_devices_before_touching_a_gpu():
    """``python bench.py --gpus 2`` on a box without two GPUs: one line on stderr, status 2, no rank started."""
    import subprocess
    import torch as _t
    if _t.cuda.device_count() >= 2:
        pytest.skip("two devices visible")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--no-cpu"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 2 and "needs 2 visible GPUs" in r.stderr and r.stdout.strip() == ""


def test_launcher_passes_a_signal_on_and_reports_it():
    """A driver that ends ``python bench.py --gpus N`` on a time-out sends the PARENT a SIGTERM: the launcher passes it on to its
    ranks, gives them ``grace_s`` to leave (they hold GPUs and an RCCL group), and returns 128 + the signal as a shell would
    (ADVICE r04); nothing of the job is left behind."""
    import signal
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, os\nsys.path.insert(0, %r)\nfrom pgmuvi_amd import launch\nprint('ready', flush=True)\n"
            "rc = launch.spawn_ranks([sys.executable, '-c', 'import time; time.sleep(120)'], 2, grace_s=5.0, out=sys.stderr)\n"
            "print('rc', rc, flush=True)\nsys.exit(rc)\n" % root)
    p = subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, text=True)
    try:
        assert p.stdout.readline().strip() == "ready"
        time.sleep(1.0)
        kids = [d for d in os.listdir("/proc") if d.isdigit() and _parent_of(d) == str(p.pid)]
        assert len(kids) == 2
        t0 = time.time()
        p.send_signal(signal.SIGTERM)
        out, _ = p.communicate(timeout=30)
        assert p.returncode == 128 + signal.SIGTERM and "rc 143" in out and time.time() - t0 < 10
        time.sleep(0.3)
        assert not [k for k in kids if os.path.exists(f"/proc/{k}")]
    finally:
        if p.poll() is None:
            p.kill()


def _parent_of(pid):
    try:
        return open(f"/proc/{pid}/stat").read().rsplit(")", 1)[1].split()[1]
    except OSError:
        return ""
