"""The N>1 path on CPU: world_size-2 ``gloo`` processes shard a batch of light curves,
evaluate their shards (oracle stand-in for the HIP call: no GPU here) and all_gather the
log-likelihoods; the result must equal the single-process evaluation of the whole batch."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from pgmuvi_amd import synthetic as syn  # noqa: E402
from pgmuvi_amd.batch import balanced_assignment, evaluate_batch, gather_logliks, shard_bounds  # noqa: E402

B, N = 5, 40


def _batch():
    xs, ys, ns, ws, mus, vs, ms = [], [], [], [], [], [], []
    for i in range(B):
        (t, y, e), per = syn.cfg3_lightcurve(i, n_obs=N)
        h = syn.cfg_hypers(3, y.double(), lead_period=per)
        xs.append(t.double().reshape(N, 1)); ys.append(y.double()); ns.append(e.double() ** 2)
        ws.append(h["w"]); mus.append(h["mu"].reshape(4, 1)); vs.append(h["v"].reshape(4, 1)); ms.append(h["mean"].expand(N))
    return tuple(torch.stack(L) for L in (xs, ys, ms, ns, ws, mus, vs))


def _worker(rank, world, port, q):
    import _oracle_backend as ob
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x, y, m, nz, w, mu, v = _batch()
    lo, hi = shard_bounds(B, rank, world)
    out = evaluate_batch(x[lo:hi], y[lo:hi], m[lo:hi], nz[lo:hi], w[lo:hi], mu[lo:hi], v[lo:hi], _compute=ob.mll_value_grad)
    ll = gather_logliks(out["mll"], B)
    gw = gather_logliks(out["g_w"], B)
    eq = gather_logliks(torch.full((2, 3), float(rank), dtype=torch.float64), 2 * world)   # equal shards: single-collective path
    assert eq.shape == (2 * world, 3) and eq[:, 0].tolist() == [float(r) for r in range(world) for _ in range(2)]
    q.put((rank, ll, gw))
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_shard_and_gather_equals_single_process():
    import _oracle_backend as ob
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    x, y, m, nz, w, mu, v = _batch()
    full = evaluate_batch(x, y, m, nz, w, mu, v, _compute=ob.mll_value_grad)
    for rank, ll, gw in got:
        assert ll.shape == (B,) and torch.equal(ll, full["mll"])       # every rank holds the whole vector
        assert torch.equal(gw, full["g_w"])


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
