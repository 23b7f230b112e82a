"""Many-light-curve / many-chain batches (configs 3 and 5): shard, evaluate, gather.

Light curves are independent, so a batch shards over GPUs with no data-path
collective: rank r evaluates a contiguous block of the batch on its own device (all
problems of a shard advance together through the batched C-ABI entry point, batch on
gridDim.z), and one ``all_gather`` of the per-curve log-likelihoods (+ optionally the
gradients) over RCCL/xGMI closes the step.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from . import _hip


def shard_bounds(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block partition: the first ``total % world`` ranks get one extra."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def balanced_assignment(costs: Sequence[float], world: int):
    """Ragged N: sort by cost (N^3) and deal to the currently lightest rank (longest processing time first); ties go to the
    lower rank / lower index, so every rank computes the same table.  Used by :func:`make_ragged_shard`."""
    order = sorted(range(len(costs)), key=lambda i: -costs[i])
    load = [0.0] * world
    owner = [0] * len(costs)
    for i in order:
        r = min(range(world), key=lambda k: load[k])
        owner[i] = r
        load[r] += costs[i]
    return owner


def evaluate_batch(x, y, mean, noise, w, mu, v, noise_scalar=None, dim_order=0, need_grad=True, chunk: Optional[int] = None,
                   _compute=None) -> Dict[str, torch.Tensor]:
    """MLL (+ gradients) of ``B`` equal-length light curves on the local device.

    x (B,N,d) y (B,N) mean (B,N) noise (B,N)|None w (B,Q) mu (B,Q,d) v (B,Q,d).  ``chunk``
    bounds how many problems share the workspace at once (memory = chunk * 8 N^2 bytes).
    """
    compute = _compute or _hip.mll_value_grad
    B = y.shape[0]
    chunk = B if chunk is None else max(1, min(chunk, B))
    outs = []
    for lo in range(0, B, chunk):
        hi = min(B, lo + chunk)
        sl = slice(lo, hi)
        o = compute(x[sl], y[sl], mean[sl], None if noise is None else noise[sl],
                    None if noise_scalar is None else noise_scalar[sl], w[sl], mu[sl], v[sl], dim_order, 0.0, need_grad)
        outs.append(o)
    keys = ["mll", "info"] + (["g_w", "g_mu", "g_v", "g_noise", "g_mean"] if need_grad else [])
    return {k: torch.cat([o[k] for o in outs]) for k in keys}


def pad_curves(curves: Sequence[Dict[str, torch.Tensor]], device=None) -> Tuple[Dict[str, torch.Tensor], List[int]]:
    """Light curves of different lengths -> the padded arrays of the ragged entry point.  Each curve is a dictionary with
    x (N_i, d) | (N_i,), y (N_i,), noise (N_i,), mean () | (N_i,), w (Q,), mu (Q, d) | (Q,), v (Q, d) | (Q,).  Returns
    ({x (B,S,d), y, mean, noise (B,S), w (B,Q), mu, v (B,Q,d)}, lengths) with S = the longest light curve; the padding
    entries are zero (never read by the library)."""
    B = len(curves)
    lengths = [int(c["y"].shape[0]) for c in curves]
    S = max(lengths)
    d = 1 if curves[0]["x"].dim() == 1 else int(curves[0]["x"].shape[1])
    q = int(curves[0]["w"].numel())
    D = torch.float64
    out = dict(x=torch.zeros(B, S, d, dtype=D), y=torch.zeros(B, S, dtype=D), mean=torch.zeros(B, S, dtype=D),
               noise=torch.zeros(B, S, dtype=D), w=torch.zeros(B, q, dtype=D), mu=torch.zeros(B, q, d, dtype=D),
               v=torch.zeros(B, q, d, dtype=D))
    for b, c in enumerate(curves):
        n = lengths[b]
        out["x"][b, :n] = c["x"].to(D).reshape(n, d).cpu()
        out["y"][b, :n] = c["y"].to(D).cpu()
        out["noise"][b, :n] = c["noise"].to(D).cpu()
        out["mean"][b, :n] = c["mean"].to(D).cpu().expand(n)
        out["w"][b] = c["w"].to(D).reshape(q).cpu()
        out["mu"][b] = c["mu"].to(D).reshape(q, d).cpu()
        out["v"][b] = c["v"].to(D).reshape(q, d).cpu()
    if device is not None:
        out = {k: t.to(device) for k, t in out.items()}
    return out, lengths


_twin_workspaces: Dict[tuple, list] = {}


def _drop_twins():
    _twin_workspaces.clear()


_hip.on_release(_drop_twins)                                     # (``_hip.release_workspaces()`` gives these back as well)


def _two_streams(padded, lengths, dim_order, need_grad, max_batch, streams):
    """Two launch sets at a time: a set of a few dozen light curves leaves CUs idle in the latency-bound links of its sweep (its
    diagonal blocks are one workgroup per member), which another set's updates can use.  The sets (``pgm_ragged_plan``) are
    dealt alternately to two ragged calls, each on a stream and a workspace of its own (512 x N ~ U{1024..2048}: 58.2 -> 54.1 ms
    per pass on one MI355X; the values do not depend on it).  None when there is nothing to overlap (``streams`` None: from
    four sets on) -- or when the pair of workspaces does not fit: the caller then runs the sets one after the other.

    ``max_batch`` is the caller's memory bound for ONE workspace (``default_chunk``): each of the two gets half of it, so the
    pair holds what the one-stream path would."""
    set_of, nbs = _hip.ragged_plan(lengths, max_batch)
    if len(nbs) < (2 if streams == 2 else 4):
        return None
    dev = padded["y"].device
    halves = [[i for i, s_ in enumerate(set_of) if s_ % 2 == p] for p in (0, 1)]
    B, S = padded["y"].shape
    q, d = padded["w"].shape[-1], padded["x"].shape[-1]
    slots = max(1, min(max(len(h) for h in halves), (max_batch + 1) // 2))
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), (max(lengths) + 127) // 128 * 128, q, d, slots)
    if key not in _twin_workspaces:
        _twin_workspaces.clear()                                 # (one pair at a time: the previous batch shape's buffers go)
        made = []
        try:
            # (the cache's byte budget covers these two as well: what is cached beyond it goes first)
            _hip.trim_cache(reserve_bytes=2 * _hip.workspace_bytes_estimate(key[1], q, d, slots))
            for _ in range(2):
                made.append(_hip.Workspace(dev, key[1], q, d, slots))
        except RuntimeError:                                     # (no room for the pair: one stream, one workspace)
            for w in made:
                w.close()
            return None
        _twin_workspaces[key] = made + [[torch.cuda.Stream(device=dev) for _ in range(2)]]
    wss, sts = _twin_workspaces[key][:2], _twin_workspaces[key][2]
    cur = torch.cuda.current_stream(dev)
    parts = []
    for idx, ws, st in zip(halves, wss, sts):
        ix = torch.as_tensor(idx, device=dev)
        sub = {k: v.index_select(0, ix) for k, v in padded.items() if torch.is_tensor(v)}
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            parts.append((ix, _hip.mll_value_grad_ragged(sub["x"], sub["y"], sub["mean"], sub.get("noise"), sub.get("noise_scalar"),
                                                       [lengths[i] for i in idx], sub["w"], sub["mu"], sub["v"], dim_order, 0.0,
                                                       need_grad, workspace=ws)))
    for st in sts:
        cur.wait_stream(st)
    keys = ["mll", "info"] + (["g_w", "g_mu", "g_v", "g_noise", "g_mean"] if need_grad else [])
    out = {}
    for k in keys:
        first = parts[0][1][k]
        full = torch.zeros((B,) + tuple(first.shape[1:]), dtype=first.dtype, device=dev)
        for ix, o in parts:
            full.index_copy_(0, ix, o[k])
        out[k] = full
    out["_keep"] = parts                                         # (the halves' inputs stay alive until the results have been used)
    out["launch_sets"] = [nb for _, o in parts for nb in o["launch_sets"]]     # (what the two calls ran, not the plan they were dealt from)
    return out


def evaluate_ragged(curves=None, padded: Optional[Dict[str, torch.Tensor]] = None, lengths: Optional[Sequence[int]] = None,
                    dim_order=0, need_grad=True, chunk: Optional[int] = None, device=None, streams: Optional[int] = None,
                    _compute=None) -> Dict[str, object]:
    """MLL (+ gradients) of light curves of DIFFERENT lengths on the local device -- what a real many-light-curve batch is
    (every pgmuvi ``Lightcurve`` has its own N, ``/root/reference/pgmuvi/lightcurve.py:1724-1733, 2150-2181``).

    Either ``curves`` (a list of per-curve dictionaries, see :func:`pad_curves`) or the ``padded`` arrays with ``lengths``.
    The library sorts the light curves by their block-row count -- i.e. by N^3, SURVEY.md section 8e -- and runs them in
    launch sets that share a chain length (``pgm_mll_value_grad_ragged_f64``); ``chunk`` bounds the light curves per launch
    set (workspace memory, default :func:`default_chunk` of the longest); ``streams``: 1 = the sets one after the other on the
    current stream, 2 = alternately on two streams and two workspaces, None = two from four sets on.  Returns mll (B,), info (B,), g_w (B,Q), g_mu,
    g_v (B,Q,d) and g_noise / g_mean as lists of B vectors of each light curve's own length."""
    if padded is None:
        padded, lengths = pad_curves(curves, device=device)
    lengths = [int(n) for n in lengths]
    compute = _compute or _hip.mll_value_grad_ragged
    B = len(lengths)
    if B != padded["y"].shape[0] or min(lengths) < 1 or max(lengths) > padded["y"].shape[1]:
        raise ValueError(f"evaluate_ragged: {padded['y'].shape[0]} lengths in [1, {padded['y'].shape[1]}] expected")
    chunk = chunk or default_chunk(max(lengths), device=padded["y"].device)
    o = None
    if _compute is None and streams != 1 and padded["y"].is_cuda:
        o = _two_streams(padded, lengths, dim_order, need_grad, max(1, min(chunk, B)), streams)
    if o is None:
        o = compute(padded["x"], padded["y"], padded["mean"], padded.get("noise"), padded.get("noise_scalar"), lengths,
                    padded["w"], padded["mu"], padded["v"], dim_order, 0.0, need_grad, max_batch=max(1, min(chunk, B)))
    out = {"mll": o["mll"], "info": o["info"], "lengths": lengths}
    if "launch_sets" in o:
        out["launch_sets"] = list(o["launch_sets"])              # block rows of the launch sets that ran
    if need_grad:
        out.update(g_w=o["g_w"], g_mu=o["g_mu"], g_v=o["g_v"])
        out["g_noise"] = [o["g_noise"][b, :n] for b, n in enumerate(lengths)]
        out["g_mean"] = [o["g_mean"][b, :n] for b, n in enumerate(lengths)]
    return out


def ragged_lengths(total: int, n_lo: int, n_hi: int, seed: int = 4) -> List[int]:
    """Lengths of a synthetic ragged batch: N_i ~ U{n_lo .. n_hi} from ``default_rng(seed)`` (a function of i only)."""
    import numpy as np
    return [int(v) for v in np.random.default_rng(seed).integers(n_lo, n_hi + 1, size=total)]


def make_ragged_shard(total: int, rank: int, world: int, n_lo: int, n_hi: int, seed: int = 4, device=None) -> Dict[str, object]:
    """This rank's light curves of a synthetic ragged batch: light curve i (the cfg-3 recipe at its own length N_i) depends on
    (i, N_i) only; the owner of every light curve comes from :func:`balanced_assignment` on the costs N_i^3 (SURVEY.md
    section 8e: "for ragged N sort by N^3 and deal round-robin") -- the same table on every rank."""
    from . import synthetic as syn
    lengths = ragged_lengths(total, n_lo, n_hi, seed)
    owner = balanced_assignment([float(n) ** 3 for n in lengths], world)
    mine = [i for i in range(total) if owner[i] == rank]
    curves = []
    for i in mine:
        (t, y, e), per = syn.cfg3_lightcurve(i, n_obs=lengths[i])
        h = syn.cfg_hypers(3, y.double(), lead_period=per)
        curves.append(dict(x=t.double(), y=y.double(), noise=e.double() ** 2, mean=h["mean"], w=h["w"], mu=h["mu"], v=h["v"]))
    padded, lens = pad_curves(curves, device=device) if curves else ({}, [])
    return dict(index=mine, owner=owner, lengths=lengths, curves=curves, padded=padded, local_lengths=lens)


def sharded_ragged_step(shard: Dict[str, object], need_grad=True, chunk: Optional[int] = None, group=None, device=None,
                        _compute=None) -> Tuple[Dict[str, object], torch.Tensor]:
    """One evaluation of a ragged batch whose light curves are dealt to the ranks by cost (see :func:`make_ragged_shard`):
    each rank evaluates its own light curves, then ONE all_gather of the log-likelihoods.  Returns (local outputs, the
    log-likelihoods of the whole batch in the batch's own order, identical on every rank)."""
    total = len(shard["owner"])
    if shard["index"]:
        out = evaluate_ragged(padded=shard["padded"], lengths=shard["local_lengths"], need_grad=need_grad, chunk=chunk,
                              _compute=_compute)
        local = out["mll"]
    else:
        dev = device if device is not None else "cpu"
        out = dict(mll=torch.zeros(0, dtype=torch.float64, device=dev), info=torch.zeros(0, dtype=torch.int32, device=dev))
        local = out["mll"]
    if group is False:
        return out, local
    return out, gather_by_owner(local, shard["owner"], group=group)


def _staged_on_host(local: torch.Tensor, group=None) -> bool:
    """Device values over a process group that has no device collectives (``gloo``: the test-only mode in which several ranks
    share ONE GPU, ``bench.py --share-gpu``; RCCL cannot put two ranks on one device): the few KB of log-likelihoods go
    through host memory for the collective and come back to the device afterwards.  Never the case on RCCL ("nccl")."""
    return local.is_cuda and dist.get_backend(group) == "gloo"


def gather_by_owner(local: torch.Tensor, owner: Sequence[int], group=None) -> torch.Tensor:
    """all_gather of per-curve values when light curve i lives on rank owner[i] (each rank holds its own in ascending i):
    one collective of equal-sized padded buffers, then every value goes to its place in the batch's order."""
    total = len(owner)
    if not (dist.is_available() and dist.is_initialized()):
        return local
    if _staged_on_host(local, group):
        return gather_by_owner(local.cpu(), owner, group=group).to(local.device)
    world = dist.get_world_size(group)
    index = [[i for i in range(total) if owner[i] == r] for r in range(world)]
    width = max(1, max(len(ix) for ix in index))
    pad = torch.zeros((width,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    buf = torch.empty((world * width,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf, pad, group=group)
    out = torch.empty((total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    for r, ix in enumerate(index):
        if ix:
            out[torch.as_tensor(ix, device=local.device)] = buf[r * width: r * width + len(ix)]
    return out


def make_shard(total: int, rank: int, world: int, n: int, recipe: str = "cfg3", device=None) -> Dict[str, torch.Tensor]:
    """This rank's contiguous block of a synthetic ``total``-light-curve batch: light curve i of the
    batch depends on (recipe, i, n) only -- never on the partition -- so any world size evaluates the same ``total`` problems.
    ``recipe`` "cfg3": leading period ~ U(30, 300) per light curve (BASELINE config 3); "cfg2": config 2's periods, seed 2 + i."""
    from . import synthetic as syn
    lo, hi = shard_bounds(total, rank, world)
    cols = {k: [] for k in ("x", "y", "mean", "noise", "w", "mu", "v")}
    for i in range(lo, hi):
        if recipe == "cfg3":
            (t, y, e), per = syn.cfg3_lightcurve(i, n_obs=n)
            h = syn.cfg_hypers(3, y.double(), lead_period=per)
        else:
            t, y, e = syn.cfg2(n_obs=n, seed=2 + i)
            h = syn.cfg_hypers(2, y.double())
        q = h["w"].numel()
        cols["x"].append(t.double().reshape(n, 1)); cols["y"].append(y.double()); cols["noise"].append(e.double() ** 2)
        cols["mean"].append(h["mean"].expand(n)); cols["w"].append(h["w"])
        cols["mu"].append(h["mu"].reshape(q, 1)); cols["v"].append(h["v"].reshape(q, 1))
    if lo == hi:                                                  # more ranks than light curves: an empty shard of the right shapes
        z = lambda *s: torch.zeros(s, dtype=torch.float64)
        out = dict(x=z(0, n, 1), y=z(0, n), mean=z(0, n), noise=z(0, n), w=z(0, 4), mu=z(0, 4, 1), v=z(0, 4, 1))
    else:
        out = {k: torch.stack(v).contiguous() for k, v in cols.items()}
    return out if device is None else {k: v.to(device) for k, v in out.items()}


def default_chunk(n: int, budget_bytes: float = 40e9, device=None) -> int:
    """Light curves per launch set of a shard: as many as fit ``budget_bytes`` of workspace (8 N^2 bytes each and ~15 % of
    side buffers), at most 512.  More light curves per launch set means fuller launches of the latency-bound links of the
    sweep (a diagonal-block launch holds one workgroup per light curve): 64 x N=2048 per call 5020 evaluations/s, 128: 5310,
    256: 5440, 512: 5500; N=4096: 790 / 807 / 807 (one MI355X, round 3).  With a ``device`` the budget is also held to 60 %
    of the memory that is free there now (a smaller or shared GPU, or another large workspace still cached)."""
    np_ = (n + 127) // 128 * 128
    if device is not None and torch.cuda.is_available() and torch.device(device).type == "cuda":
        try:
            budget_bytes = min(budget_bytes, 0.6 * torch.cuda.mem_get_info(device)[0])
        except Exception:
            pass
    c = int(max(1, min(512, budget_bytes // (9.2 * np_ * np_))))
    return c - c % 8 if c >= 8 else c          # (multiples of 8 keep the XCD-aware placement of a batch's workgroups)


def sharded_batch_step(shard: Dict[str, torch.Tensor], total: int, chunk: Optional[int] = None, need_grad=True, group=None,
                       _compute=None) -> Tuple[Dict[str, torch.Tensor], torch.Tensor]:
    """One evaluation of a ``total``-light-curve batch whose local block is ``shard`` (see :func:`make_shard`): the shard in
    memory-bounded chunks through the batched entry point, then ONE all_gather of the log-likelihoods (RCCL on GPUs).  Returns
    (local outputs, the length-``total`` log-likelihood vector, identical on every rank)."""
    nloc = shard["y"].shape[0]
    if nloc:
        out = evaluate_batch(shard["x"], shard["y"], shard["mean"], shard["noise"], shard["w"], shard["mu"], shard["v"],
                             need_grad=need_grad, chunk=chunk or default_chunk(shard["y"].shape[1], device=shard["y"].device), _compute=_compute)
    else:
        out = dict(mll=torch.zeros(0, dtype=torch.float64, device=shard["y"].device),
                   info=torch.zeros(0, dtype=torch.int32, device=shard["y"].device))
    if group is False:                                           # local only (warm-up of a chunk shape): no collective
        return out, out["mll"]
    return out, gather_logliks(out["mll"], total, group=group)


def gather_logliks(local: torch.Tensor, total: int, group=None) -> torch.Tensor:
    """all_gather of the per-curve values of every rank's shard (block partition) into
    one length-``total`` vector, identical on every rank.  Single process: identity."""
    if not (dist.is_available() and dist.is_initialized()):
        return local
    if _staged_on_host(local, group):
        return gather_logliks(local.cpu(), total, group=group).to(local.device)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    counts = [shard_bounds(total, r, world)[1] - shard_bounds(total, r, world)[0] for r in range(world)]
    if all(c == counts[0] for c in counts):
        # equal shards (the usual case): one collective straight into the result, no padding, no list of buffers
        out = torch.empty((total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    width = max(counts)
    pad = torch.zeros((width,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: counts[rank]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([b[:c] for b, c in zip(bufs, counts)])
