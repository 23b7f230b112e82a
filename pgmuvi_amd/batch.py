"""Many-light-curve / many-chain batches (configs 3 and 5): shard, evaluate, gather.

Light curves are independent, so a batch shards over GPUs with no data-path
collective: rank r evaluates a contiguous block of the batch on its own device (all
problems of a shard advance together through the batched C-ABI entry point, batch on
gridDim.z), and one ``all_gather`` of the per-curve log-likelihoods (+ optionally the
gradients) over RCCL/xGMI closes the step.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence, Tuple

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
    """Ragged N: sort by cost (N^3) and deal to the currently lightest rank (LPT)."""
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


def gather_logliks(local: torch.Tensor, total: int, group=None) -> torch.Tensor:
    """all_gather of the per-curve values of every rank's shard (block partition) into
    one length-``total`` vector, identical on every rank.  Single process: identity."""
    if not (dist.is_available() and dist.is_initialized()):
        return local
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
