"""One rank of a job whose ranks SHARE device 0 (test mode of a one-GPU box; ``pgmuvi_amd.launch.spawn_ranks(share_gpu=True)``
starts it): the HIP path in several processes of one job.  The batch is sharded and evaluated exactly as ``bench.py
--total-batch`` does (``make_shard`` + ``sharded_batch_step``), then a ragged batch (``make_ragged_shard`` +
``sharded_ragged_step``); the process group is gloo -- RCCL cannot put two ranks on one device -- and the log-likelihoods are
staged through host memory for the collective (``pgmuvi_amd.batch._staged_on_host``).  Every rank writes what it ends up with
to ``--outdir``; rank 0 prints ONE JSON line.  No oracle here: the test compares with the one-process HIP result."""
import argparse
import datetime
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=2)
    ap.add_argument("--total-batch", type=int, default=16)
    ap.add_argument("--npoints", type=int, default=512)
    ap.add_argument("--chunk", type=int, default=8)
    ap.add_argument("--ragged", type=int, nargs=3, default=None, metavar=("TOTAL", "N_LO", "N_HI"))
    ap.add_argument("--chains", type=int, nargs=2, default=None, metavar=("CHAINS", "N"),
                    help="config 5's layout instead: CHAINS chains, each on its own light curve of N points, dealt over the ranks")
    ap.add_argument("--outdir", required=True)
    ap.add_argument("--fail-rank", type=int, default=-1, help="this rank leaves with status 3 AFTER its first HIP evaluation")
    args = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import torch
    import torch.distributed as dist
    from pgmuvi_amd import _hip, launch
    from pgmuvi_amd.batch import make_ragged_shard, make_shard, sharded_batch_step, sharded_ragged_step
    assert os.environ.get(launch.SHARE_VAR) == "1" and torch.cuda.device_count() == 1
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", init_method=launch.init_method_of(), rank=rank, world_size=world,
                            timeout=datetime.timedelta(seconds=120))
    if args.chains:
        return chains(args, rank, world, dev, dist, _hip)
    shard = make_shard(args.total_batch, rank, world, args.npoints, "cfg3", dev)
    if rank == args.fail_rank:
        out, _ = sharded_batch_step(shard, args.total_batch, args.chunk, group=False)     # the HIP call, no collective
        torch.cuda.synchronize()
        assert int(out["info"].abs().max()) == 0
        sys.stderr.write(f"rank {rank}: leaving with status 3 after {out['mll'].numel()} HIP evaluations\n")
        sys.exit(3)                                              # the others now wait in the all_gather
    out, ll = sharded_batch_step(shard, args.total_batch, args.chunk)
    torch.cuda.synchronize()
    keep = {"ll": ll.cpu(), "info": out["info"].cpu(), "nloc": shard["y"].shape[0]}
    for k in ("g_w", "g_mu", "g_v", "g_noise", "g_mean"):
        keep[k] = out[k].cpu()
    if args.ragged:
        total, lo, hi = args.ragged
        rs = make_ragged_shard(total, rank, world, lo, hi, device=dev)
        rout, rll = sharded_ragged_step(rs, device=dev)
        torch.cuda.synchronize()
        keep.update(ragged_ll=rll.cpu(), ragged_index=rs["index"], ragged_g_w=rout["g_w"].cpu() if rs["index"] else None)
    torch.save(keep, os.path.join(args.outdir, f"rank{rank}.pt"))
    seen = [None] * world
    dist.all_gather_object(seen, (rank, os.getpid(), int(shard["y"].shape[0]), torch.cuda.current_device()))
    dist.barrier()
    if rank == 0:
        print(json.dumps({"world": world, "ranks": seen, "lib": os.path.basename(_hip.lib_path()), "backend": dist.get_backend()}))
    dist.destroy_process_group()


def chain_problem(C, n):
    """C seeded light curves of n points (the generator of config 3) and the start every chain is given."""
    import numpy as np
    import torch
    from pgmuvi_amd import synthetic as syn
    xs, ys, ns = [], [], []
    for c in range(C):
        (t, y, e), _ = syn.cfg3_lightcurve(7000 + c, n_obs=n)
        xs.append(t.double().reshape(n, 1)); ys.append(y.double()); ns.append(e.double() ** 2)
    init = {"mean_module.mean_prior": np.array(0.0), "covar_module.mixture_weights_prior": np.array([0.5, 0.1]),
            "covar_module.mixture_means_prior": np.array([1 / 150.0, 1 / 67.0]).reshape(2, 1, 1),
            "covar_module.mixture_scales_prior": np.array([1 / 1500.0, 1 / 700.0]).reshape(2, 1, 1)}
    kw = dict(num_mixtures=2, num_samples=5, warmup_steps=5, seed=5, group_by_chain=True, max_tree_depth=3, initial_values=init)
    return torch.stack(xs), torch.stack(ys), torch.stack(ns), kw


def chains(args, rank, world, dev, dist, _hip):
    """``mcmc.run_mcmc`` as a rank of the job: its block of the chains through the native potential (``pgm_pot_*``), one gather
    of the draws at the end."""
    import torch
    from pgmuvi_amd import mcmc
    C, n = args.chains
    x, y, nz, kw = chain_problem(C, n)
    out = mcmc.run_mcmc(x.to(dev), y.to(dev), nz.to(dev), **kw)
    torch.cuda.synchronize()
    keep = {k: v for k, v in out.items() if k != "_diagnostics"}
    keep.update(n_leapfrog=out["_diagnostics"]["n_leapfrog"], potential_energy=out["_diagnostics"]["potential_energy"],
                step_size=out["_diagnostics"]["step_size"])
    torch.save(keep, os.path.join(args.outdir, f"chains_rank{rank}.pt"))
    seen = [None] * world
    dist.all_gather_object(seen, (rank, os.getpid()))
    dist.barrier()
    if rank == 0:
        print(json.dumps({"world": world, "ranks": seen, "lib": os.path.basename(_hip.lib_path()), "backend": dist.get_backend(),
                          "chains": C}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
