"""One rank of a CPU (gloo) job started by ``pgmuvi_amd.launch.spawn_ranks`` -- the test twin of what ``bench.py --gpus N``
starts on GPUs: rank, world size and rendezvous come from the environment the launcher made, the batch is sharded and
evaluated as ``bench.py --total-batch`` does (``make_shard`` + ``sharded_batch_step``), with the oracle stand-in for the HIP
call (no GPU here).  Rank 0 prints ONE JSON line."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--total-batch", type=int, default=5)
    ap.add_argument("--npoints", type=int, default=40)
    ap.add_argument("--fail-rank", type=int, default=-1)
    args = ap.parse_args()
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    if rank == args.fail_rank:
        sys.exit(3)                                              # before the rendezvous: the other ranks would wait for ever
    import torch
    import torch.distributed as dist
    import _oracle_backend as ob
    from pgmuvi_amd import launch
    from pgmuvi_amd.batch import make_shard, sharded_batch_step
    dist.init_process_group("gloo", init_method=launch.init_method_of(), rank=rank, world_size=world)   # the launcher's file rendezvous
    shard = make_shard(args.total_batch, rank, world, args.npoints, "cfg3")
    out, ll = sharded_batch_step(shard, args.total_batch, 2, _compute=ob.mll_value_grad)
    seen = [None] * world
    dist.all_gather_object(seen, (rank, local, shard["y"].shape[0]))
    dist.barrier()
    if rank == 0:
        print(json.dumps({"world": world, "gpus_arg": args.gpus, "ranks": seen, "loglik": ll.tolist(),
                          "master": [os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"]], "init": launch.init_method_of(), "child_mark": os.environ.get("PGM_LAUNCH_CHILD")}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
