"""A stand-in for a multi-rank entry point, run by the launcher test: started as ONE plain process with
--gpus N it relaunches itself as N gloo ranks (the path bench.py takes with RCCL), reduces a value over
the ranks and has rank 0 print one JSON line."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from mad_amd.launch import needs_relaunch, relaunch_per_gpu  # noqa: E402


def main():
    n = int(sys.argv[sys.argv.index("--gpus") + 1])
    if needs_relaunch(n):
        raise SystemExit(relaunch_per_gpu(os.path.abspath(__file__), sys.argv[1:], n))
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert world == n
    if "--fail-rank" in sys.argv and rank == int(sys.argv[sys.argv.index("--fail-rank") + 1]):
        raise SystemExit(3)
    dist.init_process_group("gloo")
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    s = torch.tensor([1.0], dtype=torch.float64)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    if rank == 0:
        print(json.dumps({"n_gpus": world, "max": float(t.item()), "ranks_seen": int(s.item())}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
