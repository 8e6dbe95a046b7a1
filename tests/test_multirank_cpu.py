"""The N>1 path on CPU: two gloo ranks shard a list as the reference does, with no data-path
collective; only the timing reduction (max over ranks) and the item count use the process group."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mad_amd.shard import job_throughput, shard_list


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, items, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine, offset = shard_list(items, world, rank)
    # "translate" the shard: independent items, written under their global index
    for i, it in enumerate(mine):
        with open(os.path.join(out_dir, f"pred_{i + offset}.txt"), "w") as f:
            f.write(str(it * 2))
    dist.barrier()
    total, secs, rate = job_throughput(len(mine), 1.0 + rank, None)     # rank 1 is "slower"
    assert total == len(items) and secs == float(world) and abs(rate - len(items) / world) < 1e-9
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_ranks_cover_the_list(tmp_path):
    items = list(range(7))
    mp.spawn(_worker, args=(2, _free_port(), items, str(tmp_path)), nprocs=2, join=True)
    for i in items:
        assert open(tmp_path / f"pred_{i}.txt").read() == str(i * 2)
    assert len(list(tmp_path.iterdir())) == len(items)


def test_single_process_needs_no_group():
    assert job_throughput(8, 2.0) == (8, 2.0, 4.0)


def _grad_worker(rank, world, port):
    """The training step's only collective: the bucketed sum of the flat gradient buffer."""
    from mad_amd.train import allreduce_flat, warmup_lr
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 1000
    flat = torch.arange(n, dtype=torch.float32) * (rank + 1)
    calls = allreduce_flat(flat, bucket_elems=300)                    # 4 ranges, the last one ragged
    assert calls == 4
    assert torch.equal(flat, torch.arange(n, dtype=torch.float32) * sum(r + 1 for r in range(world)))
    assert warmup_lr(1e-4, 0, 1000) == 0.0 and warmup_lr(1e-4, 500, 1000) == 5e-5 and warmup_lr(1e-4, 1000, 1000) == 1e-4
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_ranks_sum_the_flat_gradient():
    mp.spawn(_grad_worker, args=(2, _free_port()), nprocs=2, join=True)


# ---- the self-launch path of the benchmarks: `python script.py --gpus N` as ONE plain process
def test_relaunch_is_only_for_a_plain_process():
    from mad_amd.launch import needs_relaunch, relaunch_command
    assert needs_relaunch(2, environ={})
    assert not needs_relaunch(1, environ={})
    assert not needs_relaunch(2, environ={"RANK": "0", "WORLD_SIZE": "2"})        # already a rank
    cmd = relaunch_command("bench.py", ["--gpus", "4", "--steps", "2"], 4, port=29999)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-5:] == ["bench.py", "--gpus", "4", "--steps", "2"]


@pytest.mark.timeout(300)
def test_plain_process_starts_its_own_ranks():
    import json
    import subprocess
    import sys
    probe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rank_probe.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, probe, "--gpus", "2"], env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    assert json.loads(lines[0]) == {"n_gpus": 2, "max": 2.0, "ranks_seen": 2}
    # a rank that dies makes the launcher, and so the plain process, exit non-zero
    r = subprocess.run([sys.executable, probe, "--gpus", "2", "--fail-rank", "1"], env=env, capture_output=True,
                       text=True, timeout=280)
    assert r.returncode != 0
