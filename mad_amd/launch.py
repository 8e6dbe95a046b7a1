"""Start one process per GPU for a script that was invoked as a single plain process.

The multi-GPU entry points (``bench.py``, ``bench_train.py``) follow the one-process-per-GPU model of the
reference's workers (``/root/reference/generate_translation.py:100-119``: one joblib worker per
``cuda:{i}``).  When such a script is started as ``python script.py --gpus N`` with no rendezvous
variables in the environment, ``relaunch_per_gpu`` runs ``torch.distributed.run`` as a CHILD process
(N ranks on 127.0.0.1, a free port) and returns its exit code; the parent never touches the GPU, so
nothing that has initialised a device is ever replaced by another program.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
from typing import List, Optional, Sequence


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def needs_relaunch(n_ranks: int, environ=None) -> bool:
    """True when N > 1 ranks are asked for and this process is not already one of them."""
    env = os.environ if environ is None else environ
    return n_ranks > 1 and "RANK" not in env and int(env.get("WORLD_SIZE", "1")) == 1


def relaunch_command(script: str, argv: Sequence[str], n_ranks: int, port: Optional[int] = None) -> List[str]:
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
            "--master-addr", "127.0.0.1", "--master-port", str(port or free_port()), script, *argv]


def relaunch_per_gpu(script: str, argv: Sequence[str], n_ranks: int, extra_env: Optional[dict] = None) -> int:
    """Run ``script argv`` as ``n_ranks`` ranks under torch.distributed.run; stdout/stderr pass through
    (rank 0 prints the result line).  Returns the launcher's exit code."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this driver
    if extra_env:
        env.update(extra_env)
    return subprocess.run(relaunch_command(script, argv, n_ranks), env=env).returncode
