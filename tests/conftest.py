import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def golden():
    return load_golden


def rel_err(a, b):
    """max |a-b| / max |b| (the relative error used for denoiser parity)."""
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def _ensure_built():
    """The HIP library is a build artefact (not in git): compile it when a fresh tree lacks it."""
    so = os.path.join(REPO, "mad_amd", "libmad_hip.so")
    if not os.path.exists(so):
        import subprocess
        subprocess.run(["make", "-C", os.path.join(REPO, "mad_amd", "csrc"), "-j8"], check=True)


_ensure_built()


def pytest_sessionstart(session):
    """Refuse to test a stale binary: the id embedded in libmad_hip.so must be the hash of the native sources in
    this tree (mad_amd/buildid.py)."""
    from mad_amd.buildid import check_library_is_current
    try:
        bid = check_library_is_current()
    except RuntimeError as e:
        pytest.exit(str(e), returncode=3)
    print(f"libmad_hip.so build id {bid} matches the sources")
