"""Identity of the native sources: one hash over everything libmad_hip.so is built from.

The Makefile embeds ``source_build_id()`` into the library (``mad_build_id()``); the test session and the benchmark
compare the embedded value with the hash of the sources lying next to the library, so a stale binary is refused
instead of silently tested or measured.  Run as a script it prints the id (that is what the Makefile calls).
"""
from __future__ import annotations

import glob
import hashlib
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_HEADER = os.path.join(os.path.dirname(_HERE), "include", "mad_hip.h")


def source_files():
    """Every file the build reads: kernels, shared headers, the host glue, the public header and the Makefile
    (which holds the compiler flags)."""
    files = []
    for pat in ("*.hip", "*.h", "*.cpp", "Makefile"):
        files += glob.glob(os.path.join(_CSRC, pat))
    return sorted(files) + [_HEADER]


def source_build_id() -> str:
    h = hashlib.sha256()
    for f in source_files():
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
        h.update(b"\0")
    return h.hexdigest()[:16]


def library_build_id() -> str:
    from . import _lib
    return _lib.lib.mad_build_id().decode()


def check_library_is_current() -> str:
    """The id, or RuntimeError when the loaded library was built from other sources than the ones beside it."""
    want, got = source_build_id(), library_build_id()
    if want != got:
        from . import _lib
        raise RuntimeError(f"{_lib.LIB_PATH} was built from other sources (library {got}, sources {want}): "
                           "rebuild it with `make -C mad_amd/csrc`")
    return got


if __name__ == "__main__":
    print(source_build_id())
