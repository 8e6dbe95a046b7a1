#!/usr/bin/env python3
"""Interleaved A/B of library builds on whole denoiser evaluations, in ONE process (GPU only).

    python tools/ab_eval.py [--config CFG] [--batch B] [--rounds R] [--evals K] variant...

"base" is mad_amd/libmad_hip.so, any other name mad_amd/libmad_hip_<name>.so (tools/build_variant.sh); a variant
may carry environment settings for its launches, "name:VAR=VALUE,VAR2=VALUE2" (the library's tuning switches).  Every
library is loaded side by side; each round runs K evaluations with each of them in turn (so clock and temperature
drift hit all variants alike) and the summary gives median and minimum ms per evaluation over the rounds.
"""
import argparse, ctypes, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mad_amd import _lib, ops
from mad_amd.config import load_cfg
from mad_amd.model import build_model

p = argparse.ArgumentParser()
p.add_argument("--config", default=os.path.join(ROOT, "configs/model_256_256.yaml"))
p.add_argument("--batch", type=int, default=32)
p.add_argument("--rounds", type=int, default=5)
p.add_argument("--evals", type=int, default=10)
p.add_argument("variants", nargs="+")
a = p.parse_args()


def load(name):
    name = name.split(":")[0]
    path = os.path.join(ROOT, "mad_amd", "libmad_hip.so" if name == "base" else f"libmad_hip_{name}.so")
    lib = ctypes.CDLL(path)
    for fn, (res, args) in _lib.SIGNATURES.items():
        if hasattr(lib, fn):
            f = getattr(lib, fn); f.restype = res; f.argtypes = args
    return lib


libs = {v: load(v) for v in a.variants}
envs = {v: dict(kv.split("=", 1) for kv in v.split(":", 1)[1].split(",")) if ":" in v else {} for v in a.variants}
all_vars = sorted({k for e in envs.values() for k in e})
dev = torch.device("cuda:0")
m = build_model(load_cfg(a.config))
m.unet.init_seeded(0)
m = m.to(dev).eval()
R = m.spec.img_resolution
x = torch.randn(a.batch, 3, R, R, device=dev)
lab = torch.eye(2, device=dev)[:1].expand(a.batch, -1)
t = torch.tensor([500.0], device=dev)
times = {v: [] for v in a.variants}
for r in range(a.rounds + 1):
    for v in a.variants:
        ops.lib = libs[v]
        for k in all_vars:
            os.environ.pop(k, None)
        os.environ.update(envs[v])
        m.unet.invalidate()                     # packed weights are library-independent, but keep variants isolated
        m(x, t, class_labels=lab)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(a.evals):
            m(x, t, class_labels=lab)
        e1.record()
        torch.cuda.synchronize()
        if r > 0:                               # round 0 warms every variant up
            times[v].append(e0.elapsed_time(e1) / a.evals)
for v in a.variants:
    ts = times[v]
    print(f"{v:28s} ms/eval median {statistics.median(ts):7.3f}  min {min(ts):7.3f}  max {max(ts):7.3f}  ({len(ts)} rounds x {a.evals})")
