#!/usr/bin/env python3
"""Eager launches against hipGraph replay of one denoiser evaluation (256x256 model, B = 32), alternating in one
process (GPU only).  Measured equal within 0.3 %: the stream never idles between launches, so replay stays off by default."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from mad_amd.config import load_cfg
from mad_amd.model import build_model
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
cfg = load_cfg(os.path.join(ROOT, "configs/model_256_256.yaml"))
dev = torch.device("cuda:0")
m = build_model(cfg); m.unet.init_seeded(0); m = m.to(dev).eval()
B = 32
x = torch.randn(B, 3, 256, 256, device=dev)
lab = torch.eye(2, device=dev)[:1].expand(B, -1)
t = torch.tensor([500.0], device=dev)
def run(k):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): m(x, t, class_labels=lab)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3
for r in range(4):
    m.graph_replay = False; run(2); a = run(20)
    m.graph_replay = True; run(2); b = run(20)
    print(f"eager {a:7.3f}  graph {b:7.3f} ms/eval", flush=True)
