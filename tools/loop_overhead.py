#!/usr/bin/env python3
"""What one denoiser evaluation costs inside the translation loop against the bare denoiser (GPU only): the same
process alternates (a) K bare evaluations and (b) a removal loop of S steps (2S - 1 evaluations + sampler steps)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mad_amd.config import load_cfg
from mad_amd.model import build_model
from mad_amd.translation import TranslationDiffusion
from mad_amd.synthetic import synthetic_faces

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = load_cfg(os.path.join(ROOT, "configs/model_256_256.yaml"))
dev = torch.device("cuda:0")
B, S = 32, int(sys.argv[1]) if len(sys.argv) > 1 else 40
m = build_model(cfg); m.unet.init_seeded(0); m = m.to(dev).eval()
td = TranslationDiffusion(cfg, dev)
src = synthetic_faces(B, 256, seed=3).to(dev)
x = torch.randn(B, 3, 256, 256, device=dev)
lab = torch.eye(2, device=dev)[:1].expand(B, -1)
t = torch.tensor([500.0], device=dev)

def bare(k):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k):
        m(x, t, class_labels=lab)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3

def loop():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    td.domain_translation(m, m, src, 0, 1, start_from_step=S)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / (2 * S - 1) * 1e3

bare(3); loop()
for r in range(3):
    print(f"bare {bare(2 * S - 1):7.3f} ms/eval   loop {loop():7.3f} ms/eval", flush=True)
