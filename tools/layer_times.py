#!/usr/bin/env python3
"""Per-launch timing of every MFMA kernel launch in one denoiser evaluation, grouped by shape (GPU only)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mad_amd import ops
from mad_amd.config import load_cfg
from mad_amd.model import build_model

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg_file = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "configs/model_256_256.yaml")
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device("cuda:0")
cfg = load_cfg(cfg_file)
m = build_model(cfg)
m.unet.init_seeded(0)
m = m.to(dev).eval()
R = m.spec.img_resolution
x = torch.randn(B, 3, R, R, device=dev)
lab = torch.eye(2, device=dev)[torch.zeros(B, dtype=torch.long)]
t = torch.tensor([500.0], device=dev)
records = []
orig = ops.TIMER.bracket

def bracket(name, flops):
    import inspect
    f = inspect.currentframe().f_back
    loc = f.f_locals
    shape = []
    for k in ("B", "H", "W_", "C1", "C2", "Cout", "k", "C", "heads", "Cin", "upsample", "res_mode"):
        if k in loc:
            shape.append(f"{k}={loc[k]}")
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    records.append((name + " " + " ".join(shape), flops, a, b))
    return b

ops.TIMER.bracket = bracket
ops.TIMER.active = True
for it in range(3):
    records.clear()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    m(x, t, class_labels=lab)
    e1.record()
    torch.cuda.synchronize()
agg = collections.OrderedDict()
for name, flops, a, b in records:
    d = agg.setdefault(name, [0, 0.0, 0.0])
    d[0] += 1; d[1] += flops; d[2] += a.elapsed_time(b)
tot = sum(d[2] for d in agg.values())
print(f"eval wall {e0.elapsed_time(e1):.2f} ms, MFMA kernels (event-bracketed) {tot:.2f} ms")
for name, (n, fl, ms) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
    print(f"{ms:7.3f} ms  n={n:2d}  {fl / ms / 1e9:7.1f} TF  {name}")
