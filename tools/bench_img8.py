#!/usr/bin/env python3
"""Back-to-back launches of the image-resident 8x8 conv and of the implicit GEMM on the same shapes (GPU only)."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mad_amd import ops

dev = torch.device("cuda:0")
dt = torch.bfloat16
N = 50
for B, C, Cout, fuse in [(32, 64, 512, True), (32, 512, 512, True), (32, 512, 512, False), (32, 1024, 512, True),
                         (64, 512, 512, True), (64, 768, 768, True), (128, 512, 512, True)]:
    x = torch.randn(B, 8, 8, C, device=dev).to(dt)
    w = (torch.randn(Cout, 3, 3, C, device=dev) / math.sqrt(9 * C)).to(dt)
    bias = torch.randn(Cout, device=dev)
    coef = torch.stack([1 + 0.1 * torch.randn(B, C, device=dev), 0.1 * torch.randn(B, C, device=dev)], -1).contiguous()
    fl = 2.0 * B * 64 * Cout * C * 9
    out = []
    for name, fn in (("img8", lambda: ops.conv3x3_img8(x, w, bias, coef=coef if fuse else None, want_stats=True)),
                     ("igemm", lambda: ops.conv2d(x, w, bias, want_stats=True))):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(N):
            fn()
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) * 1e3 / N
        out.append(f"{name} {us:7.1f} us {fl / us / 1e6:6.0f} TF")
    print(f"B={B} C={C} Cout={Cout} fuse={fuse}: " + " | ".join(out), flush=True)
