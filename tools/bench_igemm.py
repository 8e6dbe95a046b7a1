#!/usr/bin/env python3
"""Tile sweep of the implicit-GEMM conv on the denoiser's 1x1 and small-image shapes (GPU only)."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mad_amd import ops
from bench_conv import timeit

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
# (H, Cin, Cout, ksize)
shapes = [(8, 512, 512, 3), (8, 1024, 512, 3), (8, 512, 512, 1), (8, 512, 1536, 1), (16, 512, 512, 1),
          (16, 512, 1536, 1), (32, 256, 768, 1), (32, 256, 256, 1), (64, 512, 256, 1), (128, 256, 128, 1),
          (256, 256, 128, 1), (16, 512, 512, 3)]
if B == 64:      # the 64x64 model's shapes
    shapes = [(8, 768, 768, 3), (8, 1536, 768, 3), (8, 576, 576, 3), (8, 768, 768, 1), (8, 768, 2304, 1), (16, 576, 576, 1),
              (16, 576, 1728, 1), (32, 384, 1152, 1), (32, 384, 384, 1)]
tiles = ["128x128x2", "128x128x3", "128x64x2", "64x128x3", "64x64x4"]
for H, Cin, Cout, k in shapes:
    x = (torch.randn(B, H, H, Cin, device=dev) * 0.7).to(torch.bfloat16)
    w = (torch.randn(Cout, k, k, Cin, device=dev) / math.sqrt(k * k * Cin)).to(torch.bfloat16)
    bias = torch.randn(Cout, device=dev)
    flops = 2.0 * B * H * H * Cout * Cin * k * k
    line = f"H={H:4d} Cin={Cin:5d} Cout={Cout:5d} k={k}"
    ref = None
    for t in tiles:
        os.environ["MAD_IGEMM_TILE"] = t
        ms = timeit(lambda: ops.conv2d(x, w, bias))
        y = ops.conv2d(x, w, bias).float()
        if ref is None:
            ref = y
        err = (y - ref).abs().max().item()
        line += f" | {t} {ms * 1e3:7.1f}us {flops / ms / 1e9:6.0f}TF" + (f" ERR{err:.2g}" if err > 1e-2 else "")
    os.environ.pop("MAD_IGEMM_TILE")
    for ks in (2, 4, 8):
        os.environ["MAD_IGEMM_KSPLIT"] = str(ks)
        ms = timeit(lambda: ops.conv2d(x, w, bias))
        err = (ops.conv2d(x, w, bias).float() - ref).abs().max().item()
        line += f" | ks{ks} {ms * 1e3:7.1f}us" + (f" ERR{err:.2g}" if err > 1e-2 else "")
    os.environ.pop("MAD_IGEMM_KSPLIT", None)
    ms = timeit(lambda: ops.conv2d(x, w, bias))
    line += f" | auto {ms * 1e3:7.1f}us"
    print(line, flush=True)
