#!/usr/bin/env python3
"""1x1 layers of the two denoisers: implicit GEMM against the 256-pixel-tile GEMM kernel (GPU only).
   python tools/bench_conv1x1.py [B]     (B = 32: 256x256 model's shapes, B = 64: the 64x64 model's)"""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from mad_amd import ops, _lib
from bench_conv import timeit

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
# (H, Cin, Cout, residual + statistics)
shapes = [(32, 256, 768, False), (32, 256, 256, True), (16, 512, 1536, False), (16, 512, 512, True),
          (8, 512, 1536, False), (8, 512, 512, True), (8, 1024, 512, False)]
if B == 64:
    shapes = [(32, 384, 1152, False), (32, 384, 384, True), (16, 576, 1728, False), (16, 576, 576, True),
              (8, 768, 2304, False), (8, 768, 768, True), (8, 1536, 768, False)]
for H, Cin, Cout, proj in shapes:
    x = (torch.randn(B, H, H, Cin, device=dev) * 0.7).to(torch.bfloat16)
    w = (torch.randn(Cout, 1, 1, Cin, device=dev) / math.sqrt(Cin)).to(torch.bfloat16)
    bias = torch.randn(Cout, device=dev)
    r = torch.randn(B, H, H, Cout, device=dev).to(torch.bfloat16) if proj else None
    kw = dict(residual=r, res_mode=_lib.RES_SAME, want_stats=True) if proj else {}
    flops = 2.0 * B * H * H * Cout * Cin
    line = f"H={H:3d} Cin={Cin:5d} Cout={Cout:5d} {'proj' if proj else 'qkv '}"
    ref = None
    for name, v in (("igemm", "0"), ("auto", None), ("gemm", "2")):
        os.environ.pop("MAD_CONV1X1", None)
        if v is not None:
            os.environ["MAD_CONV1X1"] = v
        ms = timeit(lambda: ops.conv2d(x, w, bias, **kw))
        y = ops.conv2d(x, w, bias, **kw)
        y = (y[0] if proj else y).float()
        if ref is None:
            ref = y
        err = (y - ref).abs().max().item()
        line += f" | {name} {ms * 1e3:6.1f}us {flops / ms / 1e9:5.0f}TF" + (f" ERR{err:.2g}" if err > 5e-2 else "")
    os.environ.pop("MAD_CONV1X1", None)
    print(line, flush=True)
