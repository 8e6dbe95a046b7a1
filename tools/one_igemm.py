#!/usr/bin/env python3
"""One implicit-GEMM conv shape, repeated (for rocprofv3 / tile x split sweeps).  usage: one_igemm.py B H Cin Cout k [sweep]"""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mad_amd import ops
from bench_conv import timeit

B, H, Cin, Cout, k = (int(v) for v in sys.argv[1:6])
dev = torch.device("cuda:0")
x = (torch.randn(B, H, H, Cin, device=dev) * 0.7).to(torch.bfloat16)
w = (torch.randn(Cout, k, k, Cin, device=dev) / math.sqrt(k * k * Cin)).to(torch.bfloat16)
bias = torch.randn(Cout, device=dev)
res = torch.randn(B, H, H, Cout, device=dev).to(torch.bfloat16)
flops = 2.0 * B * H * H * Cout * Cin * k * k
if len(sys.argv) > 6:
    for t in ["128x128x2", "128x64x2", "64x128x3", "64x64x4"]:
        os.environ["MAD_IGEMM_TILE"] = t
        line = f"{t:10s}"
        for ks in (1, 2, 3, 4, 6, 8):
            os.environ["MAD_IGEMM_KSPLIT"] = str(ks)
            ms = timeit(lambda: ops.conv2d(x, w, bias, residual=res, res_mode=1, want_stats=True))
            line += f" | ks{ks} {ms * 1e3:7.1f}us {flops / ms / 1e9:5.0f}TF"
        print(line, flush=True)
else:
    for _ in range(20):
        ops.conv2d(x, w, bias, residual=res, res_mode=1, want_stats=True)
    torch.cuda.synchronize()
