#!/usr/bin/env python3
"""Cost of the 1x1 projection folded into the halo conv (decoder conv1): with and without the extra K-tiles (GPU only)."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mad_amd import ops
from bench_conv import timeit

dev = torch.device("cuda:0")
B = 32
for H, Cin, Cout, SC in [(256, 128, 128, 256), (128, 128, 128, 256), (128, 128, 128, 384), (64, 256, 256, 512), (64, 256, 256, 384)]:
    x = (torch.randn(B, H, H, Cin, device=dev) * 0.7).to(torch.bfloat16)
    s1 = (torch.randn(B, H, H, SC // 2, device=dev) * 0.7).to(torch.bfloat16)
    s2 = (torch.randn(B, H, H, SC - SC // 2, device=dev) * 0.7).to(torch.bfloat16)
    w = (torch.randn(Cout, 3, 3, Cin, device=dev) / math.sqrt(9 * Cin)).to(torch.bfloat16)
    sw = (torch.randn(Cout, SC, device=dev) / math.sqrt(SC)).to(torch.bfloat16)
    bias = torch.randn(Cout, device=dev)
    coef = torch.stack([1 + 0.1 * torch.randn(B, Cin, device=dev), 0.1 * torch.randn(B, Cin, device=dev)], -1).contiguous()
    t0 = timeit(lambda: ops.conv3x3_fused(x, w, bias, coef=coef, want_stats=True))
    t1 = timeit(lambda: ops.conv3x3_fused(x, w, bias, coef=coef, want_stats=True, skip=(s1, s2, sw)))
    f0 = 2.0 * B * H * H * Cout * Cin * 9
    f1 = f0 + 2.0 * B * H * H * Cout * SC
    print(f"H={H} Cin={Cin} Cout={Cout} SC={SC}: plain {t0:.3f} ms {f0 / t0 / 1e9:.0f} TF | with projection {t1:.3f} ms {f1 / t1 / 1e9:.0f} TF"
          f" | marginal {(t1 - t0):.3f} ms = {(f1 - f0) / (t1 - t0) / 1e9:.0f} TF", flush=True)
