#!/usr/bin/env python3
"""One conv shape, a few launches: the target of rocprofv3 --pmc passes (GPU only).
usage: pmc_conv.py [halo|halo_plain|igemm] [H] [Cin] [Cout] [B]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mad_amd import ops

kind = sys.argv[1] if len(sys.argv) > 1 else "halo"
H, Cin, Cout, B = (int(v) for v in (sys.argv[2:6] + ["256", "128", "128", "32"][len(sys.argv) - 2:]))
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
x = (torch.randn(B, H, H, Cin, device=dev, generator=g) * 0.7).to(torch.bfloat16)
w = (torch.randn(Cout, 3, 3, Cin, device=dev, generator=g) / math.sqrt(9 * Cin)).to(torch.bfloat16)
bias = torch.randn(Cout, device=dev, generator=g)
coef = torch.stack([1 + 0.1 * torch.randn(B, Cin, device=dev, generator=g),
                    0.1 * torch.randn(B, Cin, device=dev, generator=g)], -1).contiguous()
res = (torch.randn(B, H, H, Cout, device=dev, generator=g)).to(torch.bfloat16)
for _ in range(5):
    if kind == "halo":
        ops.conv3x3_fused(x, w, bias, coef=coef, residual=res, res_mode=1, want_stats=True)
    elif kind == "halo_plain":
        ops.conv3x3_fused(x, w, bias, coef=None)
    else:
        ops.conv2d(x, w, bias)
torch.cuda.synchronize()
print("done", kind, H, Cin, Cout, B)
