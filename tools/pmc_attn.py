#!/usr/bin/env python3
"""A few launches of the attention kernel on one shape, for rocprofv3 counter passes.  usage: pmc_attn.py B L heads"""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mad_amd import ops
B, L, heads = (int(v) for v in sys.argv[1:4])
H = math.isqrt(L)
qkv = (torch.randn(B, H, H, heads * 3 * 64, device="cuda:0") * 1.5).to(torch.bfloat16)
for _ in range(6):
    ops.attention(qkv, heads)
torch.cuda.synchronize()
