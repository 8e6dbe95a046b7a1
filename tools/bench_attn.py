#!/usr/bin/env python3
"""Self-attention kernel on the denoiser's shapes: time, TFLOP/s and error against an fp32 softmax (GPU only).
MAD_ATTN_OLD=1 in the environment selects the generic kernel for bf16."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mad_amd import ops
from bench_conv import timeit

dev = torch.device("cuda:0")
shapes = [(32, 1024, 4), (32, 256, 8), (32, 64, 8), (64, 1024, 6), (64, 256, 9), (64, 64, 12)]
for B, L, heads in shapes:
    g = torch.Generator(device="cpu").manual_seed(L + heads)
    H = int(math.isqrt(L))
    qkv = (torch.randn(B, H, H, heads * 3 * 64, generator=g) * 1.5).to(torch.bfloat16).to(dev)
    ms = timeit(lambda: ops.attention(qkv, heads))
    y = ops.attention(qkv, heads).float()
    x = qkv.float().reshape(B, L, heads, 3, 64)
    q, k, v = x[:, :, :, 0].transpose(1, 2), x[:, :, :, 1].transpose(1, 2), x[:, :, :, 2].transpose(1, 2)
    p = torch.softmax(q @ k.transpose(-1, -2) * 0.125, dim=-1)
    ref = (p @ v).transpose(1, 2).reshape(B, H, H, heads * 64)
    err = (y - ref).abs().max().item() / ref.abs().max().item()
    flops = 4.0 * B * heads * L * L * 64
    print(f"B={B:3d} L={L:5d} heads={heads:2d}  {ms * 1e3:7.1f} us  {flops / ms / 1e9:6.0f} TF  rel err {err:.2e}", flush=True)
