#!/usr/bin/env python3
"""Per-role segment times of the role-split halo conv (needs the MAD_H2_STAMPS build installed as
mad_amd/libmad_hip.so; GPU only)."""
import sys, os, math, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mad_amd import ops, _lib
dev = torch.device("cuda:0")
B, H, Cin, Cout = 32, 256, 128, 128
fuse = (sys.argv[1] if len(sys.argv) > 1 else "1") == "1"
os.environ["MAD_HALO_V2"] = "1"
x = (torch.randn(B, H, H, Cin, device=dev) * 0.7).to(torch.bfloat16)
w = (torch.randn(Cout, 3, 3, Cin, device=dev) / math.sqrt(9 * Cin)).to(torch.bfloat16)
bias = torch.randn(Cout, device=dev)
coef = torch.stack([1 + 0.1 * torch.randn(B, Cin, device=dev), 0.1 * torch.randn(B, Cin, device=dev)], -1).contiguous()
for _ in range(3):
    ops.conv3x3_fused(x, w, bias, coef=coef if fuse else None, want_stats=fuse)
torch.cuda.synchronize()
buf = (ctypes.c_uint64 * 12)()
assert _lib.lib.mad_debug_read_h2_stamps(buf) == 0
taps = 32 * 4 * 9          # tiles per workgroup x chunks x taps
names = {0: ["barrier", "weight copies", "reads+mfma", "wait copies"], 1: ["barrier", "transform (avg over 9)", "halo copies", "wait copies"],
         2: ["barrier", "store", "transform (avg over 9)", "-"]}
for role, rn in enumerate(["MFMA", "LOAD", "STORE"]):
    v = [buf[role * 4 + i] / taps for i in range(4)]
    print(f"{rn:6s} per tap: " + "  ".join(f"{n} {x_:7.0f}" for n, x_ in zip(names[role], v)) + f"   sum {sum(v):7.0f}")
