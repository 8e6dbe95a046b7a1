#!/usr/bin/env python3
"""Per-segment cycle breakdown of one workgroup of the halo conv (needs the MAD_HALO_STAMPS build
installed as mad_amd/libmad_hip.so; GPU only)."""
import sys, os, math, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mad_amd import ops, _lib

dev = torch.device("cuda:0")
B, H, Cin, Cout = int(sys.argv[3]) if len(sys.argv) > 3 else 32, int(sys.argv[1]) if len(sys.argv) > 1 else 256, int(sys.argv[4]) if len(sys.argv) > 4 else 128, 128
fuse = (sys.argv[2] if len(sys.argv) > 2 else "1") == "1"
x = (torch.randn(B, H, H, Cin, device=dev) * 0.7).to(torch.bfloat16)
w = (torch.randn(Cout, 3, 3, Cin, device=dev) / math.sqrt(9 * Cin)).to(torch.bfloat16)
bias = torch.randn(Cout, device=dev)
coef = torch.stack([1 + 0.1 * torch.randn(B, Cin, device=dev), 0.1 * torch.randn(B, Cin, device=dev)], -1).contiguous()
for _ in range(3):
    ops.conv3x3_fused(x, w, bias, coef=coef if fuse else None, want_stats=fuse)
torch.cuda.synchronize()
lib = _lib.lib
buf = (ctypes.c_uint32 * (8 * 9 * 8))()
assert lib.mad_debug_read_stamps(buf) == 0
full = np.array(buf, dtype=np.int64).reshape(8, 9, 8)
st = full[:4, :, :5]
names = ["wait copies", "barrier", "copies+mfma issue", "transform", "-> next tap top"]
for wv in range(4):
    print(f"wave {wv}:")
    for tap in range(9):
        d = [(st[wv, tap, i + 1] - st[wv, tap, i]) & 0xffffffff for i in range(4)]
        nxt = ((st[wv, tap + 1, 0] - st[wv, tap, 0]) & 0xffffffff) if tap < 8 else -1
        print(f"  tap {tap}: " + "  ".join(f"{n} {v:5d}" for n, v in zip(names, d)) + f"   tap total {nxt}")
tot = (st[:, 8, 0] - st[:, 0, 0]) & 0xffffffff
print("8 taps, cycles per wave:", tot, " ideal MFMA issue 8*512 = 4096")
ph = np.concatenate([full[:4, :9, 5], full[:4, :2, 6]], axis=1)      # 11 phase stamps per wave
names = ["prologue", "main loop", "(wait residual+sync)", "h0 acc->LDS+sync", "h0 read/store", "(sync)", "h1 acc->LDS+sync",
         "h1 read/store", "stats", "store drain"]
for wv in range(4):
    d = [(ph[wv, i + 1] - ph[wv, i]) & 0xffffffff for i in range(10)]
    print(f"wave {wv}: " + "  ".join(f"{n} {v}" for n, v in zip(names, d)))
pro = full[:4, 2:6, 6]      # prologue stamps 11..14: geometry done, copies issued, halo landed, transform done
t0 = full[:4, 0, 5]
t1 = full[:4, 1, 5]
for wv in range(4):
    d = [(pro[wv, 0] - t0[wv]) & 0xffffffff] + [(pro[wv, i + 1] - pro[wv, i]) & 0xffffffff for i in range(3)] + [(t1[wv] - pro[wv, 3]) & 0xffffffff]
    print(f"wave {wv} prologue: geometry {d[0]}  issue copies {d[1]}  wait halo {d[2]}  transform {d[3]}  wait weights {d[4]}")
