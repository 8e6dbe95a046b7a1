#!/usr/bin/env python3
"""Micro-benchmark of the conv kernels on the denoiser's dominant shapes (GPU only)."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mad_amd import ops


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    dev = torch.device("cuda:0")
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    shapes = [(256, 128, 128), (256, 256, 128), (128, 128, 128), (128, 256, 256), (128, 384, 128), (64, 256, 256),
              (64, 512, 256), (32, 512, 512), (32, 256, 256), (16, 1024, 512), (16, 512, 512)]
    os.environ["MAD_HALO_BN"] = "128"
    for H, Cin, Cout in shapes:
        x = (torch.randn(B, H, H, Cin, device=dev) * 0.7).to(torch.bfloat16)
        w = (torch.randn(Cout, 3, 3, Cin, device=dev) / math.sqrt(9 * Cin)).to(torch.bfloat16)
        bias = torch.randn(Cout, device=dev)
        if os.environ.get("BENCH_ZERO"):          # all-zero operands: the clock the chip holds without data toggling
            x.zero_(); w.zero_()
        coef = torch.stack([1 + 0.1 * torch.randn(B, Cin, device=dev), 0.1 * torch.randn(B, Cin, device=dev)], -1).contiguous()
        flops = 2.0 * B * H * H * Cout * Cin * 9
        t_ig = timeit(lambda: ops.conv2d(x, w, bias))
        line = f"H={H:4d} Cin={Cin:5d} Cout={Cout:4d}  igemm {t_ig:8.3f} ms {flops / t_ig / 1e9:8.1f} TF"
        if ops.fused_conv_ok(H, H, Cout):
            t_h0 = timeit(lambda: ops.conv3x3_fused(x, w, bias, coef=None))
            t_h1 = timeit(lambda: ops.conv3x3_fused(x, w, bias, coef=coef, want_stats=True))
            line += f" | halo {t_h0:8.3f} ms {flops / t_h0 / 1e9:8.1f} TF | halo+gn+stats {t_h1:8.3f} ms {flops / t_h1 / 1e9:8.1f} TF"
            os.environ["MAD_HALO_BN"] = "64"
            t_n0 = timeit(lambda: ops.conv3x3_fused(x, w, bias, coef=None))
            t_n1 = timeit(lambda: ops.conv3x3_fused(x, w, bias, coef=coef, want_stats=True))
            os.environ["MAD_HALO_BN"] = "128"
            line += f" | narrow {flops / t_n0 / 1e9:7.1f} TF, fused {flops / t_n1 / 1e9:7.1f} TF"
        print(line, flush=True)


if __name__ == "__main__":
    main()
