#!/usr/bin/env python3
"""The clock the chip holds under the halo conv, measured INSIDE the kernel (GPU only; needs the MAD_HALO_STAMPS build:
tools/build_variant.sh stamps conv_halo.hip -DMAD_HALO_STAMPS, run with MAD_LIB_PATH=mad_amd/libmad_hip_stamps.so).

The dominant layer shape (256x256, 128 -> 128 channels, batch 32) is launched back to back for a few seconds; every
workgroup records shader cycles (s_memtime) and 100 MHz reference ticks (s_memrealtime) between its first and last
instruction.  clock = cycles / ticks * 100 MHz, median over the workgroups of the last launches.  Cases: loader
transform + statistics on random data, plain on random data, both on all-zero operands.  One JSON line per case."""
import ctypes, json, math, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mad_amd import _lib, ops

dev = torch.device("cuda:0")
B, H, Cin, Cout = 32, 256, 128, 128
lib = _lib.lib
lib.mad_debug_read_clock.restype = ctypes.c_int
lib.mad_debug_read_clock.argtypes = [ctypes.c_void_p, ctypes.c_int]
N = 8192


def case(name, zero, fuse, seconds=3.0):
    x = (torch.randn(B, H, H, Cin, device=dev) * 0.7).to(torch.bfloat16)
    w = (torch.randn(Cout, 3, 3, Cin, device=dev) / math.sqrt(9 * Cin)).to(torch.bfloat16)
    if zero:
        x.zero_(); w.zero_()
    bias = torch.randn(Cout, device=dev)
    coef = torch.stack([1 + 0.1 * torch.randn(B, Cin, device=dev), 0.1 * torch.randn(B, Cin, device=dev)], -1).contiguous()
    fn = (lambda: ops.conv3x3_fused(x, w, bias, coef=coef, want_stats=True)) if fuse else (lambda: ops.conv3x3_fused(x, w, bias, coef=None))
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        n += 50
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        fn()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    buf = (ctypes.c_uint64 * (2 * N))()
    assert lib.mad_debug_read_clock(buf, N) == 0
    mhz = [buf[2 * i] / buf[2 * i + 1] * 100.0 for i in range(N) if buf[2 * i + 1] > 0]
    cyc = [buf[2 * i] for i in range(N) if buf[2 * i + 1] > 0]
    tf = 2.0 * B * H * H * Cout * Cin * 9 / ms / 1e9
    print(json.dumps({"case": name, "tflops": round(tf, 1), "ms": round(ms, 4), "clock_mhz_median": round(statistics.median(mhz), 1),
                      "clock_mhz_p10": round(sorted(mhz)[len(mhz) // 10], 1), "clock_mhz_p90": round(sorted(mhz)[9 * len(mhz) // 10], 1),
                      "cycles_per_workgroup_median": int(statistics.median(cyc)), "workgroups": len(mhz),
                      "per_clock_frac_of_mfma_peak": round(tf / (2500.0 * statistics.median(mhz) / 2400.0), 3)}), flush=True)


case("loader transform + statistics, random data", False, True)
case("plain, random data", False, False)
case("loader transform + statistics, all-zero operands", True, True)
case("plain, all-zero operands", True, False)
