#!/usr/bin/env python3
"""Shader clock and socket power the chip holds under the halo conv (GPU only): the dominant layer shape is
launched back to back for a few seconds on (a) real data with the loader transform, (b) real data without it,
(c) all-zero operands, while `rocm-smi --showclocks --showpower --json` is sampled from the host.  One JSON line
per case on stdout: TFLOP/s, the median / min / max sclk (MHz) and power (W) over the samples."""
import json, math, os, re, statistics, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mad_amd import ops


RAW = os.environ.get("PROBE_RAW") == "1"


def smi_sample():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True,
                             timeout=20).stdout
        d = json.loads(out)
    except Exception as e:                       # noqa: BLE001 -- the probe reports what it could not read
        return {"error": repr(e)[:200]}
    card = d.get("card0") or next(iter(d.values()))
    s = {}
    for k, v in card.items():
        kl = k.lower()
        if "sclk" in kl:                          # "(1915Mhz)" next to a level index: take the MHz figure
            m = re.search(r"([0-9.]+)\s*mhz", str(v), re.I)
            if m:
                s["sclk_mhz"] = float(m.group(1))
        elif "power" in kl and "(w)" in kl:
            m = re.search(r"([0-9.]+)", str(v))
            if m:
                s["power_w"] = float(m.group(1))
    if RAW:
        s["raw"] = card
    return s


def run_case(name, fn, flops, seconds=6.0):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        fn()
    b.record()
    torch.cuda.synchronize()
    per = a.elapsed_time(b) / 20
    n = int(seconds * 1e3 / per)
    samples = []

    def work():                                  # launches from a second thread: the GPU stays busy while this one samples
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()

    th = threading.Thread(target=work)
    th.start()
    time.sleep(0.7)
    t_end = time.time() + seconds * 0.7
    while time.time() < t_end and th.is_alive():
        samples.append(smi_sample())
    th.join()
    ms = a.elapsed_time(b) / n
    clk = [s["sclk_mhz"] for s in samples if "sclk_mhz" in s]
    pw = [s["power_w"] for s in samples if "power_w" in s]
    rec = {"case": name, "ms": round(ms, 4), "tflops": round(flops / ms / 1e9, 1), "samples": len(samples)}
    if clk:
        rec["sclk_mhz"] = {"median": statistics.median(clk), "min": min(clk), "max": max(clk)}
    if pw:
        rec["power_w"] = {"median": statistics.median(pw), "min": min(pw), "max": max(pw)}
    if not clk and samples:
        rec["smi"] = samples[0]
    print(json.dumps(rec), flush=True)


def main():
    dev = torch.device("cuda:0")
    B, H, Cin, Cout = 32, 256, 128, 128
    x = (torch.randn(B, H, H, Cin, device=dev) * 0.7).to(torch.bfloat16)
    w = (torch.randn(Cout, 3, 3, Cin, device=dev) / math.sqrt(9 * Cin)).to(torch.bfloat16)
    bias = torch.randn(Cout, device=dev)
    coef = torch.stack([1 + 0.1 * torch.randn(B, Cin, device=dev), 0.1 * torch.randn(B, Cin, device=dev)], -1).contiguous()
    flops = 2.0 * B * H * H * Cout * Cin * 9
    global RAW
    RAW = True
    print(json.dumps({"case": "idle", **smi_sample()}), flush=True)      # with the tool's own fields, once
    RAW = os.environ.get("PROBE_RAW") == "1"
    run_case("halo+gn+stats, real data", lambda: ops.conv3x3_fused(x, w, bias, coef=coef, want_stats=True), flops)
    run_case("halo plain, real data", lambda: ops.conv3x3_fused(x, w, bias, coef=None), flops)
    xz, wz = torch.zeros_like(x), torch.zeros_like(w)
    run_case("halo+gn+stats, zero operands", lambda: ops.conv3x3_fused(xz, wz, bias, coef=coef, want_stats=True), flops)
    run_case("halo plain, zero operands", lambda: ops.conv3x3_fused(xz, wz, bias, coef=None), flops)
    # a plain bf16 GEMM of the library the image ships, as the yardstick of what the chip sustains on real data
    a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
    bm = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
    run_case("torch.matmul 8192^3 bf16 (hipBLASLt), real data", lambda: torch.matmul(a, bm), 2.0 * 8192 ** 3)


if __name__ == "__main__":
    main()
