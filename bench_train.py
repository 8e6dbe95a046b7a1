#!/usr/bin/env python3
"""Training-step benchmark: one optimizer step of the reference's main.py loop (main.py:223-257) per
"step" -- GRADIENT_ACCUMULATION_STEPS micro-batches of forward (train mode, dropout) + backward, the
data-parallel gradient all-reduce over RCCL, nan_to_num + AdamW + EMA -- on synthetic images.

    python bench_train.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench_train.py --gpus N --steps K --warmup W

Same contract as bench.py: W untimed steps, K timed steps between barrier + synchronize, max over ranks,
rank 0 prints ONE JSON line.  `value` = images trained per second over all ranks.
"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
PEAK_BF16_TFLOPS = 2500.0


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=2)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--config", default=os.path.join(REPO, "configs", "model_256_256.yaml"))
    p.add_argument("--batch", type=int, default=0, help="images per GPU per micro-batch (0: TRAIN.BATCH_SIZE)")
    p.add_argument("--accum", type=int, default=0, help="micro-batches per optimizer step (0: the config's)")
    p.add_argument("--split", default="fused", choices=["fused", "micro"],
                   help="fused: the accum*batch images of one optimizer step run as ONE batch (same mean gradient; "
                        "288 GB of HBM hold its activations); micro: accum separate micro-batches as the reference runs them")
    p.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    p.add_argument("--no-cpu-baseline", action="store_true")
    return p.parse_args()


def cpu_baseline(spec, B):
    """The oracle's forward + autograd backward + torch AdamW on this host's cores for ONE micro-batch of B=1."""
    import torch.nn.functional as F
    from mad_amd.init import seeded_state_dict
    from mad_amd.synthetic import synthetic_faces
    from oracle.unet_ref import unet_forward
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(cores, int(os.environ.get("OMP_NUM_THREADS", cores)))
    torch.set_num_threads(cores)
    sd = seeded_state_dict(spec, 0)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if "resample_filter" not in k}
    full = dict(sd); full.update(params)
    opt = torch.optim.AdamW(list(params.values()), lr=1e-4, betas=(0.95, 0.999), eps=1e-7)
    kw = dict(img_resolution=spec.img_resolution, model_channels=spec.model_channels,
              channel_mult=spec.channel_mult, num_blocks=spec.num_blocks, label_dim=spec.label_dim)
    x = synthetic_faces(1, spec.img_resolution, seed=0)
    noise = torch.randn_like(x)
    lab = torch.eye(2)[:1]
    t0 = time.perf_counter()
    with torch.enable_grad():
        pred = unet_forward(full, x + noise, torch.tensor([500.0]), lab, **kw)
        F.mse_loss(pred, noise).backward()
    opt.step()
    dt = time.perf_counter() - t0
    return {"value": 1.0 / dt, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"one micro-batch of 1 image: forward + autograd backward + AdamW in {dt:.1f} s (fp32, no dropout)"}


def hbm_kernels(dev, B, R, C, nparams, iters=10):
    """Achieved GB/s of the bandwidth-bound training kernels on the workload's largest activation shape
    (outside the timed region): GroupNorm backward partial sums (4 B/element: x, dy) and apply (6 B/element:
    x, dy in, dx out), forward apply with dropout (4 B/element), the fused AdamW+EMA update (36 B/parameter)."""
    from mad_amd import ops
    from mad_amd.unet_spec import gn_groups

    def timed(fn):
        for _ in range(2):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / iters * 1e-3

    out = {}
    x = torch.randn(B, R, R, C, device=dev).to(torch.bfloat16)
    dy = torch.randn(B, R, R, C, device=dev).to(torch.bfloat16)
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    st = ops.gn_partial(x)
    coef = ops.gn_finalize(st, R * R, gamma, beta, gn_groups(C), 1e-5)
    nsg = ops.gn_nsplit(R * R)
    gsum = torch.empty((B, nsg, C, 2), dtype=torch.float32, device=dev)
    bcoef = torch.rand((B, C, 4), device=dev) * 0.1
    dx = torch.empty_like(x)
    code = ops._dtype_code(x)
    t = timed(lambda: ops.check(ops.lib.mad_gn_bwd_partial(x.data_ptr(), C, None, 0, dy.data_ptr(), 0, coef.data_ptr(), 1, 0, 0.0,
                                                           code, gsum.data_ptr(), B, R, R, nsg, ops._stream()), "mad_gn_bwd_partial"))
    out["gn_bwd_partial"] = {"GB/s": 4.0 * x.numel() / t / 1e9, "bytes_per_element": 4, "us": t * 1e6}
    t = timed(lambda: ops.check(ops.lib.mad_gn_bwd_apply(x.data_ptr(), C, None, 0, dy.data_ptr(), 0, coef.data_ptr(), 1, 0, 0.0,
                                                         bcoef.data_ptr(), None, 0, None, 0, dx.data_ptr(), None, code, B, R, R,
                                                         ops._stream()), "mad_gn_bwd_apply"))
    out["gn_bwd_apply"] = {"GB/s": 6.0 * x.numel() / t / 1e9, "bytes_per_element": 6, "us": t * 1e6}
    t = timed(lambda: ops.gn_apply(x, coef, act=True, drop_seed=7, drop_p=0.1))
    out["gn_apply_dropout"] = {"GB/s": 4.0 * x.numel() / t / 1e9, "bytes_per_element": 4, "us": t * 1e6}
    del x, dy, dx
    p, g, m, v, e = (torch.rand(nparams, device=dev) for _ in range(5))
    t = timed(lambda: ops.adamw_step_(p, g, m, v, e, 1e-4, 0.95, 0.999, 1e-7, 1e-2, 3, ema_decay=0.9))
    out["adamw_ema"] = {"GB/s": 36.0 * nparams / t / 1e9, "bytes_per_parameter": 36, "us": t * 1e6}
    return out


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    from mad_amd.launch import needs_relaunch, relaunch_per_gpu
    if needs_relaunch(args.gpus):
        # a plain `python bench_train.py --gpus N`: one child launcher, N ranks; this process never touches the GPU
        raise SystemExit(relaunch_per_gpu(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match the {world} ranks this process was started with")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    from mad_amd import ops
    from mad_amd.config import load_cfg
    from mad_amd.model import build_model
    from mad_amd.synthetic import synthetic_faces
    from mad_amd.train import UNetTrainer

    cfg = load_cfg(args.config)
    B = args.batch or int(cfg.TRAIN.BATCH_SIZE)
    accum = args.accum or int(cfg.TRAIN.GRADIENT_ACCUMULATION_STEPS)
    model = build_model(cfg, precision=args.precision)
    model.unet.init_seeded(0)
    model = model.to(dev).train()
    spec = model.spec
    R = spec.img_resolution
    tr = UNetTrainer(model.unet, lr=float(cfg.TRAIN.LR), dropout=0.10, seed=rank)
    betas = torch.linspace(float(cfg.TRAIN.NOISE_SCHEDULER.BETA_START), float(cfg.TRAIN.NOISE_SCHEDULER.BETA_END),
                           int(cfg.TRAIN.TIME_STEPS), dtype=torch.float32)
    alphas = torch.cumprod(1.0 - betas, 0).to(dev)
    gen = torch.Generator(device=dev).manual_seed(77 + rank)
    x0 = synthetic_faces(B * accum, R, seed=300 + rank).to(dev)
    labels = torch.eye(2, device=dev)[torch.arange(B * accum, device=dev) % 2]

    passes = accum if args.split == "micro" else 1        # forward/backward passes per optimizer step

    def one_step():
        return tr.train_step(x0, labels, alphas, generator=gen, num_timesteps=int(cfg.TRAIN.TIME_STEPS),
                             accumulation=passes, warmup_steps=int(cfg.TRAIN.LR_WARMUP))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = one_step()
    barrier()
    elapsed = time.perf_counter() - t0
    loss = float(loss)
    assert loss == loss and bool(torch.isfinite(tr.flat).all()), "training diverged"
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # one more (untimed) micro-batch with per-launch stream events on the MFMA kernels
    ops.TIMER.active = True
    Bp = B * accum // passes
    tr.forward_backward(x0[:Bp], torch.full((Bp,), 500.0, device=dev), labels[:Bp], torch.randn_like(x0[:Bp]))
    ops.TIMER.active = False
    kern = ops.TIMER.summary()

    if rank == 0:
        f_fwd = spec.flops_per_image()["total"]
        images = world * B * accum * args.steps
        value = images / elapsed
        mfma = {k: v for k, v in kern.items() if v["ms"] > 0}
        dom = max(mfma, key=lambda k: mfma[k]["ms"])
        d = mfma[dom]
        achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12
        line = {
            "metric": f"training images/sec @{R}x{R}, DDP step (forward + backward + gradient all-reduce + AdamW + EMA)",
            "value": value, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": f"{os.path.basename(args.config)} training step (main.py loop body): {B * accum} images "
                                   f"per GPU per optimizer step (the config's {accum} micro-batches of {B}), run as "
                                   + (f"{accum} forward/backward passes of {B}" if passes > 1 else
                                      f"one forward/backward pass of {B * accum} (same mean gradient)")
                                   + ", dropout 0.1, epsilon prediction, MSE, AdamW(0.95, 0.999) + EMA, random-init "
                                     "(seeded) weights",
                       "batch_per_gpu": B, "accumulation": accum, "passes_per_step": passes, "resolution": R,
                       "parallelism": f"dp{world}", "peak_hbm_gib": torch.cuda.max_memory_allocated(dev) / 2 ** 30,
                       "parameters": int(tr.flat.numel()), "gflop_per_image_forward": f_fwd / 1e9,
                       "final_loss": loss},
            "mfma_util_end_to_end": value * 3.0 * f_fwd / 1e12 / (PEAK_BF16_TFLOPS * world),
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_BF16_TFLOPS, "traffic": None, "kernel": dom,
                         "launches_timed": d["launches"], "avg_launch_ms": d["ms"] / d["launches"],
                         "mfma_kernels": {k: {"tflops": v["flops"] / (v["ms"] * 1e-3) / 1e12, "ms_per_pass": v["ms"],
                                              "launches": v["launches"]} for k, v in mfma.items()}},
        }
        line["hbm_kernels"] = hbm_kernels(dev, B * accum // passes, R, spec.model_channels, int(tr.flat.numel()))
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(spec, B)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
