#!/usr/bin/env python3
"""Headline benchmark: translated images/sec at 256x256 for the makeup-removal loop
(noise encoding + regeneration, DDPM, 1000-step table, --num-of-step steps each way) on the HIP path.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one full `domain_translation` of a batch of --batch synthetic faces per GPU
(2*S-1 denoiser evaluations plus the sampler updates): about half a minute at the headline size.  Rank 0
prints ONE JSON line.  Each rank processes its own batch (independent images, no data-path collective):
weak scaling.

Time budget.  A whole run is meant to finish in well under ten minutes, so --steps is a CAP: after the
warm-up the cost of one step is estimated from the warm-up's measured time per denoiser evaluation, and
the run times exactly min(K, what fits --time-budget) full steps (never fewer than one; all ranks agree on
the number).  The line reports `steps` = the steps timed and `steps_requested` = K.  Warm-up steps run the
same loop on a shortened schedule (--warmup-num-of-step evaluations each way: the same kernels, shapes and
allocations per evaluation), and the first one also allocates and frees the full-size noise slab so the
timed steps find it in the allocator's cache.

Started as a plain process with --gpus N > 1 and no rendezvous variables, it starts its N ranks itself
(torch.distributed.run as a child process) before anything touches the GPU.
"""
import argparse
import json
import os
import sys
import time

T_PROCESS_START = time.time()

import torch  # noqa: E402

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

PEAK_BF16_TFLOPS = 2500.0          # MI355X dense bf16 MFMA peak


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=1)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--config", default=os.path.join(REPO, "configs", "model_256_256.yaml"))
    p.add_argument("--batch", type=int, default=32, help="images per GPU per step")
    p.add_argument("--num-of-step", type=int, default=700,
                   help="start_from_step of the removal loop (reference default 700)")
    p.add_argument("--workload", default="removal", choices=["removal", "transfer", "train"],
                   help="removal: domain_translation (headline); transfer: image_translation with inpainting and "
                        "per-step component masks, S = 180 (a secondary configuration, not the headline)")
    p.add_argument("--graph", type=int, default=0, help="1: replay each denoiser evaluation as one captured HIP graph")
    p.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-evals", type=int, default=40, help="denoiser evaluations in the CPU sample")
    p.add_argument("--profile-every", type=int, default=100,
                   help="bracket the MFMA kernels of every n-th denoiser evaluation with stream events")
    p.add_argument("--time-budget", type=float, default=0.0,
                   help="seconds for the timed steps; 0: what is left of --total-budget after start-up and warm-up, "
                        "minus the time the lines after the timed region take")
    p.add_argument("--total-budget", type=float, default=480.0, help="seconds for the whole run (default budget rule)")
    p.add_argument("--warmup-num-of-step", type=int, default=6,
                   help="start_from_step of the warm-up translations (same kernels and shapes, shorter schedule)")
    return p.parse_args()


def cpu_baseline(cfg, spec, num_of_step, n_evals):
    """The oracle (fp32 PyTorch-CPU restatement of the same loop) on this host's cores, on a bounded
    sample: B=1, a removal loop shortened to ~n_evals denoiser evaluations, scaled per evaluation."""
    from mad_amd.init import seeded_state_dict
    from mad_amd.synthetic import synthetic_faces
    from oracle.scheduler_ref import RefScheduler
    from oracle.translation_ref import RefTranslation
    from oracle.unet_ref import unet_forward
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(cores, int(os.environ.get("OMP_NUM_THREADS", cores)))
    torch.set_num_threads(cores)
    sd = seeded_state_dict(spec, 0)
    kw = dict(img_resolution=spec.img_resolution, model_channels=spec.model_channels,
              channel_mult=spec.channel_mult, num_blocks=spec.num_blocks, label_dim=spec.label_dim)
    calls = [0]

    def model(x, t, labels):
        calls[0] += 1
        return unet_forward(sd, x, t, labels, **kw)

    s_cpu = max(2, (n_evals + 1) // 2)
    tr = RefTranslation(RefScheduler("ddpm"), spec.label_dim, sample_steps=1000, eta=cfg.EVAL.ETA)
    image = synthetic_faces(1, spec.img_resolution, seed=0)
    with torch.no_grad():
        model(image, torch.tensor([10.0]), tr.process_label(0, 1))      # warm the CPU kernels
        calls[0] = 0
        t0 = time.perf_counter()
        eps = tr.encode(image, tr.process_label(0, 1), model, start_from_step=s_cpu)
        tr.generate(model, eps_list=eps, class_label=tr.process_label(1, 1), source_image=image,
                    start_from_step=s_cpu)
        dt = time.perf_counter() - t0
    per_eval = dt / calls[0]
    evals_per_image = 2 * num_of_step - 1
    return {"value": 1.0 / (per_eval * evals_per_image), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"B=1 removal loop shortened to start_from_step={s_cpu} ({calls[0]} denoiser evaluations, "
                      f"{dt:.1f} s), scaled to the {evals_per_image} evaluations per image of start_from_step="
                      f"{num_of_step}",
            "evals_per_s": 1.0 / per_eval}


def hbm_kernels(dev, B, R, C, iters=20):
    """Achieved GB/s of the bandwidth-bound kernels on the bench shapes (outside the timed region):
    the sampler step (epsilon, x_t, z in, x_{t-1} out = 16 B per element, fp32) and the GroupNorm
    statistics / apply passes (2 B read; 2 B read + 2 B write per bf16 element)."""
    from mad_amd import ops
    from mad_amd._lib import SamplerCoef

    def timed(fn):
        for _ in range(3):
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
    eps, xt, z = (torch.randn(B, 3, R, R, device=dev) for _ in range(3))
    c = SamplerCoef(algo=0, pred_type=0, op=0, clip=1, use_clipped=0, add_variance=1, known_noise=1,
                    sqrt_a_t=0.8, sqrt_b_t=0.6, c0=0.1, c1=0.9, std=0.05, sqrt_a_prev=0.85, sqrt_b_prev=0.52,
                    dir_coef=0.0, clip_range=1.0, eps_div=1.0)
    t = timed(lambda: ops.sampler_update(c, eps, xt, noise=z))
    out["sampler_step"] = {"GB/s": 16.0 * eps.numel() / t / 1e9, "bytes_per_element": 16, "us": t * 1e6}
    x = torch.randn(B, R, R, C, device=dev).to(torch.bfloat16)
    coef = torch.rand(B, C, 2, device=dev)
    t = timed(lambda: ops.gn_partial(x))
    out["gn_partial"] = {"GB/s": 2.0 * x.numel() / t / 1e9, "bytes_per_element": 2, "us": t * 1e6}
    t = timed(lambda: ops.gn_apply(x, coef, act=True))
    out["gn_apply"] = {"GB/s": 4.0 * x.numel() / t / 1e9, "bytes_per_element": 4, "us": t * 1e6}
    return out


def main():
    args = parse_args()
    if args.workload == "train":
        # the DDP training step has its own driver (same launch contract and JSON line)
        import bench_train
        sys.argv = [sys.argv[0], "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup),
                    "--config", args.config, "--precision", args.precision] + (["--no-cpu-baseline"] if args.no_cpu_baseline else [])
        return bench_train.main()
    from mad_amd.launch import needs_relaunch, relaunch_per_gpu
    if needs_relaunch(args.gpus):
        # a plain `python bench.py --gpus N`: one child launcher, N ranks; this process never touches the GPU
        raise SystemExit(relaunch_per_gpu(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match the {world} ranks this process was started with")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    from mad_amd import ops
    from mad_amd.buildid import check_library_is_current
    from mad_amd.config import load_cfg
    from mad_amd.model import build_model
    from mad_amd.synthetic import synthetic_faces, synthetic_parsing_map
    build_id = check_library_is_current()          # a library built from other sources is not measured
    from mad_amd.translation import TranslationDiffusion

    cfg = load_cfg(args.config)
    model = build_model(cfg, precision=args.precision)
    model.unet.init_seeded(0)
    model = model.to(dev).eval()
    spec = model.spec
    R, B, S = spec.img_resolution, args.batch, args.num_of_step
    td = TranslationDiffusion(cfg, dev)
    td.generator = torch.Generator(device=dev).manual_seed(1234 + rank)
    # this rank's shard of the job: its own B images and parsing maps, resident in HBM
    images = synthetic_faces(B, R, seed=100 + rank).to(dev)
    maps = torch.stack([synthetic_parsing_map(R, seed=rank * B + i) for i in range(B)]).to(dev)

    def progress(msg):
        # a line on stderr now and then: the result line on stdout stays the only thing printed there
        if rank == 0:
            print(f"[bench {time.time() - T_PROCESS_START:6.1f}s] {msg}", file=sys.stderr, flush=True)

    progress(f"model built ({os.path.basename(args.config)}, batch {B}, {args.workload})")
    eval_count = [0]
    inner = model.forward
    model.graph_replay = bool(args.graph)

    def counted(*a, **k):
        # sample every n-th denoiser evaluation for per-kernel stream-event timing (launched one by one:
        # events cannot bracket kernels inside a replayed graph)
        sample = timed[0] and (eval_count[0] % args.profile_every == 0)
        ops.TIMER.active = sample
        sampled[0] += int(sample)
        eval_count[0] += 1
        if eval_count[0] % 500 == 0:
            progress(f"{eval_count[0]} denoiser evaluations")
        if sample:
            model.graph_replay = False
        out = inner(*a, **k)
        model.graph_replay = bool(args.graph)
        ops.TIMER.active = False
        return out

    timed = [False]
    sampled = [0]
    model.forward = counted

    guides = synthetic_faces(B, R, seed=900 + rank).to(dev) if args.workload == "transfer" else None
    if args.workload == "transfer":
        from mad_amd.translation import MAX_STEP
        S = MAX_STEP                                   # the transfer loop starts at its longest component step

    from mad_amd.translation import COMP_INFO, DEFAULT_COMP_SETUP

    def one_step(s_run=None):
        """One batch translated; `s_run` shortens the schedule (warm-up only)."""
        if args.workload == "transfer":
            setup = DEFAULT_COMP_SETUP
            if s_run is not None:      # same component order, steps scaled down to the short schedule
                setup = {c: COMP_INFO(ci.scale, max(2, ci.step * s_run // S)) for c, ci in DEFAULT_COMP_SETUP.items()}
            return td.image_translation(model, model, images, images, 1, 0, source_parsing_mask=maps, use_cam=True,
                                        inpainting=True, inverse_image=guides, comp_weight_step=setup)
        return td.domain_translation(model, model, images, source_class_label=0, target_class_label=1,
                                     parsing_mask=maps, start_from_step=S if s_run is None else s_run)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- warm-up: short schedules; the first also brings the full-size noise slab into the allocator's cache
    s_warm = max(2, min(S, args.warmup_num_of_step))
    per_eval = None
    for i in range(args.warmup):
        if i == 0:
            slab = torch.empty((S, B, 3, R, R), dtype=torch.float32, device=dev)
            del slab
        torch.cuda.synchronize()
        tw = time.perf_counter()
        one_step(s_warm)
        torch.cuda.synchronize()
        if i > 0 or args.warmup == 1:
            per_eval = (time.perf_counter() - tw) / (2 * s_warm - 1)
        progress(f"warm-up {i + 1}/{args.warmup} done")
    # ---- how many full steps fit the budget (a cap on --steps; every rank must run the same number)
    steps = max(1, args.steps)
    budget = args.time_budget
    if budget <= 0:
        after = 10.0 + (0.0 if (args.no_cpu_baseline or world > 1) else 30.0)   # lines after the timed region
        budget = args.total_budget - (time.time() - T_PROCESS_START) - after
    if per_eval is not None:
        est_step = per_eval * (2 * S - 1)
        steps = max(1, min(steps, int(budget / est_step)))
    if world > 1:
        t = torch.tensor([steps], device=dev, dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        steps = int(t.item())
    barrier()
    timed[0] = True
    t0 = time.perf_counter()
    progress(f"timing {steps} step(s) of {2 * S - 1} evaluations (asked for {args.steps}, budget {budget:.0f} s)")
    for i_step in range(steps):
        out = one_step()
        progress(f"step {i_step + 1}/{steps} issued")
    barrier()
    elapsed = time.perf_counter() - t0
    timed[0] = False
    assert tuple(out.shape) == (B, 3, R, R) and bool(torch.isfinite(out).all())
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern = ops.TIMER.summary()

    if rank == 0:
        flops_img_eval = spec.flops_per_image()["total"]
        evals = 2 * S - 1
        images_total = world * B * steps
        value = images_total / elapsed
        # dominant kernel = the MFMA kernel with the largest share of the sampled time
        mfma = {k: v for k, v in kern.items() if v["ms"] > 0}
        dom = max(mfma, key=lambda k: mfma[k]["ms"]) if mfma else None
        conv = mfma.get(dom, {"launches": 0, "flops": 0.0, "ms": 0.0})
        achieved = conv["flops"] / (conv["ms"] * 1e-3) / 1e12 if conv["ms"] > 0 else 0.0
        kernel_names = {"conv3x3_halo": "conv3x3_halo_kernel (3x3 convs with fused GroupNorm+SiLU loader)",
                        "conv_igemm": "conv_igemm_kernel (1x1 convs and low-resolution 3x3 convs)",
                        "attention": "attention_kernel", "conv_out": "conv3x3_halo_kernel (last layer)"}
        per_kernel = {k: {"tflops": v["flops"] / (v["ms"] * 1e-3) / 1e12, "ms_per_eval": v["ms"] / max(1, sampled[0]),
                          "launches": v["launches"]} for k, v in mfma.items()}
        # HBM bytes per launch of the dominant kernel come from separate PMC passes over this command
        # (tools/pmc_bench.sh); counters cannot be read from inside the timed run
        traffic, traffic_src, traffic_build = None, None, None
        tfile = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_halo_traffic.json")
        if dom == "conv3x3_halo" and R == 256 and os.path.exists(tfile):
            with open(tfile) as f:
                tj = json.load(f)
            traffic, traffic_src = tj.get("hbm_bytes_per_launch"), "profiles/pmc_halo_traffic.json (rocprofv3 --pmc FETCH_SIZE x2, WRITE_SIZE)"
            # the counters are a property of the library they were taken on: say which, and whether it is this one
            traffic_build = tj.get("build_id")
        line = {
            "metric": ("translated images/sec @256x256, fixed-step inversion+denoise (makeup removal)"
                       if R == 256 else f"translated images/sec @{R}x{R}, fixed-step inversion+denoise")
                      if args.workload == "removal" else
                      f"translated images/sec @{R}x{R}, makeup transfer (inpainting + component masks)",
            "value": value, "unit": "images/s", "n_gpus": world, "steps": steps, "steps_requested": args.steps,
            "warmup": args.warmup, "warmup_num_of_step": s_warm if args.warmup else None,
            "time_budget_s": round(budget, 1), "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic", "build_id": build_id,
            "config": {"workload": f"{os.path.basename(args.config)} "
                                   + ("makeup removal (noise encoding + regeneration), " if args.workload == "removal"
                                      else "makeup transfer (noise encoding + regeneration from a synthetic guide, "
                                           "inpainting with per-step component masks), ")
                                   + f"DDPM 1000-step table, start_from_step={S} ({evals} denoiser evaluations per "
                                   f"image), batch {B} per GPU, random-init (seeded) weights",
                       "batch_per_gpu": B, "num_of_step": S, "resolution": R, "parallelism": f"dp{world}",
                       "denoiser_evals_per_s": world * steps * evals / elapsed,
                       "gflop_per_image_per_eval": flops_img_eval / 1e9},
            "mfma_util_end_to_end": value * flops_img_eval * evals / 1e12 / (PEAK_BF16_TFLOPS * world),
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_BF16_TFLOPS, "traffic": traffic, "traffic_unit": "bytes/launch",
                         "traffic_source": traffic_src, "traffic_build_id": traffic_build,
                         "traffic_from_this_build": traffic_build == build_id if traffic_build else None,
                         "kernel": kernel_names.get(dom, dom),
                         "launches_timed": conv["launches"],
                         "avg_launch_ms": conv["ms"] / conv["launches"] if conv["launches"] else None,
                         "mfma_kernels": per_kernel},
        }
        progress("timed region done; bandwidth-bound kernels")
        line["hbm_kernels"] = hbm_kernels(dev, B, R, spec.model_channels)
        if not args.no_cpu_baseline and world == 1:
            progress(f"CPU baseline ({args.cpu_evals} evaluations of the oracle)")
            line["cpu_baseline"] = cpu_baseline(cfg, spec, S, args.cpu_evals)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
