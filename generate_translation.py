#!/usr/bin/env python3
"""Makeup removal / beauty filter over a list of images, on the HIP path.

Same flags and behaviour as the reference script (``/root/reference/generate_translation.py``):
the list is cut into ``--num-process`` contiguous shards (``len // n`` each, remainder to the last),
one worker process per GPU ``cuda:{i}``, output ``pred_{idx}.png`` with the global index, existing
outputs are skipped (idempotent resume), a failing image is logged and counted, not fatal.
Additions: ``--batch`` images are translated together per denoiser call, ``--checkpoint`` loads a
``state_dict`` (reference key names), ``--seed`` seeds the random initialiser when none is given.
"""
import argparse
import os
import sys
from typing import List, Tuple

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from mad_amd.config import create_cfg, merge_possible_with_base, show_config  # noqa: E402
from mad_amd.shard import shard_list  # noqa: E402


def parse_args(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--config", default=None, type=str)
    parser.add_argument("--save-folder", default="batch_images", type=str)
    parser.add_argument("--source-root", required=True, type=str)
    parser.add_argument("--source-list", required=True, type=str)
    parser.add_argument("--source-label", required=True, type=int)
    parser.add_argument("--target-label", required=True, type=int)
    parser.add_argument("--num-process", default=1, type=int)
    parser.add_argument("--num-of-step", default=700, type=int)
    parser.add_argument("--batch", default=8, type=int)
    parser.add_argument("--checkpoint", default=None, type=str)
    parser.add_argument("--seed", default=0, type=int)
    parser.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    parser.add_argument("--opts", nargs=argparse.REMAINDER, default=None, type=str)
    return parser.parse_args(argv)


def load_model(cfg, device, checkpoint, seed, precision):
    from mad_amd.model import build_model
    model = build_model(cfg, precision=precision)
    if checkpoint:
        from mad_amd.checkpoint import load_checkpoint
        load_checkpoint(model, checkpoint)           # EMA weights by position if present, else by name
    else:
        model.unet.init_seeded(seed)
    return model.to(device).eval()


def save_png(image: torch.Tensor, path: str) -> None:
    from PIL import Image
    Image.fromarray((image.permute(1, 2, 0).cpu().numpy() * 255).astype("uint8")).save(path)


def generate_image(cfg, save_folder: str, source_list: List[Tuple[str, str]], source_label: int, target_label: int,
                   offset: int, device: str, num_of_step: int, batch: int = 8, checkpoint=None, seed: int = 0,
                   precision: str = "bf16") -> int:
    """Translate this worker's shard; returns the number of failed images."""
    from mad_amd.translation import TranslationDiffusion
    torch.cuda.set_device(device)
    model = load_model(cfg, device, checkpoint, seed, precision)
    diffuser = TranslationDiffusion(cfg, device)
    os.makedirs(save_folder, exist_ok=True)
    todo = []
    for idx, (image, mask) in enumerate(source_list):
        name = os.path.join(save_folder, f"pred_{idx + offset}.png")
        if os.path.exists(name):
            continue
        if mask.endswith("jpg"):
            mask = mask.replace("jpg", "png")
        todo.append((name, image, mask))
    def run(group) -> None:
        images = torch.cat([diffuser.process_image(p) for _, p, _ in group], 0)
        size = tuple(images.shape[-2:][::-1])
        masks = torch.cat([diffuser._label_map(m, size) for _, _, m in group], 0)
        out = diffuser.domain_translation(source_model=model, target_model=model, source_image=images,
                                          source_class_label=source_label, target_class_label=target_label,
                                          parsing_mask=masks, start_from_step=num_of_step)
        for (name, _, _), img in zip(group, out):
            save_png(img, name)

    count_error = 0
    for i in range(0, len(todo), batch):
        group = todo[i:i + batch]
        try:
            run(group)
        except Exception as e:
            # per-item isolation, as the reference: a bad file costs only itself
            for item in group if len(group) > 1 else []:
                try:
                    run([item])
                except Exception as e1:
                    print(f"[{device}] error on {item[1]}: {e1}", file=sys.stderr)
                    count_error += 1
            if len(group) == 1:
                print(f"[{device}] error on {group[0][1]}: {e}", file=sys.stderr)
                count_error += 1
    if count_error:
        print(f"Error in {device}: {count_error}")
    return count_error


def read_list(source_root: str, source_list: str) -> List[Tuple[str, str]]:
    out = []
    with open(source_list, "r") as f:
        for line in f.readlines():
            line = line.strip()
            if not line:
                continue
            image_path = os.path.join(source_root, line)
            out.append((image_path, image_path.replace("images", "parsing")))
    return out


def main(argv=None):
    args = parse_args(argv)
    cfg = create_cfg()
    if args.config:
        merge_possible_with_base(cfg, args.config)
    if args.opts:
        cfg.merge_from_list(args.opts)
    show_config(cfg)
    source_list = read_list(args.source_root, args.source_list)
    jobs = []
    for gpu_idx in range(args.num_process):
        shard, offset = shard_list(source_list, args.num_process, gpu_idx)
        jobs.append(dict(cfg=cfg, save_folder=args.save_folder, source_list=shard, source_label=args.source_label,
                         target_label=args.target_label, offset=offset, device=f"cuda:{gpu_idx}",
                         num_of_step=args.num_of_step, batch=args.batch, checkpoint=args.checkpoint, seed=args.seed,
                         precision=args.precision))
    if args.num_process == 1:
        generate_image(**jobs[0])
    else:
        import torch.multiprocessing as mp
        ctx = mp.get_context("spawn")          # one process per GPU, no shared state
        procs = [ctx.Process(target=generate_image, kwargs=j) for j in jobs]
        for p in procs:
            p.start()
        for p in procs:
            p.join()
        failed = [(i, p.exitcode) for i, p in enumerate(procs) if p.exitcode != 0]
        if failed:          # a worker that crashed must not look like a finished shard
            raise SystemExit(f"worker(s) failed (index, exit code): {failed}")


if __name__ == "__main__":
    main()
