#!/usr/bin/env python3
"""Makeup transfer over paired lists of images, on the HIP path.

Flags follow the reference script (``/root/reference/generate_transfer.py``): ``--inpainting`` keeps
the source outside the edited components, ``--cam`` switches the components on step by step.
The guide image the reference builds on the host (landmark detection, histogram matching and face
morphing with mediapipe / dlib / OpenCV) is outside this library: ``--guide-root`` names a folder
holding one pre-morphed guide per source file name.  Work is sharded as in the reference: contiguous
``len // n`` slices, remainder to the last worker, one process per GPU (generate_transfer.py:118-147).  Inside a
worker ``--batch`` pairs go through ONE ``image_translation`` call (stacked sources, guides and label maps: the
reference translates one pair per call, :33-91; every op on the path is per-sample, so the images come out as they
do alone); a group that fails is retried pair by pair, so a bad file costs only itself.
"""
import argparse
import os
import sys
from typing import List, Tuple

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from generate_translation import load_model, save_png  # noqa: E402
from mad_amd.config import create_cfg, merge_possible_with_base, show_config  # noqa: E402
from mad_amd.shard import shard_list  # noqa: E402


def parse_args(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--config", default=None, type=str)
    parser.add_argument("--save-folder", default="batch_images", type=str)
    parser.add_argument("--source-root", required=True, type=str)
    parser.add_argument("--target-root", required=True, type=str)
    parser.add_argument("--guide-root", required=True, type=str)
    parser.add_argument("--source-list", required=True, type=str)
    parser.add_argument("--target-list", required=True, type=str)
    parser.add_argument("--source-label", required=True, type=int)
    parser.add_argument("--target-label", required=True, type=int)
    parser.add_argument("--num-process", default=1, type=int)
    parser.add_argument("--inpainting", action="store_true", default=False)
    parser.add_argument("--cam", action="store_true", default=False)
    parser.add_argument("--batch", default=8, type=int, help="pairs translated together per GPU")
    parser.add_argument("--checkpoint", default=None, type=str)
    parser.add_argument("--seed", default=0, type=int)
    parser.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    parser.add_argument("--opts", nargs=argparse.REMAINDER, default=None, type=str)
    return parser.parse_args(argv)


def generate_image(cfg, save_folder: str, pairs: List[Tuple[str, str, str, str]], guide_root: str, source_label: int,
                   target_label: int, offset: int, device: str, inpainting: bool, use_cam: bool, batch: int = 8,
                   checkpoint=None, seed: int = 0, precision: str = "bf16") -> int:
    """Translate this worker's shard of pairs, `batch` at a time; returns the number of failed pairs."""
    from mad_amd.translation import TranslationDiffusion
    torch.cuda.set_device(device)
    model = load_model(cfg, device, checkpoint, seed, precision)
    diffuser = TranslationDiffusion(cfg, device)
    os.makedirs(save_folder, exist_ok=True)
    todo = []
    for idx, (source_image, source_mask, target_image, target_mask) in enumerate(pairs):
        name = os.path.join(save_folder, f"pred_{idx + offset}.png")
        if os.path.exists(name):
            continue
        if source_mask.endswith("jpg"):
            source_mask = source_mask.replace("jpg", "png")
        todo.append((name, source_image, source_mask, target_image, target_mask))

    def run(group) -> None:
        sources = torch.cat([diffuser.process_image(s) for _, s, _, _, _ in group], 0)
        targets = torch.cat([diffuser.process_image(t) for _, _, _, t, _ in group], 0)
        guides = torch.cat([diffuser.process_image(os.path.join(guide_root, os.path.basename(s)))
                            for _, s, _, _, _ in group], 0)
        size = tuple(sources.shape[-2:][::-1])
        maps = torch.cat([diffuser._label_map(m, size) for _, _, m, _, _ in group], 0)
        out = diffuser.image_translation(source_model=model, target_model=model, source_image=sources,
                                         target_image=targets, source_class_label=source_label,
                                         target_class_label=target_label, source_parsing_mask=maps,
                                         target_parsing_mask=None, use_morphing=True, use_encode_eps=True,
                                         use_cam=use_cam, inpainting=inpainting, inverse_image=guides)
        for item, img in zip(group, out):
            save_png(img, item[0])

    count_error = 0
    for i in range(0, len(todo), max(1, batch)):
        group = todo[i:i + max(1, batch)]
        try:
            run(group)
        except Exception as e:
            # per-pair isolation, as the reference's try/except per pair (generate_transfer.py:66-84)
            for item in group if len(group) > 1 else []:
                try:
                    run([item])
                except Exception as e1:
                    print(f"Error in {device}: {item[1]}: {e1}", file=sys.stderr)
                    count_error += 1
            if len(group) == 1:
                print(f"Error in {device}: {group[0][1]}: {e}", file=sys.stderr)
                count_error += 1
    if count_error:
        print(f"Error in {device}: {count_error}")
    return count_error


def main(argv=None):
    args = parse_args(argv)
    cfg = create_cfg()
    if args.config:
        merge_possible_with_base(cfg, args.config)
    if args.opts:
        cfg.merge_from_list(args.opts)
    show_config(cfg)
    pairs = []
    with open(args.source_list) as fs, open(args.target_list) as ft:
        for s, t in zip(fs.readlines(), ft.readlines()):
            if s.strip() == "" or t.strip() == "":
                continue
            sp, tp = os.path.join(args.source_root, s.strip()), os.path.join(args.target_root, t.strip())
            pairs.append((sp, sp.replace("images", "parsing"), tp, tp.replace("images", "parsing")))
    jobs = []
    for gpu_idx in range(args.num_process):
        shard, offset = shard_list(pairs, args.num_process, gpu_idx)
        jobs.append(dict(cfg=cfg, save_folder=args.save_folder, pairs=shard, guide_root=args.guide_root,
                         source_label=args.source_label, target_label=args.target_label, offset=offset,
                         device=f"cuda:{gpu_idx}", inpainting=args.inpainting, use_cam=args.cam,
                         batch=args.batch, checkpoint=args.checkpoint, seed=args.seed, precision=args.precision))
    if args.num_process == 1:
        generate_image(**jobs[0])
    else:
        import torch.multiprocessing as mp
        ctx = mp.get_context("spawn")
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
