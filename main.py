#!/usr/bin/env python3
"""Training entry point on the HIP path: the reference's ``main.py`` (``/root/reference/main.py``) with
the same flags (``--config --generate-only --save-file-name --opts``), config keys and checkpoint
format, and the loop body replaced by ``mad_amd.train.UNetTrainer``.

    python main.py --config configs/model_256_256.yaml --opts TRAIN.ROOT "['data/mtdataset/images']"
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 main.py --config ...

One process per GPU; with a process group (backend nccl = RCCL) every rank reads its own slice of the
shuffled image list and the flat gradient is summed over the ranks once per optimizer step.
Image folders follow the reference's dataset (``dataset/makeup_dataset.py:51-66``): ``<root>/makeup`` is
label 0, ``<root>/non-makeup`` label 1, 90/10 train/validation split with seed 42.  ``--synthetic N``
trains on N generated images instead (no dataset on the machine).  Checkpoints hold ``state_dict``,
``optimizer``, ``lr_scheduler``, ``iter`` and ``ema_state_dict`` (``main.py:278-288``); the text branch of
the dataset (CLIP token ids) is unused by this denoiser and not produced.
"""
import argparse
import glob
import os
import os.path as osp
import random
import sys
import time

import torch

REPO = osp.dirname(osp.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from mad_amd.config import create_cfg, merge_possible_with_base, show_config  # noqa: E402


def parse_args(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--config", default=None, type=str)
    parser.add_argument("--generate-only", action="store_true", default=False)
    parser.add_argument("--save-file-name", default="generated.png", type=str)
    parser.add_argument("--synthetic", default=0, type=int, help="train on this many generated images")
    parser.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    parser.add_argument("--seed", default=0, type=int)
    parser.add_argument("--opts", nargs=argparse.REMAINDER, default=None, type=str)
    return parser.parse_args(argv)


# ---------------------------------------------------------------------------------------- data
def list_images(cfg):
    """(path, label) pairs of the training split, as MakeupDataset builds it."""
    roots = cfg.TRAIN.ROOT if isinstance(cfg.TRAIN.ROOT, (list, tuple)) else [cfg.TRAIN.ROOT]
    use_label = cfg.MODEL.LABEL_DIM > 0
    files = []
    for root in roots:
        cats = ["", "makeup", "non-makeup"] if use_label else ["makeup" if cfg.TRAIN.MAKEUP else "non-makeup"]
        for cat in cats:
            d = osp.join(root, cat)
            if osp.exists(d):
                files += glob.glob(osp.join(d, "*.png")) + glob.glob(osp.join(d, "*.jpg"))
    files.sort()
    if not files:
        raise SystemExit(f"no training images under {roots} (or pass --synthetic N)")
    from sklearn.model_selection import train_test_split
    train, _ = train_test_split(files, test_size=0.1, random_state=42)
    return [(f, 1 if osp.basename(osp.dirname(f)) == "non-makeup" else 0) for f in train]


def load_image(path: str, size: int, flip: bool) -> torch.Tensor:
    """Resize((size, size)) -> RandomHorizontalFlip -> ToTensor -> Normalize(0.5, 0.5)  (main.py:162-169)."""
    import numpy as np
    from PIL import Image
    img = Image.open(path).convert("RGB").resize((size, size), Image.BILINEAR)
    a = torch.from_numpy(np.asarray(img).copy()).permute(2, 0, 1).float() / 255.0
    if flip:
        a = a.flip(2)
    return (a - 0.5) / 0.5


class Batches:
    """Endless shuffled batches of this rank's slice of the list (drop_last), or synthetic images."""

    def __init__(self, cfg, n_images, rank, world, synthetic, seed):
        self.cfg, self.n, self.rank, self.world = cfg, n_images, rank, world
        self.rng = random.Random(seed * 1000 + rank)
        self.size = cfg.TRAIN.IMAGE_SIZE
        self.label_dim = cfg.MODEL.LABEL_DIM
        if synthetic:
            from mad_amd.synthetic import synthetic_faces
            self.images = synthetic_faces(synthetic, self.size, seed=seed)
            self.items = [(i, i % max(1, self.label_dim)) for i in range(synthetic)][rank::world]
        else:
            self.images = None
            self.items = list_images(cfg)[rank::world]
        if len(self.items) < n_images:
            raise SystemExit(f"rank {rank}: {len(self.items)} images are fewer than one optimizer step ({n_images})")
        self.order, self.pos = [], 0

    def next(self):
        if self.pos + self.n > len(self.order):
            self.order = list(range(len(self.items)))
            self.rng.shuffle(self.order)
            self.pos = 0
        pick = [self.items[i] for i in self.order[self.pos:self.pos + self.n]]
        self.pos += self.n
        if self.images is not None:
            x = torch.stack([self.images[i].flip(2) if self.rng.random() < 0.5 else self.images[i] for i, _ in pick])
        else:
            x = torch.stack([load_image(f, self.size, self.rng.random() < 0.5) for f, _ in pick])
        lab = None
        if self.label_dim:
            lab = torch.nn.functional.one_hot(torch.tensor([l for _, l in pick]), self.label_dim).float()
        return x, lab


# ---------------------------------------------------------------------------------------- sampling
@torch.no_grad()
def evaluate(cfg, model, scheduler, device, filename, rank, world):
    """main.py:42-93: EVAL.BATCH_SIZE images from noise with the configured sampler, saved as one grid."""
    from PIL import Image
    model.eval()
    n = cfg.EVAL.BATCH_SIZE
    g = torch.Generator(device=device).manual_seed(1234 + rank)
    images = torch.randn((n, cfg.MODEL.IN_CHANNELS, cfg.TRAIN.IMAGE_SIZE, cfg.TRAIN.IMAGE_SIZE), device=device, generator=g)
    labels = None
    if cfg.MODEL.LABEL_DIM > 0:
        idx = torch.tensor([0] * (n // 2) + [1] * (n - n // 2), device=device)
        labels = torch.nn.functional.one_hot(idx, cfg.MODEL.LABEL_DIM).float()
    scheduler.set_timesteps(cfg.EVAL.SAMPLE_STEPS, device=device)
    for t in scheduler.timesteps:
        out = model(images, t.reshape(-1), text=None, class_labels=labels).float()
        if cfg.EVAL.SCHEDULER == "ddim":
            images = scheduler.step(out, t, images, use_clipped_model_output=True, eta=cfg.EVAL.ETA,
                                    generator=g).prev_sample
        else:
            images = scheduler.step(out, t, images, generator=g).prev_sample
    if world > 1:
        import torch.distributed as dist
        parts = [torch.empty_like(images) for _ in range(world)]
        dist.all_gather(parts, images)
        images = torch.cat(parts, 0)
    if rank == 0:
        im = ((images.float().clamp(-1, 1) + 1) / 2 * 255).round().to(torch.uint8).cpu()
        cols = 4
        rows = max(1, im.shape[0] // cols)
        R = im.shape[2]
        grid = torch.zeros((3, rows * R, cols * R), dtype=torch.uint8)
        for i in range(min(im.shape[0], rows * cols)):
            grid[:, (i // cols) * R:(i // cols + 1) * R, (i % cols) * R:(i % cols + 1) * R] = im[i]
        Image.fromarray(grid.permute(1, 2, 0).numpy()).save(filename)
        print(f"Save generated samples to {filename}...", flush=True)
    model.train()


# ---------------------------------------------------------------------------------------- checkpoints
def save_checkpoint(path, model, trainer, cur_iter):
    """The dictionary of main.py:278-284 (optimizer state in torch.optim.AdamW's layout, parameter order)."""
    params = list(model.unet.named_parameters())
    state, off = {}, 0
    for i, (_, p) in enumerate(params):
        n = p.numel()
        state[i] = {"step": torch.tensor(float(trainer.step_count)),
                    "exp_avg": trainer.m[off:off + n].view(p.shape).cpu().clone(),
                    "exp_avg_sq": trainer.v[off:off + n].view(p.shape).cpu().clone()}
        off += n
    shadow, off = [], 0
    for _, p in params:
        n = p.numel()
        shadow.append(trainer.ema[off:off + n].view(p.shape).cpu().clone())
        off += n
    max_decay, inv_gamma, power = trainer.ema_cfg
    torch.save({
        "state_dict": {k: v.detach().cpu().clone() for k, v in model.unet.state_dict().items()},
        "optimizer": {"state": state, "param_groups": [{"lr": trainer.lr, "betas": tuple(trainer.betas), "eps": trainer.eps,
                                                         "weight_decay": trainer.weight_decay,
                                                         "params": list(range(len(params)))}]},
        "lr_scheduler": {"last_epoch": trainer.step_count},
        "iter": cur_iter,
        "ema_state_dict": {"decay": max_decay, "min_decay": 0.0, "optimization_step": trainer.step_count,
                           "update_after_step": 0, "use_ema_warmup": True, "inv_gamma": inv_gamma, "power": power,
                           "shadow_params": shadow},
    }, path)


def resume(path, model, trainer):
    """main.py:185-194: weights, AdamW moments, step counters and the EMA copy."""
    st = torch.load(path, map_location="cpu", weights_only=True)
    model.unet.load_state_dict(st["state_dict"])          # parameters are views of trainer.flat: copied in place
    dev = trainer.flat.device
    off = 0
    for i, (_, p) in enumerate(model.unet.named_parameters()):
        n = p.numel()
        s = st["optimizer"]["state"][i]
        trainer.m[off:off + n].copy_(s["exp_avg"].reshape(-1).to(dev))
        trainer.v[off:off + n].copy_(s["exp_avg_sq"].reshape(-1).to(dev))
        trainer.ema[off:off + n].copy_(st["ema_state_dict"]["shadow_params"][i].reshape(-1).to(dev))
        off += n
    trainer.step_count = int(st["ema_state_dict"]["optimization_step"])
    model.unet.invalidate()
    return int(st["iter"]) + 1


# ---------------------------------------------------------------------------------------- main
def main(argv=None):
    args = parse_args(argv)
    cfg = create_cfg()
    if args.config:
        merge_possible_with_base(cfg, args.config)
    if args.opts:
        cfg.merge_from_list(args.opts)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=device)
    if rank == 0:
        show_config(cfg)
        os.makedirs(osp.join(cfg.PROJECT_DIR, "checkpoints"), exist_ok=True)
        os.makedirs(osp.join(cfg.PROJECT_DIR, "generate"), exist_ok=True)

    from mad_amd.checkpoint import load_checkpoint
    from mad_amd.model import build_model
    from mad_amd.scheduler import CustomDDIMScheduler, CustomDDPMScheduler
    from mad_amd.train import UNetTrainer, warmup_lr

    model = build_model(cfg, precision=args.precision)
    if cfg.TRAIN.RESUME is None and cfg.MODEL.PRETRAINED is not None:
        load_checkpoint(model, cfg.MODEL.PRETRAINED, use_ema=False, strict=False)
    elif cfg.TRAIN.RESUME is None:
        model.unet.init_seeded(args.seed)
    model = model.to(device).train()
    sched_cls = {"ddpm": CustomDDPMScheduler, "ddim": CustomDDIMScheduler}[cfg.EVAL.SCHEDULER]
    scheduler = sched_cls(num_train_timesteps=cfg.TRAIN.SAMPLE_STEPS, prediction_type=cfg.TRAIN.NOISE_SCHEDULER.PRED_TYPE,
                          beta_schedule=cfg.TRAIN.NOISE_SCHEDULER.TYPE, beta_start=cfg.TRAIN.NOISE_SCHEDULER.BETA_START,
                          beta_end=cfg.TRAIN.NOISE_SCHEDULER.BETA_END)
    if cfg.TRAIN.NOISE_SCHEDULER.PRED_TYPE != "epsilon":
        raise ValueError("Not supported prediction type.")                      # main.py:242-245
    trainer = UNetTrainer(model.unet, lr=cfg.TRAIN.LR, dropout=0.10, ema_max_decay=cfg.TRAIN.EMA_MAX_DECAY,
                          ema_inv_gamma=cfg.TRAIN.EMA_INV_GAMMA, ema_power=cfg.TRAIN.EMA_POWER, seed=args.seed * 64 + rank)
    start_iter = 0
    if cfg.TRAIN.RESUME is not None:
        assert osp.exists(cfg.TRAIN.RESUME), "Resume file not found."
        start_iter = resume(cfg.TRAIN.RESUME, model, trainer)
        if rank == 0:
            print(f"Resume checkpoint from {cfg.TRAIN.RESUME}...", flush=True)

    def with_ema(fn):
        """ema_model.store / copy_to / restore around sampling (main.py:293-297)."""
        keep = trainer.flat.clone()
        trainer.flat.copy_(trainer.ema)
        model.unet.invalidate()
        try:
            fn()
        finally:
            trainer.flat.copy_(keep)
            model.unet.invalidate()

    if args.generate_only:
        with_ema(lambda: evaluate(cfg, model, scheduler, device, args.save_file_name, rank, world))
        return

    accum = cfg.TRAIN.GRADIENT_ACCUMULATION_STEPS
    per_step = cfg.TRAIN.BATCH_SIZE * accum
    data = Batches(cfg, per_step, rank, world, args.synthetic, args.seed)
    gen = torch.Generator(device=device).manual_seed(args.seed * 7919 + rank)
    alphas = scheduler.alphas_cumprod
    # One iteration of the reference is one micro-batch; an optimizer step closes every `accum` of them.  Here
    # the accum * BATCH_SIZE images of one optimizer step are one forward/backward pass (same mean gradient),
    # so the loop advances `accum` reference iterations at a time and keeps the reference's interval units.
    t_avg, n_avg = 0.0, 0
    cur_iter = start_iter
    max_iter = cfg.TRAIN.MAX_ITER
    while cur_iter < max_iter:
        t0 = time.time()
        x, lab = data.next()
        loss = trainer.train_step(x.to(device), None if lab is None else lab.to(device), alphas, generator=gen,
                                  num_timesteps=cfg.TRAIN.TIME_STEPS, accumulation=1, warmup_steps=cfg.TRAIN.LR_WARMUP)
        prev, cur_iter = cur_iter, min(max_iter, cur_iter + accum)
        crossed = lambda k: k > 0 and (cur_iter // k) > (prev // k)
        last = cur_iter == max_iter
        if crossed(cfg.TRAIN.LOG_INTERVAL) or last:
            lv = float(loss)
            t_avg, n_avg = t_avg + (time.time() - t0), n_avg + 1
            if rank == 0:
                lr_now = warmup_lr(cfg.TRAIN.LR, trainer.step_count - 1, cfg.TRAIN.LR_WARMUP)
                print(f"iter: [{cur_iter}/{max_iter}]\ttime: {t_avg / n_avg:.3f}\tlr: {lr_now:.2e}\tloss: {lv:.4f}", flush=True)
        if (crossed(cfg.TRAIN.SAVE_INTERVAL) or last) and rank == 0:
            name = f"checkpoint_{cur_iter}.pth" if not last else "final.pth"
            save_checkpoint(osp.join(cfg.PROJECT_DIR, "checkpoints", name), model, trainer, cur_iter - 1)
            print(f"Save checkpoint to {name}...", flush=True)
        if crossed(cfg.TRAIN.SAMPLE_INTERVAL) or last:
            fn = osp.join(cfg.PROJECT_DIR, "generate", f"iter_{cur_iter:03d}.png")
            with_ema(lambda: evaluate(cfg, model, scheduler, device, fn, rank, world))
        if world > 1:
            dist.barrier()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
