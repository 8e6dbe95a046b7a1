"""The generate_translation entry point end to end on tiny synthetic files (GPU only)."""
import os

import numpy as np
import pytest
import torch

from conftest import REPO

pytestmark = pytest.mark.gpu


def test_generate_translation_cli(tmp_path):
    from PIL import Image
    import generate_translation as gt
    from mad_amd.synthetic import synthetic_faces, synthetic_parsing_map
    img_dir, par_dir = tmp_path / "images", tmp_path / "parsing"
    img_dir.mkdir(); par_dir.mkdir()
    faces = synthetic_faces(3, 16, seed=3)
    names = []
    for i in range(3):
        arr = ((faces[i].permute(1, 2, 0).numpy() * 0.5 + 0.5) * 255).astype("uint8")
        Image.fromarray(arr).save(img_dir / f"f{i}.png")
        Image.fromarray(synthetic_parsing_map(16, seed=i).numpy()).save(par_dir / f"f{i}.png")
        names.append(f"f{i}.png")
    names.append("missing.png")                                   # a broken entry must not stop the run
    (tmp_path / "list.txt").write_text("\n".join(names) + "\n")
    out = tmp_path / "out"
    argv = ["--config", os.path.join(REPO, "configs", "model_tiny.yaml"), "--save-folder", str(out),
            "--source-root", str(img_dir), "--source-list", str(tmp_path / "list.txt"), "--source-label", "0",
            "--target-label", "1", "--num-of-step", "6", "--batch", "2"]
    gt.main(argv)
    made = sorted(os.listdir(out))
    assert made == ["pred_0.png", "pred_1.png", "pred_2.png"]
    first = np.array(Image.open(out / "pred_0.png"))
    assert first.shape == (16, 16, 3)
    stamp = os.path.getmtime(out / "pred_0.png")
    gt.main(argv)                                                 # second run skips what exists
    assert os.path.getmtime(out / "pred_0.png") == stamp


def test_generate_transfer_cli(tmp_path):
    """Transfer entry point: paired lists, parsing maps found by the images -> parsing path rule,
    guides from --guide-root, --inpainting --cam, one unreadable pair skipped."""
    from PIL import Image
    import generate_transfer as gtr
    from mad_amd.synthetic import synthetic_faces, synthetic_parsing_map
    img_dir, par_dir, guide_dir = tmp_path / "images", tmp_path / "parsing", tmp_path / "guides"
    for d in (img_dir, par_dir, guide_dir):
        d.mkdir()
    faces = synthetic_faces(4, 16, seed=5)
    for i in range(4):
        arr = ((faces[i].permute(1, 2, 0).numpy() * 0.5 + 0.5) * 255).astype("uint8")
        Image.fromarray(arr).save(img_dir / f"p{i}.png")
        Image.fromarray(arr[::-1].copy()).save(guide_dir / f"p{i}.png")
        Image.fromarray(synthetic_parsing_map(16, seed=i).numpy()).save(par_dir / f"p{i}.png")
    (tmp_path / "src.txt").write_text("p0.png\nnope.png\np1.png\np3.png\n")
    (tmp_path / "tgt.txt").write_text("p2.png\np2.png\np3.png\np0.png\n")
    out = tmp_path / "out"
    argv = ["--config", os.path.join(REPO, "configs", "model_tiny.yaml"), "--save-folder", str(out),
            "--source-root", str(img_dir), "--target-root", str(img_dir), "--guide-root", str(guide_dir),
            "--source-list", str(tmp_path / "src.txt"), "--target-list", str(tmp_path / "tgt.txt"),
            "--source-label", "1", "--target-label", "0", "--inpainting", "--cam"]
    # pairs go two at a time; the group holding the unreadable pair falls back to pair-by-pair and loses only it
    gtr.main(argv + ["--batch", "2"])
    assert sorted(os.listdir(out)) == ["pred_0.png", "pred_2.png", "pred_3.png"]
    img = np.array(Image.open(out / "pred_2.png"))
    assert img.shape == (16, 16, 3) and img.std() > 0


def test_transfer_batch_equals_pair_by_pair():
    """Stacked sources, guides and label maps through ONE image_translation call give each pair the image it gets
    alone (same draws per image): what --batch relies on.  fp32 mode, tolerance 1e-5 on [0, 1] images."""
    from mad_amd.config import load_cfg
    from mad_amd.model import build_model
    from mad_amd.synthetic import synthetic_faces, synthetic_parsing_map
    from mad_amd.translation import TranslationDiffusion
    dev = torch.device("cuda:0")
    cfg = load_cfg(os.path.join(REPO, "configs", "model_tiny.yaml"))
    m = build_model(cfg, precision="fp32")
    m.unet.init_seeded(2)
    m = m.to(dev).eval()
    B, R = 3, 16
    src, guide = synthetic_faces(B, R, seed=7).to(dev), synthetic_faces(B, R, seed=8).to(dev)
    maps = torch.stack([synthetic_parsing_map(R, seed=i) for i in range(B)]).to(dev)
    setup = {4: (0.4, 9), 9: (0.8, 5), 1: (0.8, 7), 6: (0.8, 7)}
    pool = torch.randn(24, B, 3, R, R, generator=torch.Generator().manual_seed(3)).to(dev)

    def run(rows):
        td = TranslationDiffusion(cfg, dev)
        k = iter(range(pool.shape[0]))
        td.noise_source = lambda shape: pool[next(k)][rows]
        return td.image_translation(m, m, src[rows].contiguous(), src[rows].contiguous(), 1, 0,
                                    source_parsing_mask=maps[rows].contiguous(), use_cam=True, inpainting=True,
                                    inverse_image=guide[rows].contiguous(), comp_weight_step=setup)

    together = run(slice(0, B))
    for i in range(B):
        alone = run(slice(i, i + 1))
        assert float((alone - together[i:i + 1]).abs().max()) < 1e-5, i


def test_training_cli(tmp_path):
    """main.py on synthetic images: iterations, checkpoint in the reference's format, resume, EMA sampling."""
    import main as train_main
    from mad_amd.checkpoint import load_checkpoint
    from mad_amd.config import load_cfg
    from mad_amd.model import build_model
    proj = tmp_path / "run"
    common = ["--config", os.path.join(REPO, "configs", "model_tiny.yaml"), "--synthetic", "16", "--opts",
              "PROJECT_DIR", str(proj), "TRAIN.BATCH_SIZE", "2", "TRAIN.GRADIENT_ACCUMULATION_STEPS", "2",
              "TRAIN.LOG_INTERVAL", "2", "TRAIN.SAVE_INTERVAL", "4", "TRAIN.SAMPLE_INTERVAL", "1000",
              "TRAIN.LR_WARMUP", "2", "EVAL.SAMPLE_STEPS", "5"]
    train_main.main(common + ["TRAIN.MAX_ITER", "6"])
    ck = proj / "checkpoints"
    assert sorted(os.listdir(ck)) == ["checkpoint_4.pth", "final.pth"]
    assert os.path.exists(proj / "generate" / "iter_006.png")
    st = torch.load(ck / "final.pth", map_location="cpu", weights_only=True)
    assert set(st) == {"state_dict", "optimizer", "lr_scheduler", "iter", "ema_state_dict"} and st["iter"] == 5
    cfg = load_cfg(os.path.join(REPO, "configs", "model_tiny.yaml"))
    m = build_model(cfg)
    assert len(st["ema_state_dict"]["shadow_params"]) == len(list(m.unet.parameters()))
    assert load_checkpoint(m, str(ck / "final.pth"))["source"] == "ema_state_dict.shadow_params"
    assert load_checkpoint(m, str(ck / "final.pth"), use_ema=False)["iter"] == 5
    # resume from the intermediate checkpoint and finish; then sample only
    train_main.main(common + ["TRAIN.MAX_ITER", "8", "TRAIN.RESUME", str(ck / "checkpoint_4.pth")])
    st2 = torch.load(ck / "final.pth", map_location="cpu", weights_only=True)
    assert st2["iter"] == 7 and st2["ema_state_dict"]["optimization_step"] == 4
    out = tmp_path / "samples.png"
    train_main.main(["--config", common[1], "--generate-only", "--save-file-name", str(out), "--opts", "PROJECT_DIR",
                     str(proj), "TRAIN.RESUME", str(ck / "final.pth"), "EVAL.SAMPLE_STEPS", "5"])
    from PIL import Image
    assert np.array(Image.open(out)).shape == (16, 64, 3)        # EVAL.BATCH_SIZE = 4 images of 16x16 in one row
