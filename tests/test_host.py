"""Host-side logic and the C-ABI surface, no GPU needed."""
import os
import re

import numpy as np
import pytest
import torch

from conftest import REPO
from mad_amd.config import create_cfg, load_cfg
from mad_amd.shard import shard_bounds, shard_list
from mad_amd.unet_spec import spec_from_cfg


def test_library_exports_every_declared_symbol():
    import ctypes
    from mad_amd import _lib
    header = open(os.path.join(REPO, "include", "mad_hip.h")).read()
    declared = set(re.findall(r"\b(mad_[a-z0-9_]+)\s*\(", header))
    declared.discard("mad_sampler_coef")
    assert declared, "no declarations found"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    so = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(so, name), name
    assert _lib.lib.mad_version() >= 1
    # struct layout agrees with the header: 7 int32 + 10 float
    assert ctypes.sizeof(_lib.SamplerCoef) == 17 * 4


def test_rejections_happen_before_launch():
    """Bad arguments are refused with a message and nothing is launched (no GPU is touched)."""
    from mad_amd._lib import SamplerCoef, lib
    assert lib.mad_conv2d(None, 64, None, 0, None, None, None, 0, None, 0, 1, 8, 8, 64, 3, 0, None) != 0
    assert b"null pointer" in lib.mad_last_error()
    assert lib.mad_conv2d(1, 60, None, 0, 1, None, None, 0, 1, 0, 1, 8, 8, 64, 3, 0, None) != 0
    assert b"multiples of 64" in lib.mad_last_error()
    assert lib.mad_attention(1, 1, 0, 1, 48, 1, None) != 0
    assert lib.mad_gn_partial(1, 64, None, 0, 0, 1, 1, 100, 3, None) != 0
    assert lib.mad_gn_finalize(1, 64, 1, None, 0, 0, 1, 1, None, 0, 1, 1, 7, 64, 1e-5, None) != 0
    c = SamplerCoef()
    c.op = 7
    assert lib.mad_sampler_update(c, 1, 1, None, None, None, None, 1, None, 1, 3, 4, 4, 1, None) != 0


def test_ops_refuse_cpu_tensors():
    from mad_amd import ops
    with pytest.raises(RuntimeError, match="GPU only"):
        ops.finalize_image(torch.zeros(4))
    with pytest.raises(RuntimeError, match="GPU only"):
        ops.conv2d(torch.zeros(1, 8, 8, 64), torch.zeros(64, 3, 3, 64), None)


def test_config_inheritance_and_overrides():
    cfg = load_cfg(os.path.join(REPO, "configs", "model_64_64.yaml"),
                   ["EVAL.SCHEDULER", "ddim", "EVAL.SAMPLE_STEPS", "50", "MODEL.LAYER_SCALE", "[1,2]"])
    assert cfg.TRAIN.IMAGE_SIZE == 64 and cfg.MODEL.BASE_DIM == 192 and cfg.TRAIN.MIXED_PRECISION == "fp16"
    assert cfg.EVAL.SCHEDULER == "ddim" and cfg.EVAL.SAMPLE_STEPS == 50 and cfg.MODEL.LAYER_SCALE == [1, 2]
    d = create_cfg()
    assert d.EVAL.ETA == 0.01 and d.EVAL.SCHEDULER == "ddpm" and d.MODEL.LAYER_SCALE == [1, 1, 2, 2, 4, 4]
    with pytest.raises(KeyError):
        load_cfg(None, ["MODEL.NOPE", "1"])


def test_spec_matches_survey_numbers():
    s64 = spec_from_cfg(load_cfg(os.path.join(REPO, "configs", "model_64_64.yaml")))
    s256 = spec_from_cfg(load_cfg(os.path.join(REPO, "configs", "model_256_256.yaml")))
    assert s64.num_params() == 295_132_803 and s256.num_params() == 138_287_491
    assert abs(s64.flops_per_image()["total"] / 1e9 - 219.3) < 0.1
    assert abs(s256.flops_per_image()["total"] / 1e9 - 563.2) < 0.1
    assert sum(b.attention for b in s256.blocks()) == 16 and sum(b.attention for b in s64.blocks()) == 22


def test_scheduler_scalars_match_fp32_torch_evaluation():
    """The host-side coefficients equal the reference's 0-dim fp32 tensor arithmetic bit for bit."""
    from mad_amd.scheduler import CustomDDIMScheduler, CustomDDPMScheduler
    from oracle.scheduler_ref import RefScheduler
    p, r = CustomDDPMScheduler(), RefScheduler("ddpm")
    for n in (1000, 50):
        p.set_timesteps(n); r.set_timesteps(n)
        assert p.timesteps.tolist() == r.timesteps.tolist()
        for t in p.timesteps.tolist():
            c = p._coef(t, 0)
            a_t, a_p = r._acp(t), r._acp(r._prev_t(t))
            cur_a = a_t / a_p
            assert np.float32(c.c0) == ((a_p ** 0.5 * (1 - cur_a)) / (1 - a_t)).numpy()
            assert np.float32(c.c1) == (cur_a ** 0.5 * (1 - a_p) / (1 - a_t)).numpy()
            assert np.float32(c.std) == (r.variance(t) ** 0.5).numpy()
            assert np.float32(c.sqrt_b_t) == ((1 - a_t) ** 0.5).numpy()
    d, rd = CustomDDIMScheduler(), RefScheduler("ddim")
    d.set_timesteps(50); rd.set_timesteps(50)
    for t in d.timesteps.tolist():
        c = d._coef(t, 0, 0.01)
        std = 0.01 * rd.variance(t) ** 0.5
        assert np.float32(c.std) == std.numpy()
        assert np.float32(c.dir_coef) == ((1 - rd._acp(rd._prev_t(t)) - std ** 2) ** 0.5).numpy()
    d.set_inverse_timesteps(50)
    assert d.timesteps.tolist() == list(range(0, 1000, 20))
    with pytest.raises(ValueError):
        d.set_timesteps(2000)


def test_shard_split_follows_reference():
    items = list(range(10))
    got = [shard_list(items, 4, r) for r in range(4)]
    assert [g[0] for g in got] == [[0, 1], [2, 3], [4, 5], [6, 7, 8, 9]]
    assert [g[1] for g in got] == [0, 2, 4, 6]
    assert shard_bounds(3, 8, 7) == (0, 3) and shard_bounds(3, 8, 0) == (0, 0)
    with pytest.raises(ValueError):
        shard_bounds(3, 2, 2)


def test_model_construction_and_state_dict_on_cpu():
    """Building the model, seeding it and round-tripping a state_dict needs no GPU."""
    from mad_amd.init import seeded_state_dict
    from mad_amd.model import build_model
    m = build_model(load_cfg(os.path.join(REPO, "configs", "model_tiny.yaml")))
    m.unet.init_seeded(2)
    sd = m.state_dict()
    want = seeded_state_dict(m.spec, 2)
    assert sorted(sd) == sorted(want)
    assert all(torch.equal(sd[k], want[k]) for k in want)
    m.load_state_dict(want)
    with pytest.raises(RuntimeError, match="HIP kernels only"):
        m(torch.zeros(1, 3, 16, 16), torch.tensor([1.0]))


def test_parameter_order_is_the_reference_order():
    """The reference restores EMA weights by position (app.py:15-32): names, shapes and ORDER of
    ``named_parameters()`` must equal the reference module's (fixture recorded from the reference)."""
    import json
    from mad_amd.config import load_cfg
    from mad_amd.model import build_model
    with open(os.path.join(REPO, "tests", "golden", "param_order.json")) as f:
        want = json.load(f)
    for cfg_file, ref in want.items():
        m = build_model(load_cfg(os.path.join(REPO, "configs", cfg_file)))
        mine = [[n, list(p.shape)] for n, p in m.unet.named_parameters()]
        assert mine == ref, cfg_file


def test_reference_format_checkpoint_loads(tmp_path):
    """main.py:278-288 file layout: state_dict by name, EMA shadow_params by position."""
    from mad_amd.checkpoint import load_checkpoint
    from mad_amd.config import load_cfg
    from mad_amd.init import seeded_state_dict
    from mad_amd.model import build_model
    cfg = load_cfg(os.path.join(REPO, "configs", "model_tiny.yaml"))
    m = build_model(cfg)
    raw, ema = seeded_state_dict(m.spec, 5), seeded_state_dict(m.spec, 6)
    names = [n for n, _ in m.unet.named_parameters()]
    path = str(tmp_path / "final.pth")
    torch.save({"state_dict": raw, "optimizer": {}, "lr_scheduler": {}, "iter": 9,
                "ema_state_dict": {"decay": 0.9999, "optimization_step": 10,
                                   "shadow_params": [ema[n].clone() for n in names]}}, path)
    info = load_checkpoint(m, path)                                  # EMA by position (app.py)
    assert info == {"source": "ema_state_dict.shadow_params", "iter": 9}
    got = m.state_dict()
    assert all(torch.equal(got[n], ema[n]) for n in names)
    info = load_checkpoint(m, path, use_ema=False)                   # by name (resume)
    assert info["source"] == "state_dict"
    got = m.state_dict()
    assert all(torch.equal(got[n], raw[n]) for n in names)
    torch.save(raw, path)                                            # a bare state_dict
    m2 = build_model(cfg)
    assert load_checkpoint(m2, path)["source"] == "state_dict"
    torch.save({"ema_state_dict": {"shadow_params": [ema[n] for n in names[:-1]]}}, path)
    with pytest.raises(AssertionError):
        load_checkpoint(m2, path)


def test_library_import_brings_torch_runtime_first():
    """Importing the binding in a fresh interpreter, before anything imported torch, must leave torch's own HIP
    runtime in the process ahead of the library: the other order gives two runtimes and launches that fail."""
    import subprocess
    import sys
    code = ("import sys; import mad_amd._lib as L; assert 'torch' in sys.modules; "
            "maps = open('/proc/self/maps').read(); "
            "libs = [l.split()[-1] for l in maps.splitlines() if 'libamdhip64' in l]; "
            "assert libs and all('/torch/' in p for p in libs), sorted(set(libs)); print('ok')")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr[-2000:]


def test_shipped_configs_carry_the_reference_training_values():
    """Values of the reference's configs/base.yaml, model_256_256.yaml and model_64_64.yaml that the
    trimmed copies here once dropped (same-named config, same schedule)."""
    from mad_amd.config import load_cfg
    for name, batch, accum in (("model_256_256.yaml", 4, 8), ("model_64_64.yaml", 16, 4)):
        cfg = load_cfg(os.path.join(REPO, "configs", name))
        assert cfg.TRAIN.MAX_ITER == 700000
        assert list(cfg.TRAIN.ROOT) == ["data/mtdataset/images"]
        assert cfg.TRAIN.TEXT_LABEL_PATH == "data/mt_text_anno.json"
        assert (cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.GRADIENT_ACCUMULATION_STEPS) == (batch, accum)
        assert cfg.TRAIN.MIXED_PRECISION == "fp16" and cfg.MODEL.LABEL_DIM == 2


def test_process_image_resizes_bilinear():
    """A PIL input that is not at IMAGE_SIZE is resized with the bilinear filter (torchvision's Resize default),
    then scaled to [-1, 1] (reference translation.py:193-211)."""
    import numpy as np
    from PIL import Image
    from mad_amd.translation import TranslationDiffusion
    rng = np.random.default_rng(0)
    img = Image.fromarray(rng.integers(0, 256, (40, 40, 3), dtype=np.uint8))
    td = TranslationDiffusion.__new__(TranslationDiffusion)
    td.image_size, td.device = 16, torch.device("cpu")
    got = td.process_image(img)
    want = np.asarray(img.resize((16, 16), Image.BILINEAR), dtype=np.float32) / 255.0
    want = (torch.from_numpy(want).permute(2, 0, 1) - 0.5) / 0.5
    assert tuple(got.shape) == (1, 3, 16, 16) and torch.equal(got[0], want)
    bicubic = np.asarray(img.resize((16, 16)), dtype=np.float32)
    assert not np.array_equal(bicubic, np.asarray(img.resize((16, 16), Image.BILINEAR), dtype=np.float32))
