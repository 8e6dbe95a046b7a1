"""Configuration tree for the denoising path.

Mirrors the keys and defaults the reference declares with yacs in
``/root/reference/config.py:9-81`` (``create_cfg``), the ``_BASE_`` inheritance of
``config.py:84-89`` (``merge_possible_with_base``) and the ``--opts KEY VAL`` overrides of
``generate_translation.py:88-89`` (``merge_from_list``).  yacs itself is not a dependency:
only the keys the denoising path reads are kept, in a small attribute-access node.
"""
from __future__ import annotations

import ast
import os.path as osp
from typing import Any, Dict, List

import yaml


class CfgNode(dict):
    """Nested dict with attribute access (``cfg.MODEL.BASE_DIM``)."""

    def __getattr__(self, key: str) -> Any:
        try:
            return self[key]
        except KeyError as e:  # pragma: no cover - mirrors AttributeError semantics
            raise AttributeError(key) from e

    def __setattr__(self, key: str, value: Any) -> None:
        self[key] = value

    def clone(self) -> "CfgNode":
        out = CfgNode()
        for k, v in self.items():
            out[k] = v.clone() if isinstance(v, CfgNode) else (list(v) if isinstance(v, list) else v)
        return out

    # -- merging ---------------------------------------------------------------------
    def merge_from_dict(self, other: Dict[str, Any], _path: str = "") -> None:
        for k, v in other.items():
            full = f"{_path}.{k}" if _path else k
            if k not in self:
                raise KeyError(f"Non-existent config key: {full}")
            if isinstance(v, dict):
                if not isinstance(self[k], CfgNode):
                    raise KeyError(f"Config key {full} is not a section")
                self[k].merge_from_dict(v, full)
            else:
                self[k] = v

    def merge_from_file(self, path: str) -> None:
        with open(path, "r") as f:
            data = yaml.safe_load(f) or {}
        self.merge_from_dict(data)

    def merge_from_list(self, opts: List[str]) -> None:
        """``--opts EVAL.SCHEDULER ddim EVAL.SAMPLE_STEPS 50`` style overrides."""
        if len(opts) % 2 != 0:
            raise ValueError(f"Override list has odd length: {opts}")
        for key, raw in zip(opts[0::2], opts[1::2]):
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                if p not in node:
                    raise KeyError(f"Non-existent config key: {key}")
                node = node[p]
            if parts[-1] not in node:
                raise KeyError(f"Non-existent config key: {key}")
            try:
                val = ast.literal_eval(raw) if isinstance(raw, str) else raw
            except (ValueError, SyntaxError):
                val = raw
            node[parts[-1]] = val


def create_cfg() -> CfgNode:
    """Defaults of the reference's ``create_cfg`` (``config.py:9-81``) for the keys on the path."""
    cfg = CfgNode()
    cfg._BASE_ = None
    cfg.PROJECT_NAME = "Makeup Transfer with Diffusion"
    cfg.PROJECT_DIR = None

    cfg.MODEL = CfgNode()
    cfg.MODEL.IN_CHANNELS = 3
    cfg.MODEL.OUT_CHANNELS = 3
    cfg.MODEL.LAYERS_PER_BLOCK = 2
    cfg.MODEL.BASE_DIM = 128
    cfg.MODEL.LAYER_SCALE = [1, 1, 2, 2, 4, 4]
    cfg.MODEL.PRETRAINED = None
    cfg.MODEL.LABEL_DIM = 0

    cfg.TRAIN = CfgNode()
    cfg.TRAIN.MAKEUP = None
    cfg.TRAIN.RESUME = None
    cfg.TRAIN.IMAGE_SIZE = 256
    cfg.TRAIN.LOG_INTERVAL = 20
    cfg.TRAIN.SAVE_INTERVAL = 10000
    cfg.TRAIN.SAMPLE_INTERVAL = 10000
    cfg.TRAIN.ROOT = None
    cfg.TRAIN.TEXT_LABEL_PATH = None
    cfg.TRAIN.BATCH_SIZE = 2
    cfg.TRAIN.NUM_WORKERS = 4
    cfg.TRAIN.MAX_ITER = 350000
    cfg.TRAIN.GRADIENT_ACCUMULATION_STEPS = 16
    cfg.TRAIN.MIXED_PRECISION = "no"
    cfg.TRAIN.GRAD_NORM = 1.0
    cfg.TRAIN.EMA_MAX_DECAY = 0.9999
    cfg.TRAIN.EMA_INV_GAMMA = 1.0
    cfg.TRAIN.EMA_POWER = 0.75
    cfg.TRAIN.LR = 0.0001
    cfg.TRAIN.LR_WARMUP = 1000
    cfg.TRAIN.TIME_STEPS = 1000
    cfg.TRAIN.SAMPLE_STEPS = 1000
    cfg.TRAIN.NOISE_SCHEDULER = CfgNode()
    cfg.TRAIN.NOISE_SCHEDULER.BETA_START = 1e-4
    cfg.TRAIN.NOISE_SCHEDULER.BETA_END = 0.02
    cfg.TRAIN.NOISE_SCHEDULER.TYPE = "linear"
    cfg.TRAIN.NOISE_SCHEDULER.PRED_TYPE = "epsilon"

    cfg.EVAL = CfgNode()
    cfg.EVAL.BATCH_SIZE = 4
    cfg.EVAL.SAMPLE_STEPS = 1000
    cfg.EVAL.ETA = 0.01
    cfg.EVAL.REFINE_STEPS = 0
    cfg.EVAL.REFINE_ITERATIONS = 0
    cfg.EVAL.SCHEDULER = "ddpm"
    return cfg


def merge_possible_with_base(cfg: CfgNode, config_path: str) -> None:
    """Load ``config_path``; if it names a ``_BASE_`` file merge that first (``config.py:84-89``)."""
    with open(config_path, "r") as f:
        new_cfg = yaml.safe_load(f) or {}
    base = new_cfg.get("_BASE_")
    if base:
        cfg.merge_from_file(osp.join(osp.dirname(config_path), base))
    cfg.merge_from_dict(new_cfg)


def load_cfg(config_path: str = None, opts: List[str] = None) -> CfgNode:
    cfg = create_cfg()
    if config_path:
        merge_possible_with_base(cfg, config_path)
    if opts:
        cfg.merge_from_list(opts)
    return cfg


def show_config(cfg: CfgNode, _indent: int = 0) -> str:
    lines = []
    for k in sorted(cfg.keys()):
        v = cfg[k]
        if isinstance(v, CfgNode):
            lines.append(" " * _indent + f"{k}:")
            lines.append(show_config(v, _indent + 2))
        else:
            lines.append(" " * _indent + f"{k}: {v!r}")
    text = "\n".join(lines)
    if _indent == 0:
        print(text)
    return text
