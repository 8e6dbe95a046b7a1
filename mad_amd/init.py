"""Seeded parameter initialiser for the denoiser.

The reference initialiser leaves ``out_conv``, every block's ``conv1`` and every ``proj`` at
zero (``/root/reference/modeling/model.py:307-309, 332-337, 457, 549-551``), which makes a
freshly built denoiser return exactly 0 and any parity check on it vacuous.  This initialiser
gives every tensor a non-zero, reproducible value derived from its ``state_dict`` name, so the
same weights can be rebuilt anywhere from (spec, seed) and never have to be shipped.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict

import torch

from .unet_spec import UNetSpec


def _gen(name: str, seed: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


def seeded_state_dict(spec: UNetSpec, seed: int = 0) -> Dict[str, torch.Tensor]:
    """fp32 CPU tensors keyed like the reference ``state_dict`` (parameters and buffers).

    Scales: weights are uniform with variance 1/(3*fan_in) (the reference's kaiming-uniform with
    ``init_weight=sqrt(1/3)``, ``model.py:452-456``) for every conv/linear, including the ones
    the reference zeroes; biases uniform in +-0.2; GroupNorm gains 1+-0.2, offsets +-0.2;
    ``map_label`` normal with unit variance per input (``model.py:486-493``).
    """
    out: Dict[str, torch.Tensor] = {}
    for name, shape in spec.param_shapes().items():
        g = _gen(name, seed)
        leaf = name.rsplit(".", 2)[-2] if name.count(".") >= 1 else name
        kind = name.rsplit(".", 1)[-1]
        if leaf.startswith("norm") or leaf == "out_norm":
            u = torch.rand(shape, generator=g) * 2 - 1
            t = 1.0 + 0.2 * u if kind == "weight" else 0.2 * u
        elif kind == "bias":
            t = 0.2 * (torch.rand(shape, generator=g) * 2 - 1)
        elif name == "map_label.weight":
            t = torch.randn(shape, generator=g)
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            t = math.sqrt(1.0 / fan_in) * (torch.rand(shape, generator=g) * 2 - 1)
        out[name] = t.to(torch.float32).contiguous()
    for name in spec.buffer_names():
        out[name] = torch.full((1, 1, 2, 2), 0.25, dtype=torch.float32)
    return out
