"""Synthetic inputs for benchmarks and smoke runs: smooth face-like images and face-parsing maps.

No dataset is available offline; the shapes and value ranges are those of the reference's
preprocessed inputs (``/root/reference/modeling/translation.py:171-177, 201-211``: RGB in [-1, 1];
parsing maps with the component ids of ``BASIC_COMPONENT``, translation.py:151).
"""
from __future__ import annotations

import numpy as np
import torch


def synthetic_faces(batch: int, res: int, seed: int = 0) -> torch.Tensor:
    """(B,3,R,R) fp32 in [-0.8, 0.8]: uniform noise low-pass filtered to face-like smoothness."""
    g = torch.Generator().manual_seed(seed)
    low = max(res // 16, 2)
    x = torch.rand((batch, 3, low, low), generator=g) * 2 - 1
    x = torch.nn.functional.interpolate(x, size=(res, res), mode="bicubic", align_corners=False)
    fine = (torch.rand((batch, 3, res, res), generator=g) * 2 - 1) * 0.05
    return ((x + fine).clamp(-1, 1) * 0.8).contiguous()


def synthetic_parsing_map(res: int, seed: int = 0) -> torch.Tensor:
    """(R,R) uint8 label map: face 4, nose 8, eyes 1/6, brows 2/7, lips 9/13, neck 10, hair 12 on 0."""
    rng = np.random.RandomState(seed)
    s = res / 64.0
    jx, jy = rng.uniform(-2, 2, size=2) * s
    yy, xx = np.mgrid[0:res, 0:res].astype(np.float32)

    def ell(cy, cx, ry, rx):
        return ((yy - (cy * s + jy)) / (ry * s)) ** 2 + ((xx - (cx * s + jx)) / (rx * s)) ** 2 <= 1

    lab = np.zeros((res, res), dtype=np.uint8)
    lab[ell(34, 32, 26, 20)] = 4
    lab[ell(58, 32, 8, 12)] = 10
    lab[ell(36, 32, 7, 4)] = 8
    lab[ell(26, 22, 2, 5)] = 1
    lab[ell(26, 42, 2, 5)] = 6
    lab[ell(21, 22, 1.2, 6)] = 2
    lab[ell(21, 42, 1.2, 6)] = 7
    lab[ell(47, 32, 1.5, 7)] = 9
    lab[ell(50, 32, 1.5, 6)] = 13
    lab[ell(6, 32, 6, 22)] = 12
    return torch.from_numpy(lab)
