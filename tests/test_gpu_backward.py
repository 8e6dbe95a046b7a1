"""Gradient kernels of the training step against torch autograd in fp64 (GPU only)."""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def _nhwc(t, dt, dev):
    return t.permute(0, 2, 3, 1).contiguous().to(dt).to(dev)


WGRAD_CASES = [
    # B, H, W, Cin, Cout, ksize, upsample
    (2, 16, 16, 32, 64, 3, False),       # two tiles per image, Cout below the 128-channel slice
    (3, 32, 32, 64, 192, 3, False),      # Cout = 128 + 64: ragged second slice; 3 images
    (2, 16, 32, 96, 128, 1, False),      # 1x1, non-square
    (3, 8, 8, 64, 128, 3, False),        # 8x8 images, odd batch (second image of the last tile absent)
    (4, 8, 8, 128, 64, 1, False),        # 8x8 1x1
    (2, 32, 32, 32, 32, 3, True),        # nearest x2 before the conv
    (1, 24, 48, 32, 40, 3, False),       # H multiple of 8 only, Cout multiple of 8 only
    (2, 12, 12, 24, 16, 3, False),       # generic kernel: shapes the MFMA form does not tile
]


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("case", WGRAD_CASES)
def test_conv_wgrad(dev, case, dt):
    from mad_amd import ops
    B, H, W, Cin, Cout, ks, up = case
    xs = _rand((B, Cin, H // 2, W // 2) if up else (B, Cin, H, W), 1)
    dy = _rand((B, Cout, H, W), 2)
    q = (lambda t: t.to(dt).float()) if dt == torch.bfloat16 else (lambda t: t)
    xin = q(xs).double()
    if up:
        xin = F.interpolate(xin, scale_factor=2, mode="nearest")
    w0 = torch.zeros((Cout, Cin, ks, ks), dtype=torch.float64, requires_grad=True)
    F.conv2d(xin, w0, padding=ks // 2).backward(q(dy).double())
    got = ops.conv_wgrad(_nhwc(xs, dt, dev), _nhwc(dy, dt, dev), ks, upsample=up)
    assert tuple(got.shape) == (Cout, Cin, ks, ks)
    assert rel_err(got.cpu(), w0.grad) < 2e-5
    # accumulate into an existing gradient
    base = torch.full_like(got, 0.5)
    ops.conv_wgrad(_nhwc(xs, dt, dev), _nhwc(dy, dt, dev), ks, upsample=up, out=base, accumulate=True)
    assert rel_err(base.cpu() - 0.5, w0.grad) < 1e-4
    db = ops.bias_grad(_nhwc(dy, dt, dev))
    assert rel_err(db.cpu(), q(dy).double().sum((0, 2, 3))) < 2e-5
