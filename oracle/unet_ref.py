"""TEST INFRASTRUCTURE -- CPU oracle for the denoiser.  Never imported by the product path.

A plain-PyTorch fp32 restatement of ``DhariwalUNet.forward``
(``/root/reference/modeling/model.py:553-590``) and the layers it calls, written functionally
over a ``state_dict`` so that it shares no code with ``mad_amd``.  It is pinned against golden
vectors produced by the reference's own classes (``tests/golden/make_golden.py``); the
reference carries no tests or known-answer vectors of its own for this path.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def positional_embedding(t: Tensor, num_channels: int, max_positions: int = 10000) -> Tensor:
    """``PositionalEmbedding.forward`` (model.py:392-400): cos half first, then sin."""
    half = num_channels // 2
    freqs = torch.arange(half, dtype=torch.float32, device=t.device) / half
    freqs = (1.0 / max_positions) ** freqs
    ang = torch.outer(t.to(torch.float32).reshape(-1), freqs)
    return torch.cat([ang.cos(), ang.sin()], dim=1)


def group_norm(x: Tensor, w: Tensor, b: Tensor, eps: float = 1e-5) -> Tensor:
    """``GroupNorm.forward`` (model.py:182-190); group count from model.py:177."""
    groups = min(32, x.shape[1] // 4)
    return F.group_norm(x, groups, w, b, eps)


def down2(x: Tensor) -> Tensor:
    """Depthwise [1,1] box filter, stride 2 (model.py:112-114, 155-161) == 2x2 mean."""
    return F.avg_pool2d(x, 2)


def up2(x: Tensor) -> Tensor:
    """Depthwise transposed [1,1] filter times 4, stride 2 (model.py:147-153) == nearest x2."""
    return x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)


def conv(x: Tensor, w: Optional[Tensor], b: Optional[Tensor], up: bool = False,
         down: bool = False) -> Tensor:
    """``Conv2d.forward`` non-fused branch (model.py:145-166)."""
    if up:
        x = up2(x)
    if down:
        x = down2(x)
    if w is not None:
        x = F.conv2d(x, w, padding=w.shape[-1] // 2)
    if b is not None:
        x = x + b.reshape(1, -1, 1, 1)
    return x


def attention(qkv: Tensor, num_heads: int) -> Tensor:
    """model.py:357-375 with ``AttentionOp`` (model.py:199-212): channel = head*3*64 + d*3 + {q,k,v}."""
    B, C3, H, W = qkv.shape
    C = C3 // 3
    q, k, v = qkv.reshape(B * num_heads, C // num_heads, 3, H * W).unbind(2)
    w = torch.einsum("ncq,nck->nqk", q, k / math.sqrt(k.shape[1])).softmax(dim=2)
    a = torch.einsum("nqk,nck->ncq", w, v)
    return a.reshape(B, C, H, W)


def unet_block(sd: Dict[str, Tensor], name: str, x: Tensor, emb: Tensor, up: bool, down: bool,
               num_heads: int) -> Tensor:
    """``UNetBlock.forward`` (model.py:339-377) in eval mode (dropout is the identity)."""
    g = lambda k: sd.get(f"{name}.{k}")
    orig = x
    x = conv(F.silu(group_norm(x, g("norm0.weight"), g("norm0.bias"))),
             g("conv0.weight"), g("conv0.bias"), up=up, down=down)
    params = (emb @ g("affine.weight").t() + g("affine.bias"))[:, :, None, None]
    scale, shift = params.chunk(2, dim=1)          # scale is the FIRST half (model.py:345)
    x = F.silu(shift + group_norm(x, g("norm1.weight"), g("norm1.bias")) * (scale + 1))
    x = conv(x, g("conv1.weight"), g("conv1.bias"))
    if g("skip.weight") is not None or up or down:
        x = x + conv(orig, g("skip.weight"), g("skip.bias"), up=up, down=down)
    else:
        x = x + orig
    if num_heads:
        qkv = conv(group_norm(x, g("norm2.weight"), g("norm2.bias")), g("qkv.weight"), g("qkv.bias"))
        a = attention(qkv, num_heads)
        x = conv(a, g("proj.weight"), g("proj.bias")) + x
    return x


def layer_plan(img_resolution: int, model_channels: int, channel_mult, num_blocks: int):
    """Encoder/decoder module order of ``DhariwalUNet.__init__`` (model.py:498-547) as
    (name, kind, up, down, attention) tuples; attention at resolutions 32/16/8 (model.py:444)."""
    enc, dec = [], []
    attn_res = (32, 16, 8)
    for level, _ in enumerate(channel_mult):
        res = img_resolution >> level
        if level == 0:
            enc.append((f"enc.{res}x{res}_conv", "conv", False, False, False))
        else:
            enc.append((f"enc.{res}x{res}_down", "block", False, True, False))
        for idx in range(num_blocks):
            enc.append((f"enc.{res}x{res}_block{idx}", "block", False, False, res in attn_res))
    for level, _ in reversed(list(enumerate(channel_mult))):
        res = img_resolution >> level
        if level == len(channel_mult) - 1:
            dec.append((f"dec.{res}x{res}_in0", "block", False, False, True))
            dec.append((f"dec.{res}x{res}_in1", "block", False, False, False))
        else:
            dec.append((f"dec.{res}x{res}_up", "block", True, False, False))
        for idx in range(num_blocks + 1):
            dec.append((f"dec.{res}x{res}_block{idx}", "block", False, False, res in attn_res))
    return enc, dec


def unet_forward(sd: Dict[str, Tensor], x: Tensor, noise_labels: Tensor,
                 class_labels: Optional[Tensor], *, img_resolution: int, model_channels: int,
                 channel_mult, num_blocks: int, label_dim: int,
                 taps: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """``DhariwalUNet.forward`` (model.py:553-590).  Returns the tensor (the reference wraps it
    in a 1-tuple).  ``taps`` (if given) receives every module's output by name."""
    x = x.to(torch.float32)
    emb = positional_embedding(noise_labels, model_channels)
    emb = F.silu(emb @ sd["map_layer0.weight"].t() + sd["map_layer0.bias"])
    emb = emb @ sd["map_layer1.weight"].t() + sd["map_layer1.bias"]
    if label_dim:
        if class_labels is None:
            class_labels = torch.zeros(1, label_dim)
        emb = emb + class_labels.to(torch.float32).reshape(-1, label_dim) @ sd["map_label.weight"].t()
    emb = F.silu(emb)
    if taps is not None:
        taps["emb"] = emb

    enc, dec = layer_plan(img_resolution, model_channels, channel_mult, num_blocks)
    skips: List[Tensor] = []
    for name, kind, up, down, attn in enc:
        if kind == "conv":
            x = conv(x, sd[f"{name}.weight"], sd[f"{name}.bias"])
        else:
            cout = sd[f"{name}.conv1.weight"].shape[0]
            x = unet_block(sd, name, x, emb, up, down, cout // 64 if attn else 0)
        skips.append(x)
        if taps is not None:
            taps[name] = x
    for name, kind, up, down, attn in dec:
        if x.shape[1] != sd[f"{name}.conv0.weight"].shape[1]:
            x = torch.cat([x, skips.pop()], dim=1)
        cout = sd[f"{name}.conv1.weight"].shape[0]
        x = unet_block(sd, name, x, emb, up, down, cout // 64 if attn else 0)
        if taps is not None:
            taps[name] = x
    x = conv(F.silu(group_norm(x, sd["out_norm.weight"], sd["out_norm.bias"])),
             sd["out_conv.weight"], sd["out_conv.bias"])
    return x
