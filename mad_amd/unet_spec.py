"""Static description of the ADM-style UNet the denoising path evaluates.

Layer order, channel counts and parameter names follow ``DhariwalUNet.__init__``
(``/root/reference/modeling/model.py:427-551``) and ``UNetBlock.__init__`` (``model.py:253-337``),
so a reference ``state_dict`` maps onto this spec key by key:
``map_layer0/1.{weight,bias}``, ``map_label.weight``, ``enc.{R}x{R}_conv.*``,
``enc|dec.{R}x{R}_{block{i},down,up,in0,in1}.{norm0,conv0,affine,norm1,conv1,skip,norm2,qkv,proj}.*``,
``out_norm.*``, ``out_conv.*`` and the ``resample_filter`` buffers of up/down convolutions.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

ATTN_RESOLUTIONS = (32, 16, 8)      # model.py:444
CHANNELS_PER_HEAD = 64              # model.py:460
GN_MAX_GROUPS = 32                  # model.py:175
GN_MIN_CHANNELS_PER_GROUP = 4       # model.py:175
GN_EPS = 1e-5                       # model.py:175, 266


def gn_groups(channels: int) -> int:
    """Group count of the reference GroupNorm (``model.py:177``)."""
    return min(GN_MAX_GROUPS, channels // GN_MIN_CHANNELS_PER_GROUP)


@dataclass
class BlockSpec:
    """One ``UNetBlock`` (``model.py:253-377``)."""
    name: str                 # e.g. "enc.64x64_block0"
    in_channels: int
    out_channels: int
    res_in: int               # spatial size of the block input
    res_out: int              # spatial size of the block output
    up: bool = False
    down: bool = False
    attention: bool = False
    skip_conv: bool = False   # 1x1 projection on the skip path (channel change)
    cat_skip: int = 0         # decoder only: channels of the encoder skip concatenated first

    @property
    def num_heads(self) -> int:
        return self.out_channels // CHANNELS_PER_HEAD if self.attention else 0


@dataclass
class UNetSpec:
    img_resolution: int
    in_channels: int
    out_channels: int
    label_dim: int
    model_channels: int
    channel_mult: Tuple[int, ...]
    num_blocks: int
    emb_channels: int
    enc_conv_name: str = ""
    enc_conv_out: int = 0
    enc: List[BlockSpec] = field(default_factory=list)     # UNetBlocks after the first conv
    dec: List[BlockSpec] = field(default_factory=list)
    final_channels: int = 0

    def blocks(self) -> List[BlockSpec]:
        return self.enc + self.dec

    # -- parameter table ---------------------------------------------------------------
    def param_shapes(self) -> Dict[str, Tuple[int, ...]]:
        """``state_dict`` key -> shape, in the reference's registration order."""
        p: Dict[str, Tuple[int, ...]] = {}
        C, E = self.model_channels, self.emb_channels
        p["map_layer0.weight"] = (E, C)
        p["map_layer0.bias"] = (E,)
        p["map_layer1.weight"] = (E, E)
        p["map_layer1.bias"] = (E,)
        if self.label_dim:
            p["map_label.weight"] = (E, self.label_dim)
        p[f"{self.enc_conv_name}.weight"] = (self.enc_conv_out, self.in_channels, 3, 3)
        p[f"{self.enc_conv_name}.bias"] = (self.enc_conv_out,)
        for b in self.blocks():
            n, ci, co = b.name, b.in_channels, b.out_channels
            p[f"{n}.norm0.weight"] = (ci,)
            p[f"{n}.norm0.bias"] = (ci,)
            p[f"{n}.conv0.weight"] = (co, ci, 3, 3)
            p[f"{n}.conv0.bias"] = (co,)
            p[f"{n}.affine.weight"] = (2 * co, E)
            p[f"{n}.affine.bias"] = (2 * co,)
            p[f"{n}.norm1.weight"] = (co,)
            p[f"{n}.norm1.bias"] = (co,)
            p[f"{n}.conv1.weight"] = (co, co, 3, 3)
            p[f"{n}.conv1.bias"] = (co,)
            if b.skip_conv:
                p[f"{n}.skip.weight"] = (co, ci, 1, 1)
                p[f"{n}.skip.bias"] = (co,)
            if b.attention:
                p[f"{n}.norm2.weight"] = (co,)
                p[f"{n}.norm2.bias"] = (co,)
                p[f"{n}.qkv.weight"] = (3 * co, co, 1, 1)
                p[f"{n}.qkv.bias"] = (3 * co,)
                p[f"{n}.proj.weight"] = (co, co, 1, 1)
                p[f"{n}.proj.bias"] = (co,)
        p["out_norm.weight"] = (self.final_channels,)
        p["out_norm.bias"] = (self.final_channels,)
        p["out_conv.weight"] = (self.out_channels, self.final_channels, 3, 3)
        p["out_conv.bias"] = (self.out_channels,)
        return p

    def buffer_names(self) -> List[str]:
        """``resample_filter`` buffers the reference registers on up/down convs (``model.py:112-114``)."""
        out = []
        for b in self.blocks():
            if b.up or b.down:
                out.append(f"{b.name}.conv0.resample_filter")
                out.append(f"{b.name}.skip.resample_filter")
        return out

    def num_params(self) -> int:
        n = 0
        for shp in self.param_shapes().values():
            k = 1
            for s in shp:
                k *= s
            n += k
        return n

    # -- algorithmic work --------------------------------------------------------------
    def flops_per_image(self) -> Dict[str, float]:
        """FLOPs of one evaluation for one image: 2*H*W*Cout*Cin*k*k per conv (3x3 and 1x1),
        4*L*L*C per attention, 2*in*out per linear layer.  Resampling, GroupNorm, SiLU and
        other elementwise work are excluded."""
        conv = attn = lin = 0.0
        R = self.img_resolution
        C, E = self.model_channels, self.emb_channels
        conv += 2.0 * R * R * self.enc_conv_out * self.in_channels * 9
        lin += 2.0 * (C * E + E * E + self.label_dim * E)
        for b in self.blocks():
            hw = b.res_out * b.res_out
            conv += 2.0 * hw * b.out_channels * b.in_channels * 9
            conv += 2.0 * hw * b.out_channels * b.out_channels * 9
            lin += 2.0 * E * 2 * b.out_channels
            if b.skip_conv:
                conv += 2.0 * hw * b.out_channels * b.in_channels
            if b.attention:
                conv += 2.0 * hw * 3 * b.out_channels * b.out_channels
                conv += 2.0 * hw * b.out_channels * b.out_channels
                attn += 4.0 * hw * hw * b.out_channels
        conv += 2.0 * R * R * self.out_channels * self.final_channels * 9
        return {"conv": conv, "attn": attn, "linear": lin, "total": conv + attn + lin}


def build_spec(img_resolution: int, in_channels: int, out_channels: int, label_dim: int,
               model_channels: int, channel_mult, num_blocks: int,
               channel_mult_emb: int = 4) -> UNetSpec:
    channel_mult = tuple(int(m) for m in channel_mult)
    spec = UNetSpec(img_resolution=img_resolution, in_channels=in_channels,
                    out_channels=out_channels, label_dim=label_dim or 0,
                    model_channels=model_channels, channel_mult=channel_mult,
                    num_blocks=num_blocks, emb_channels=model_channels * channel_mult_emb)
    skips: List[int] = []
    cout = in_channels
    for level, mult in enumerate(channel_mult):
        res = img_resolution >> level
        if level == 0:
            cout = model_channels * mult
            spec.enc_conv_name = f"enc.{res}x{res}_conv"
            spec.enc_conv_out = cout
            skips.append(cout)
        else:
            spec.enc.append(BlockSpec(f"enc.{res}x{res}_down", cout, cout, res * 2, res, down=True))
            skips.append(cout)
        for idx in range(num_blocks):
            cin, cout = cout, model_channels * mult
            spec.enc.append(BlockSpec(f"enc.{res}x{res}_block{idx}", cin, cout, res, res,
                                      attention=res in ATTN_RESOLUTIONS, skip_conv=cin != cout))
            skips.append(cout)
    for level, mult in reversed(list(enumerate(channel_mult))):
        res = img_resolution >> level
        if level == len(channel_mult) - 1:
            spec.dec.append(BlockSpec(f"dec.{res}x{res}_in0", cout, cout, res, res, attention=True))
            spec.dec.append(BlockSpec(f"dec.{res}x{res}_in1", cout, cout, res, res))
        else:
            spec.dec.append(BlockSpec(f"dec.{res}x{res}_up", cout, cout, res // 2, res, up=True))
        for idx in range(num_blocks + 1):
            sk = skips.pop()
            cin, cout = cout + sk, model_channels * mult
            spec.dec.append(BlockSpec(f"dec.{res}x{res}_block{idx}", cin, cout, res, res,
                                      attention=res in ATTN_RESOLUTIONS, skip_conv=cin != cout,
                                      cat_skip=sk))
    assert not skips
    spec.final_channels = cout
    return spec


def spec_from_cfg(cfg) -> UNetSpec:
    """Same seven keys ``build_model`` reads (``model.py:654-668``)."""
    return build_spec(img_resolution=cfg.TRAIN.IMAGE_SIZE, in_channels=cfg.MODEL.IN_CHANNELS,
                      out_channels=cfg.MODEL.OUT_CHANNELS, label_dim=cfg.MODEL.LABEL_DIM,
                      model_channels=cfg.MODEL.BASE_DIM, channel_mult=cfg.MODEL.LAYER_SCALE,
                      num_blocks=cfg.MODEL.LAYERS_PER_BLOCK)
