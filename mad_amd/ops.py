"""Operator-level Python wrappers over the C ABI (one function per entry point).

Tensors are torch CUDA tensors used only as device memory; every wrapper checks shapes, dtype,
contiguity and device before handing raw pointers to the library, enqueues on torch's current
stream and returns the output tensor.  Activations are NHWC: shape (B, H, W, C).
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from . import _lib
from ._lib import SamplerCoef, check, lib

Tensor = torch.Tensor


def _dtype_code(t: Tensor) -> int:
    if t.dtype == torch.bfloat16:
        return _lib.MAD_BF16
    if t.dtype == torch.float32:
        return _lib.MAD_F32
    raise TypeError(f"unsupported activation dtype {t.dtype}")


def _dev(*ts: Optional[Tensor]) -> None:
    dev = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("mad_amd operators run on the GPU only (got a CPU tensor); there is no CPU path")
        if not t.is_contiguous():
            raise ValueError("mad_amd operators need contiguous tensors")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise ValueError(f"tensors on different devices: {t.device} vs {dev}")


def _ptr(t: Optional[Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class KernelTimer:
    """Optional per-launch timing of the MFMA kernels with events recorded on the launch stream.

    While ``active`` is set, every ``conv2d`` / ``attention`` launch is bracketed by two events on
    torch's current stream (the stream the kernel is enqueued on) and logged with its algorithmic
    FLOPs.  ``summary()`` synchronises and returns totals per kernel.  Off by default: the timed
    loops enable it for a sampled subset of denoiser evaluations.
    """

    def __init__(self):
        self.active = False
        self.records = []

    def bracket(self, name: str, flops: float):
        if not self.active:
            return None
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        self.records.append((name, flops, a, b))
        return b

    def summary(self) -> dict:
        torch.cuda.synchronize()
        out: dict = {}
        for name, flops, a, b in self.records:
            d = out.setdefault(name, {"launches": 0, "flops": 0.0, "ms": 0.0})
            d["launches"] += 1
            d["flops"] += flops
            d["ms"] += a.elapsed_time(b)
        self.records = []
        return out


TIMER = KernelTimer()


# ------------------------------------------------------------------------------ sampler
def sampler_update(coef: SamplerCoef, model_output: Tensor, sample: Tensor,
                   noise: Optional[Tensor] = None, original: Optional[Tensor] = None,
                   mask: Optional[Tensor] = None, next_sample: Optional[Tensor] = None,
                   want_x0: bool = False, out: Optional[Tensor] = None):
    """One pass of the sampler kernel.  ``out`` (fp32, the sample's shape, contiguous) receives the result in
    place of a fresh tensor -- the noise-encoding loop passes a slice of its step-major slab."""
    _dev(model_output, sample, noise, original, mask, next_sample, out)
    for t in (model_output, sample, noise, original, next_sample):
        if t is not None and (t.dtype != torch.float32 or t.shape != sample.shape):
            raise ValueError("sampler tensors must be fp32 and share the sample's shape")
    B, C, H, W = sample.shape
    mask_batch = 1
    if mask is not None:
        if mask.dtype != torch.float32 or mask.numel() not in (H * W, B * H * W):
            raise ValueError(f"mask must be fp32 with H*W or B*H*W elements, got {tuple(mask.shape)}")
        mask_batch = mask.numel() // (H * W)
    if out is None:
        out = torch.empty_like(sample)
    elif out.dtype != torch.float32 or out.shape != sample.shape:
        raise ValueError("sampler_update: out must be fp32 with the sample's shape")
    x0 = torch.empty_like(sample) if want_x0 else None
    check(lib.mad_sampler_update(coef, _ptr(model_output), _ptr(sample), _ptr(noise), _ptr(original),
                                 _ptr(mask), _ptr(next_sample), _ptr(out), _ptr(x0), B, C, H, W,
                                 mask_batch, _stream()), "mad_sampler_update")
    return (out, x0) if want_x0 else out


def add_noise(x0: Tensor, noise: Tensor, sa: float, sb: float, mask: Optional[Tensor] = None) -> Tensor:
    _dev(x0, noise, mask)
    if x0.dtype != torch.float32 or noise.dtype != torch.float32 or noise.shape != x0.shape:
        raise ValueError("add_noise needs fp32 tensors of equal shape")
    B, C, H, W = x0.shape
    mask_batch = 1
    if mask is not None:
        if mask.dtype != torch.float32 or mask.numel() not in (H * W, B * H * W):
            raise ValueError("mask must be fp32 with H*W or B*H*W elements")
        mask_batch = mask.numel() // (H * W)
    out = torch.empty_like(x0)
    check(lib.mad_add_noise(_ptr(x0), _ptr(noise), _ptr(mask), _ptr(out), float(sa), float(sb), B, C, H, W,
                            mask_batch, _stream()), "mad_add_noise")
    return out


def finalize_image(x: Tensor, normalize: bool = True) -> Tensor:
    _dev(x)
    if x.dtype != torch.float32:
        raise ValueError("finalize_image needs fp32")
    out = torch.empty_like(x)
    check(lib.mad_finalize_image(_ptr(x), _ptr(out), int(normalize), x.numel(), _stream()), "mad_finalize_image")
    return out


def masked_copy_(dst: Tensor, src: Tensor, mask: Tensor) -> Tensor:
    _dev(dst, src, mask)
    B, C, H, W = dst.shape
    if src.shape != dst.shape or dst.dtype != torch.float32 or src.dtype != torch.float32:
        raise ValueError("masked_copy_ needs fp32 tensors of equal shape")
    if mask.dtype != torch.float32 or mask.numel() not in (H * W, B * H * W):
        raise ValueError("mask must be fp32 with H*W or B*H*W elements")
    check(lib.mad_masked_copy(_ptr(dst), _ptr(src), _ptr(mask), B, C, H, W, mask.numel() // (H * W), _stream()),
          "mad_masked_copy")
    return dst


def keep_mask(label_map: Tensor, comps: Sequence[int], dilate: bool = True) -> Tensor:
    """label_map (H,W) uint8 on the GPU -> (1,H,W) fp32 keep-mask, or a stack (B,H,W) -> (B,1,H,W) in one launch;
    comps already priority-sorted."""
    _dev(label_map)
    if label_map.dtype != torch.uint8 or label_map.dim() not in (2, 3):
        raise ValueError("label_map must be a (H,W) or (B,H,W) uint8 tensor")
    label_map = label_map.contiguous()
    H, W = label_map.shape[-2:]
    comps_t = torch.tensor(list(comps), dtype=torch.int32, device=label_map.device)
    if label_map.dim() == 2:
        out = torch.empty((1, H, W), dtype=torch.float32, device=label_map.device)
        check(lib.mad_keep_mask(_ptr(label_map), _ptr(out), H, W, _ptr(comps_t), len(comps), int(dilate), _stream()),
              "mad_keep_mask")
        return out
    B = label_map.shape[0]
    out = torch.empty((B, 1, H, W), dtype=torch.float32, device=label_map.device)
    check(lib.mad_keep_mask_batch(_ptr(label_map), _ptr(out), B, H, W, _ptr(comps_t), len(comps), int(dilate),
                                  _stream()), "mad_keep_mask_batch")
    return out


# ------------------------------------------------------------------------------ embedding
def pos_embedding(t: Tensor, freqs: Tensor, rows: int) -> Tensor:
    _dev(t, freqs)
    if t.dtype != torch.float32 or freqs.dtype != torch.float32:
        raise ValueError("pos_embedding needs fp32 timesteps (integer timesteps truncate the frequencies)")
    half = freqs.numel()
    out = torch.empty((rows, 2 * half), dtype=torch.float32, device=t.device)
    check(lib.mad_pos_embedding(_ptr(t), t.numel(), _ptr(freqs), _ptr(out), rows, half, _stream()),
          "mad_pos_embedding")
    return out


def linear_f32(x: Tensor, W: Tensor, bias: Optional[Tensor] = None, act_in: bool = False,
               act_out: bool = False, out: Optional[Tensor] = None, accumulate: bool = False) -> Tensor:
    _dev(x, W, bias, out)
    rows, K = x.shape
    N = W.shape[0]
    if W.shape[1] != K or x.dtype != torch.float32 or W.dtype != torch.float32:
        raise ValueError(f"linear_f32 shape/dtype mismatch: x {tuple(x.shape)} W {tuple(W.shape)}")
    if out is None:
        if accumulate:
            raise ValueError("accumulate needs an output tensor")
        out = torch.empty((rows, N), dtype=torch.float32, device=x.device)
    check(lib.mad_linear_f32(_ptr(x), _ptr(W), _ptr(bias), _ptr(out), rows, K, N, int(act_in), int(act_out),
                             int(accumulate), _stream()), "mad_linear_f32")
    return out


# ------------------------------------------------------------------------------ convolutions
def conv_in(x: Tensor, w: Tensor, bias: Tensor, dtype: torch.dtype, want_stats: bool = False):
    """x NCHW fp32 -> NHWC `dtype`; w (Cout,Cin,3,3) fp32.  With ``want_stats`` returns ``(y, part)``: the
    output's GroupNorm partial sums (B, (H/8)*(W/16), Cout, 2) from the epilogue, or None for a shape the tiled
    kernel does not cover."""
    _dev(x, w, bias)
    B, Cin, H, W_ = x.shape
    Cout = w.shape[0]
    if x.dtype != torch.float32 or w.dtype != torch.float32 or tuple(w.shape) != (Cout, Cin, 3, 3):
        raise ValueError("conv_in needs fp32 NCHW input and (Cout,Cin,3,3) fp32 weights")
    y = torch.empty((B, H, W_, Cout), dtype=dtype, device=x.device)
    if want_stats:
        if not lib.mad_conv_in_stats_ok(_dtype_code(y), H, W_, Cout):
            check(lib.mad_conv_in(_ptr(x), _ptr(w), _ptr(bias), _ptr(y), _dtype_code(y), B, Cin, H, W_, Cout,
                                  _stream()), "mad_conv_in")
            return y, None
        part = torch.empty((B, (H // 8) * (W_ // 16), Cout, 2), dtype=torch.float32, device=x.device)
        check(lib.mad_conv_in_stats(_ptr(x), _ptr(w), _ptr(bias), _ptr(y), _ptr(part), _dtype_code(y), B, Cin, H, W_,
                                    Cout, _stream()), "mad_conv_in_stats")
        return y, part
    check(lib.mad_conv_in(_ptr(x), _ptr(w), _ptr(bias), _ptr(y), _dtype_code(y), B, Cin, H, W_, Cout, _stream()),
          "mad_conv_in")
    return y


def conv_out(x: Tensor, w: Tensor, bias: Tensor) -> Tensor:
    """x NHWC -> NCHW fp32; w (Cout,3,3,Cin) fp32."""
    _dev(x, w, bias)
    B, H, W_, Cin = x.shape
    Cout = w.shape[0]
    if w.dtype != torch.float32 or tuple(w.shape) != (Cout, 3, 3, Cin):
        raise ValueError("conv_out needs (Cout,3,3,Cin) fp32 weights")
    y = torch.empty((B, Cout, H, W_), dtype=torch.float32, device=x.device)
    check(lib.mad_conv_out(_ptr(x), _dtype_code(x), _ptr(w), _ptr(bias), _ptr(y), B, Cin, H, W_, Cout, _stream()),
          "mad_conv_out")
    return y


def conv_out_fused(x: Tensor, w32: Tensor, bias: Tensor, cout: int, coef: Optional[Tensor] = None,
                   act: bool = True) -> Tensor:
    """Last layer on the MFMA halo kernel: x NHWC, w32 (32,3,3,Cin) zero-padded rows, -> NCHW fp32."""
    _dev(x, w32, bias, coef)
    B, H, W_, Cin = x.shape
    if w32.dtype != x.dtype or tuple(w32.shape) != (32, 3, 3, Cin) or bias.dtype != torch.float32:
        raise ValueError("conv_out_fused needs (32,3,3,Cin) weights in the activation dtype and fp32 bias")
    if coef is not None and (coef.dtype != torch.float32 or tuple(coef.shape) != (B, Cin, 2)):
        raise ValueError("conv_out_fused coef must be fp32 (B, Cin, 2)")
    y = torch.empty((B, cout, H, W_), dtype=torch.float32, device=x.device)
    end = TIMER.bracket("conv_out", 2.0 * B * H * W_ * cout * Cin * 9)
    check(lib.mad_conv_out_fused(_ptr(x), Cin, _ptr(coef), int(act), _ptr(w32), _ptr(bias), _ptr(y), _dtype_code(x),
                                 B, H, W_, cout, _stream()), "mad_conv_out_fused")
    if end is not None:
        end.record()
    return y


def conv2d(x1: Tensor, w: Tensor, bias: Optional[Tensor], x2: Optional[Tensor] = None,
           residual: Optional[Tensor] = None, res_mode: int = _lib.RES_NONE, upsample: bool = False,
           want_stats: bool = False):
    """Implicit-GEMM conv; x NHWC, w (Cout,k,k,C1+C2) in the activation dtype, bias fp32.  With
    ``want_stats`` (H*W a multiple of 64) also returns the output's GroupNorm partial sums
    (B, H*W/64, Cout, 2) written by the epilogue."""
    _dev(x1, x2, w, bias, residual)
    B, Hs, Ws, C1 = x1.shape
    C2 = 0 if x2 is None else x2.shape[3]
    H, W_ = (2 * Hs, 2 * Ws) if upsample else (Hs, Ws)
    Cout, k = w.shape[0], w.shape[1]
    if w.dtype != x1.dtype or tuple(w.shape) != (Cout, k, k, C1 + C2):
        raise ValueError(f"conv2d weight {tuple(w.shape)}/{w.dtype} does not match input channels {C1}+{C2}/{x1.dtype}")
    if x2 is not None and (x2.dtype != x1.dtype or tuple(x2.shape[:3]) != (B, Hs, Ws)):
        raise ValueError("conv2d second input must match the first in dtype and spatial shape")
    if bias is not None and (bias.dtype != torch.float32 or bias.numel() != Cout):
        raise ValueError("conv2d bias must be fp32 of length Cout")
    if residual is not None:
        want = {_lib.RES_SAME: (B, H, W_, Cout), _lib.RES_UP: (B, H // 2, W_ // 2, Cout),
                _lib.RES_DOWN: (B, 2 * H, 2 * W_, Cout)}.get(res_mode)
        if want is None or tuple(residual.shape) != want or residual.dtype != x1.dtype:
            raise ValueError(f"conv2d residual {tuple(residual.shape)} does not fit mode {res_mode} (want {want})")
    elif res_mode != _lib.RES_NONE:
        raise ValueError("res_mode set without a residual tensor")
    y = torch.empty((B, H, W_, Cout), dtype=x1.dtype, device=x1.device)
    end = TIMER.bracket("conv_igemm", 2.0 * B * H * W_ * Cout * (C1 + C2) * k * k)
    stats = None
    if want_stats:
        if (H * W_) % 64:
            raise ValueError(f"conv2d statistics need H*W ({H * W_}) to be a multiple of 64")
        stats = torch.empty((B, H * W_ // 64, Cout, 2), dtype=torch.float32, device=x1.device)
    ws_bytes = lib.mad_conv2d_workspace_bytes(_dtype_code(x1), B, H, W_, C1 + C2, Cout, k)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x1.device) if ws_bytes else None
    check(lib.mad_conv2d_ws(_ptr(x1), C1, _ptr(x2), C2, _ptr(w), _ptr(bias), _ptr(residual), res_mode, _ptr(y),
                            _ptr(stats), _dtype_code(x1), B, H, W_, Cout, k, int(upsample), _ptr(ws), ws_bytes,
                            _stream()), "mad_conv2d_ws")
    if end is not None:
        end.record()
    return (y, stats) if want_stats else y


def halo_tile_weights(w: Tensor, dtype: torch.dtype, scale: float = 1.0) -> Tensor:
    """(Cout, 3, 3, Cin) weights -> the tiled layout of ``mad_conv3x3_fused`` (MAD_W_TILED), in ``dtype``, multiplied
    by ``scale`` first (``_lib.PRESCALE_W`` for the prescaled form).  Result (Cout/16, Cin/CK, 9, 16, 4, VEC): one KiB
    per (16-row block, 64-byte channel chunk, tap), rows of four 16-byte pieces in the order the kernel keeps them in
    LDS (piece p of row r holds logical piece p ^ (-(r >> 2) & 3))."""
    Cout, kh, kw, Cin = w.shape
    vec = 16 // torch.empty((), dtype=dtype).element_size()
    ck = 4 * vec
    if (kh, kw) != (3, 3) or Cout % 16 or Cin % ck:
        raise ValueError(f"halo_tile_weights: shape {tuple(w.shape)} does not tile (Cout % 16, Cin % {ck})")
    wq = (w.float() * scale).to(dtype) if scale != 1.0 else w.to(dtype)
    w6 = wq.reshape(Cout // 16, 16, 9, Cin // ck, 4, vec).permute(0, 3, 2, 1, 4, 5)     # (blk, chunk, tap, row, piece, v)
    src = torch.tensor([[p ^ ((-(r >> 2)) & 3) for p in range(4)] for r in range(16)], device=w.device)
    idx = src.view(1, 1, 1, 16, 4, 1).expand(Cout // 16, Cin // ck, 9, 16, 4, vec)
    return torch.gather(w6, 4, idx).contiguous()


def conv3x3_fused(x1: Tensor, w: Tensor, bias: Optional[Tensor], coef: Optional[Tensor] = None, act: bool = True,
                  x2: Optional[Tensor] = None, residual: Optional[Tensor] = None, res_mode: int = _lib.RES_NONE,
                  upsample: bool = False, want_stats: bool = False, skip: Optional[tuple] = None,
                  prescaled: bool = False, w_tiled: bool = False):
    """Halo-tile 3x3 conv with the input's GroupNorm(+SiLU) applied in the loader (coef (B,C,2) or None)
    and, optionally, per-tile statistics of the output (B, tiles, Cout, 2) for the next GroupNorm.
    ``skip=(s1, s2_or_None, sw)`` adds the 1x1 projection ``sw (Cout, SC1+SC2)`` of the raw concatenation
    [s1, s2] in the same launch (``bias`` must then already hold the sum of both layers' biases).
    ``prescaled``: coef comes from ``gn_finalize(..., scale=_lib.PRESCALE_COEF)`` and w carries ``_lib.PRESCALE_W``;
    ``w_tiled``: w is the result of ``halo_tile_weights``."""
    _dev(x1, x2, w, bias, coef, residual)
    B, Hs, Ws, C1 = x1.shape
    C2 = 0 if x2 is None else x2.shape[3]
    H, W_ = (2 * Hs, 2 * Ws) if upsample else (Hs, Ws)
    if w_tiled:
        vec = 16 // x1.element_size()
        Cout = w.shape[0] * 16
        if w.dtype != x1.dtype or tuple(w.shape) != (Cout // 16, (C1 + C2) // (4 * vec), 9, 16, 4, vec) or not w.is_contiguous():
            raise ValueError(f"conv3x3_fused tiled weight {tuple(w.shape)}/{w.dtype} does not match input {C1}+{C2}/{x1.dtype}")
    else:
        Cout = w.shape[0]
        if w.dtype != x1.dtype or tuple(w.shape) != (Cout, 3, 3, C1 + C2):
            raise ValueError(f"conv3x3_fused weight {tuple(w.shape)}/{w.dtype} does not match input {C1}+{C2}/{x1.dtype}")
    if prescaled and (coef is None or not act):
        raise ValueError("conv3x3_fused: the prescaled form needs coef and the activation")
    mode = (_lib.ACT_SILU if act else 0) | (_lib.ACT_PRESCALED if prescaled else 0) | (_lib.W_TILED if w_tiled else 0)
    if coef is not None and (coef.dtype != torch.float32 or tuple(coef.shape) != (B, C1 + C2, 2)):
        raise ValueError("conv3x3_fused coef must be fp32 (B, C, 2)")
    if residual is not None:
        want = {_lib.RES_SAME: (B, H, W_, Cout), _lib.RES_UP: (B, H // 2, W_ // 2, Cout),
                _lib.RES_DOWN: (B, 2 * H, 2 * W_, Cout)}.get(res_mode)
        if want is None or tuple(residual.shape) != want or residual.dtype != x1.dtype:
            raise ValueError(f"conv3x3_fused residual {tuple(residual.shape)} does not fit mode {res_mode}")
    elif res_mode != _lib.RES_NONE:
        raise ValueError("res_mode set without a residual tensor")
    y = torch.empty((B, H, W_, Cout), dtype=x1.dtype, device=x1.device)
    stats = torch.empty((B, (H // 16) * (W_ // 16), Cout, 2), dtype=torch.float32, device=x1.device) if want_stats else None
    if skip is not None:
        s1, s2, sw = skip
        _dev(s1, s2, sw)
        SC1, SC2 = s1.shape[3], (0 if s2 is None else s2.shape[3])
        if residual is not None or upsample:
            raise ValueError("conv3x3_fused: the fused projection excludes a residual tensor and upsampling")
        if (tuple(s1.shape[:3]) != (B, H, W_) or (s2 is not None and tuple(s2.shape[:3]) != (B, H, W_))
                or tuple(sw.shape) != (Cout, SC1 + SC2) or sw.dtype != x1.dtype or s1.dtype != x1.dtype):
            raise ValueError("conv3x3_fused: projected input / weight shapes do not fit the output")
        end = TIMER.bracket("conv3x3_halo", 2.0 * B * H * W_ * Cout * ((C1 + C2) * 9 + SC1 + SC2))
        check(lib.mad_conv3x3_fused_skip(_ptr(x1), C1, _ptr(x2), C2, _ptr(coef), mode, _ptr(w), _ptr(bias),
                                         _ptr(s1), SC1, _ptr(s2), SC2, _ptr(sw), _ptr(y), _ptr(stats),
                                         _dtype_code(x1), B, H, W_, Cout, _stream()), "mad_conv3x3_fused_skip")
        if end is not None:
            end.record()
        return (y, stats) if want_stats else y
    end = TIMER.bracket("conv3x3_halo", 2.0 * B * H * W_ * Cout * (C1 + C2) * 9)
    check(lib.mad_conv3x3_fused(_ptr(x1), C1, _ptr(x2), C2, _ptr(coef), mode, _ptr(w), _ptr(bias), _ptr(residual),
                                res_mode, _ptr(y), _ptr(stats), _dtype_code(x1), B, H, W_, Cout, int(upsample),
                                _stream()), "mad_conv3x3_fused")
    if end is not None:
        end.record()
    return (y, stats) if want_stats else y


def conv3x3_dgrad_gn_sums(dy: Tensor, wT: Tensor, gx1: Tensor, gcoef: Tensor, act: bool, gx2: Optional[Tensor] = None,
                          drop_seed: int = 0, drop_p: float = 0.0):
    """Data gradient of a 3x3 conv (``wT`` (Cin, 3, 3, Cout): the layer's weights transposed, taps flipped) whose
    result dh is the gradient reaching h = act(GroupNorm([gx1, gx2])) -- plus, from the same launch, the partial
    sums (B, tiles, C, 2) of that GroupNorm's backward (pass them to ``gn_backward(gsum=...)``).  Returns (dh, gsum)."""
    _dev(dy, wT, gx1, gx2, gcoef)
    B, H, W_, Cin = dy.shape
    Cout = wT.shape[0]
    gC1 = gx1.shape[3]
    gC2 = 0 if gx2 is None else gx2.shape[3]
    if wT.dtype != dy.dtype or tuple(wT.shape) != (Cout, 3, 3, Cin):
        raise ValueError(f"conv3x3_dgrad_gn_sums weight {tuple(wT.shape)}/{wT.dtype} does not match gradient {Cin}/{dy.dtype}")
    if (tuple(gx1.shape[:3]) != (B, H, W_) or (gx2 is not None and tuple(gx2.shape[:3]) != (B, H, W_)) or gC1 + gC2 != Cout
            or gx1.dtype != dy.dtype or (gx2 is not None and gx2.dtype != dy.dtype)):
        raise ValueError("conv3x3_dgrad_gn_sums: the GroupNorm input does not have the shape of the data gradient")
    if gcoef.dtype != torch.float32 or tuple(gcoef.shape) != (B, Cout, 2):
        raise ValueError("conv3x3_dgrad_gn_sums coef must be fp32 (B, C, 2)")
    dh = torch.empty((B, H, W_, Cout), dtype=dy.dtype, device=dy.device)
    gsum = torch.empty((B, (H // 16) * (W_ // 16), Cout, 2), dtype=torch.float32, device=dy.device)
    end = TIMER.bracket("conv3x3_halo", 2.0 * B * H * W_ * Cout * Cin * 9)
    check(lib.mad_conv3x3_gn_bwd_sums(_ptr(dy), Cin, _ptr(wT), _ptr(dh), _ptr(gsum), _ptr(gx1), gC1, _ptr(gx2), gC2,
                                      _ptr(gcoef), int(act), drop_seed, float(drop_p), _dtype_code(dy), B, H, W_, Cout,
                                      _stream()), "mad_conv3x3_gn_bwd_sums")
    if end is not None:
        end.record()
    return dh, gsum


def conv3x3_img8(x1: Tensor, w: Tensor, bias: Optional[Tensor], coef: Optional[Tensor] = None, act: bool = True,
                 x2: Optional[Tensor] = None, residual: Optional[Tensor] = None, res_mode: int = _lib.RES_NONE,
                 want_stats: bool = False):
    """3x3 conv of an 8x8 layer with the whole image resident in LDS: the input's GroupNorm(+SiLU) from
    ``coef`` (B,C,2) or None, an optional residual (same shape, or 16x16 averaged 2x2 with RES_DOWN), and
    per-image output statistics (B,1,Cout,2)."""
    _dev(x1, x2, w, bias, coef, residual)
    B, H, W_, C1 = x1.shape
    C2 = 0 if x2 is None else x2.shape[3]
    Cout = w.shape[0]
    if (H, W_) != (8, 8) or (x2 is not None and tuple(x2.shape[:3]) != (B, 8, 8)):
        raise ValueError(f"conv3x3_img8 takes 8x8 images, got {tuple(x1.shape)}")
    if w.dtype != x1.dtype or tuple(w.shape) != (Cout, 3, 3, C1 + C2):
        raise ValueError(f"conv3x3_img8 weight {tuple(w.shape)}/{w.dtype} does not match input {C1}+{C2}/{x1.dtype}")
    if coef is not None and (coef.dtype != torch.float32 or tuple(coef.shape) != (B, C1 + C2, 2)):
        raise ValueError("conv3x3_img8 coef must be fp32 (B, C, 2)")
    if residual is not None:
        want = {_lib.RES_SAME: (B, 8, 8, Cout), _lib.RES_DOWN: (B, 16, 16, Cout)}.get(res_mode)
        if want is None or tuple(residual.shape) != want or residual.dtype != x1.dtype:
            raise ValueError(f"conv3x3_img8 residual {tuple(residual.shape)} does not fit mode {res_mode}")
    elif res_mode != _lib.RES_NONE:
        raise ValueError("res_mode set without a residual tensor")
    y = torch.empty((B, 8, 8, Cout), dtype=x1.dtype, device=x1.device)
    stats = torch.empty((B, 1, Cout, 2), dtype=torch.float32, device=x1.device) if want_stats else None
    end = TIMER.bracket("conv3x3_img8", 2.0 * B * 64 * Cout * (C1 + C2) * 9)
    check(lib.mad_conv3x3_img8(_ptr(x1), C1, _ptr(x2), C2, _ptr(coef), int(act), _ptr(w), _ptr(bias), _ptr(residual),
                               res_mode, _ptr(y), _ptr(stats), _dtype_code(x1), B, Cout, _stream()),
          "mad_conv3x3_img8")
    if end is not None:
        end.record()
    return (y, stats) if want_stats else y


def img8_conv_ok(H: int, W_: int, C1: int, C2: int, Cout: int, dtype) -> bool:
    """Shapes the image-resident 8x8 kernel covers."""
    ck = 64 if dtype == torch.bfloat16 else 32
    return (H, W_) == (8, 8) and C1 % ck == 0 and C2 % ck == 0 and Cout % 64 == 0


def img8_conv_pays(B: int, Cout: int) -> bool:
    """The image-resident kernel runs one workgroup per (image, 64 output channels) and one workgroup per
    CU; past about one and a half rounds of the 256 CUs the implicit GEMM's larger tiles (each weight
    byte reused by more pixels) are faster."""
    return B * (Cout // 64) <= 384


def fused_conv_ok(H: int, W_: int, Cout: int) -> bool:
    """Shapes the halo-tile kernel covers."""
    return H % 16 == 0 and W_ % 16 == 0 and H >= 16 and W_ >= 16 and Cout % 64 == 0


# ------------------------------------------------------------------------------ training: gradients
def conv_wgrad(x: Tensor, dy: Tensor, ksize: int, upsample: bool = False, out: Optional[Tensor] = None,
               accumulate: bool = False, dbias: Optional[Tensor] = None, bias_accumulate: bool = False) -> Tensor:
    """Weight gradient of a stride-1 conv: x NHWC (B,H,W,Cin) (or (B,H/2,W/2,Cin) with ``upsample``), dy NHWC
    (B,H,W,Cout); returns fp32 (Cout, Cin, k, k), the layout of the master weights (model.py:116-133).
    ``dbias`` (fp32 (Cout,)) also receives the bias gradient, from the same pass over dy."""
    _dev(x, dy, out, dbias)
    B, H, W_, Cout = dy.shape
    Cin = x.shape[3]
    want = (B, H // 2, W_ // 2, Cin) if upsample else (B, H, W_, Cin)
    if tuple(x.shape) != want or x.dtype != dy.dtype:
        raise ValueError(f"conv_wgrad: input {tuple(x.shape)}/{x.dtype} does not fit dy {tuple(dy.shape)}/{dy.dtype}")
    if out is None:
        out = torch.empty((Cout, Cin, ksize, ksize), dtype=torch.float32, device=x.device)
        accumulate = False
    elif tuple(out.shape) != (Cout, Cin, ksize, ksize) or out.dtype != torch.float32 or not out.is_contiguous():
        raise ValueError("conv_wgrad: out must be contiguous fp32 (Cout, Cin, k, k)")
    code = _dtype_code(x)
    nbytes = lib.mad_conv_wgrad_workspace(code, B, H, W_, Cin, Cout, ksize)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=x.device)
    end = TIMER.bracket("conv_wgrad", 2.0 * B * H * W_ * Cout * Cin * ksize * ksize)
    if dbias is not None and (tuple(dbias.shape) != (Cout,) or dbias.dtype != torch.float32 or not dbias.is_contiguous()):
        raise ValueError("conv_wgrad: dbias must be contiguous fp32 (Cout,)")
    check(lib.mad_conv_wgrad_bias(_ptr(x), _ptr(dy), _ptr(out), _ptr(dbias) if dbias is not None else None, _ptr(ws),
                                  nbytes, code, B, H, W_, Cin, Cout, ksize, int(upsample), int(accumulate),
                                  int(bias_accumulate), _stream()), "mad_conv_wgrad_bias")
    if end is not None:
        end.record()
    return out


def bias_grad(dy: Tensor, out: Optional[Tensor] = None, accumulate: bool = False) -> Tensor:
    """Sum of dy (..., C) over every axis but the last, fp32 (C,)."""
    _dev(dy, out)
    C = dy.shape[-1]
    if dy.dim() == 4 and dy.is_contiguous() and C % (8 if dy.dtype == torch.bfloat16 else 4) == 0 \
            and dy.shape[1] * dy.shape[2] >= 256:
        # large activations: the vectorised statistics kernel's sums, then the column sum of its small output
        part = gn_partial(dy)
        return bias_grad(part[..., 0].reshape(-1, C).contiguous(), out=out, accumulate=accumulate)
    rows = dy.numel() // C
    if not dy.is_contiguous():
        raise ValueError("bias_grad: dy must be contiguous")
    if out is None:
        out = torch.empty((C,), dtype=torch.float32, device=dy.device)
        accumulate = False
    nbytes = lib.mad_bias_grad_workspace(C)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=dy.device)
    check(lib.mad_bias_grad(_ptr(dy), _dtype_code(dy), rows, C, _ptr(out), _ptr(ws), nbytes, int(accumulate),
                            _stream()), "mad_bias_grad")
    return out


def conv_wgrad_edge(x: Tensor, x_nchw: bool, dy: Tensor, dy_nchw: bool, ksize: int) -> Tensor:
    """Weight gradient of the first / last conv: each operand NCHW or NHWC, bf16 or fp32; fp32 (Cout,Cin,k,k)."""
    _dev(x, dy)
    if x_nchw:
        B, Cin, H, W_ = x.shape
    else:
        B, H, W_, Cin = x.shape
    Cout = dy.shape[1] if dy_nchw else dy.shape[3]
    want = (B, Cout, H, W_) if dy_nchw else (B, H, W_, Cout)
    if tuple(dy.shape) != want or not x.is_contiguous() or not dy.is_contiguous():
        raise ValueError(f"conv_wgrad_edge: dy {tuple(dy.shape)} does not fit x {tuple(x.shape)}")
    out = torch.empty((Cout, Cin, ksize, ksize), dtype=torch.float32, device=x.device)
    nbytes = lib.mad_conv_wgrad_edge_workspace(B, H, W_, Cin, Cout, ksize)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=x.device)
    check(lib.mad_conv_wgrad_edge(_ptr(x), _dtype_code(x), int(x_nchw), _ptr(dy), _dtype_code(dy), int(dy_nchw), _ptr(out),
                                  _ptr(ws), nbytes, B, H, W_, Cin, Cout, ksize, 0, _stream()), "mad_conv_wgrad_edge")
    return out


def linear_backward(dy: Tensor, W: Tensor, xpre: Tensor, act_in: bool, need_dx: bool = True,
                    dx: Optional[Tensor] = None, accumulate: bool = False):
    """Backward of ``linear_f32(xpre, W, bias, act_in=act_in)``: returns (dx, dW, dbias); dx is the gradient
    w.r.t. ``xpre`` (before the input activation) and may be accumulated into an existing tensor."""
    _dev(dy, W, xpre, dx)
    rows, N = dy.shape
    K = W.shape[1]
    if tuple(W.shape) != (N, K) or tuple(xpre.shape) != (rows, K) or dy.dtype != torch.float32:
        raise ValueError("linear_backward: shape mismatch")
    dy = dy.contiguous()
    dW = torch.empty_like(W)
    check(lib.mad_linear_bwd_weight(_ptr(dy), _ptr(xpre), _ptr(dW), rows, K, N, int(act_in), _stream()),
          "mad_linear_bwd_weight")
    db = bias_grad(dy)
    if need_dx:
        if dx is None:
            dx, accumulate = torch.empty_like(xpre), False
        nbytes = lib.mad_linear_bwd_input_workspace(rows, K, N)
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=dy.device)
        check(lib.mad_linear_bwd_input(_ptr(dy), _ptr(W), _ptr(xpre), _ptr(dx), _ptr(ws), nbytes, rows, K, N, int(act_in),
                                       int(accumulate), _stream()), "mad_linear_bwd_input")
    return dx, dW, db


def dropout_(x: Tensor, seed: int, p: float) -> Tensor:
    """In-place dropout with the counter-based mask of (seed, element index); p = 0 is a no-op."""
    if p > 0.0:
        _dev(x)
        check(lib.mad_dropout(_ptr(x), _ptr(x), _dtype_code(x), x.numel(), seed, float(p), _stream()), "mad_dropout")
    return x


def mse_loss_grad(pred: Tensor, target: Tensor):
    """(loss as a 0-dim device tensor, dloss/dpred) of F.mse_loss(pred, target), fp32."""
    _dev(pred, target)
    if pred.shape != target.shape or pred.dtype != torch.float32 or target.dtype != torch.float32:
        raise ValueError("mse_loss_grad: fp32 tensors of one shape")
    pred, target = pred.contiguous(), target.contiguous()
    grad = torch.empty_like(pred)
    nparts = 256
    partial = torch.empty((nparts,), dtype=torch.float32, device=pred.device)
    check(lib.mad_mse_loss_grad(_ptr(pred), _ptr(target), _ptr(grad), _ptr(partial), nparts, pred.numel(), _stream()),
          "mad_mse_loss_grad")
    return partial.sum() / pred.numel(), grad


def pack_conv_weights_(master: Tensor, fwd: Tensor, dgrad: Tensor, table: Tensor, total_tiles: int) -> None:
    """Rewrite the forward and data-gradient working copies of every conv listed in ``table`` (device int64
    (n, 8): master offset, forward offset, gradient offset, Cout, Cin, k, first tile, 0) from the flat fp32
    masters, in one launch."""
    _dev(master, fwd, dgrad, table)
    if master.dtype != torch.float32 or fwd.dtype != dgrad.dtype or table.dtype != torch.int64 or table.dim() != 2 \
            or table.shape[1] != 8:
        raise ValueError("pack_conv_weights_: fp32 masters, packs of one dtype, int64 (n, 8) table")
    check(lib.mad_pack_conv_weights(_ptr(master), _ptr(fwd), _ptr(dgrad), _ptr(table), table.shape[0], int(total_tiles),
                                    _dtype_code(fwd), _stream()), "mad_pack_conv_weights")


def adamw_step_(param: Tensor, grad: Tensor, m: Tensor, v: Tensor, ema: Optional[Tensor], lr: float, beta1: float,
                beta2: float, eps: float, weight_decay: float, step: int, grad_scale: float = 1.0,
                ema_decay: float = 0.0) -> None:
    """Fused gradient clean-up + AdamW + EMA over flat fp32 buffers, in place."""
    _dev(param, grad, m, v, ema)
    n = param.numel()
    for t in (param, grad, m, v) + ((ema,) if ema is not None else ()):
        if t.dtype != torch.float32 or t.numel() != n or not t.is_contiguous():
            raise ValueError("adamw_step_: flat contiguous fp32 buffers of one length")
    check(lib.mad_adamw_step(_ptr(param), _ptr(grad), _ptr(m), _ptr(v), _ptr(ema), n, lr, beta1, beta2, eps, weight_decay,
                             step, grad_scale, ema_decay, _stream()), "mad_adamw_step")


GRAD_SAME, GRAD_FROM_HALF, GRAD_FROM_DOUBLE = 0, 1, 2     # resolution of a gradient tensor relative to x


def gn_backward(x1: Tensor, dy: Tensor, coef: Tensor, stat1: Tensor, gamma: Tensor, beta: Tensor, groups: int,
                eps: float, act: bool, x2: Optional[Tensor] = None, stat2: Optional[Tensor] = None,
                ada: Optional[Tensor] = None, ada_stride: int = 0, dada: Optional[Tensor] = None,
                dy_mode: int = GRAD_SAME, add1: Optional[Tensor] = None, add1_mode: int = GRAD_SAME,
                add2: Optional[Tensor] = None, add2_mode: int = GRAD_SAME, need_dx: bool = True,
                drop_seed: int = 0, drop_p: float = 0.0, dgamma: Optional[Tensor] = None,
                dbeta: Optional[Tensor] = None, accumulate: bool = False, gsum: Optional[Tensor] = None):
    """Backward of h = act(GroupNorm([x1, x2]) * (1 + scale) + shift).

    ``coef`` and ``stat1``/``stat2`` are the forward's (a, d) table and statistics partials; ``dy`` is the
    gradient w.r.t. h (all channels, resolution per ``dy_mode``); with ``drop_p`` > 0, h had been dropped out by
    ``gn_apply(..., drop_seed, drop_p)`` and dy is masked the same way first.  Returns (dx1, dx2, dgamma, dbeta); the
    AdaGN gradients are written into ``dada`` (a view laid out like ``ada``).  ``add1``/``add2`` are further
    gradients reaching x, added into dx.  ``dgamma``/``dbeta`` (fp32 (C,)) receive the affine gradients in place
    (added when ``accumulate``).  ``gsum`` (fp32 (B, n, C, 2)): the partial sums, when the producer of dy has
    already computed them (``conv3x3_dgrad_gn_sums``)."""
    _dev(x1, x2, dy, coef, stat1, stat2, gamma, beta, add1, add2, dgamma, dbeta, gsum)
    B, H, W_, C1 = x1.shape
    C2 = 0 if x2 is None else x2.shape[3]
    C = C1 + C2
    code = _dtype_code(x1)
    if gsum is not None:
        if gsum.dtype != torch.float32 or gsum.dim() != 4 or (gsum.shape[0], gsum.shape[2], gsum.shape[3]) != (B, C, 2) \
                or not gsum.is_contiguous():
            raise ValueError("gn_backward: gsum must be contiguous fp32 (B, n, C, 2)")
        nsg = gsum.shape[1]
    else:
        nsg = gn_nsplit(H * W_)
        gsum = torch.empty((B, nsg, C, 2), dtype=torch.float32, device=x1.device)
        check(lib.mad_gn_bwd_partial(_ptr(x1), C1, _ptr(x2), C2, _ptr(dy), dy_mode, _ptr(coef), int(act), drop_seed,
                                     float(drop_p), code, _ptr(gsum), B, H, W_, nsg, _stream()), "mad_gn_bwd_partial")
    bcoef = torch.empty((B, C, 4), dtype=torch.float32, device=x1.device)
    dgb = torch.empty((B, 2 * C), dtype=torch.float32, device=x1.device)
    ns1 = stat1.shape[1]
    ns2 = 0 if stat2 is None else stat2.shape[1]
    check(lib.mad_gn_bwd_finalize(_ptr(stat1), C1, ns1, _ptr(stat2), C2, ns2, _ptr(gsum), nsg, _ptr(gamma), _ptr(beta),
                                  None if ada is None else ada.data_ptr(), ada_stride, _ptr(bcoef), _ptr(dgb),
                                  None if dada is None else dada.data_ptr(), B, groups, H * W_, float(eps), _stream()),
          "mad_gn_bwd_finalize")
    if dgamma is None or dbeta is None:
        dgamma = torch.empty((C,), dtype=torch.float32, device=x1.device)
        dbeta = torch.empty((C,), dtype=torch.float32, device=x1.device)
        accumulate = False
    elif dgamma.numel() != C or dbeta.numel() != C or dgamma.dtype != torch.float32 or dbeta.dtype != torch.float32:
        raise ValueError("gn_backward: dgamma / dbeta must be fp32 with C elements")
    check(lib.mad_gn_affine_grad(_ptr(dgb), B, C, _ptr(dgamma), _ptr(dbeta), int(accumulate), _stream()),
          "mad_gn_affine_grad")
    dx1 = dx2 = None
    if need_dx:
        dx1 = torch.empty_like(x1)
        dx2 = None if x2 is None else torch.empty_like(x2)
        check(lib.mad_gn_bwd_apply(_ptr(x1), C1, _ptr(x2), C2, _ptr(dy), dy_mode, _ptr(coef), int(act), drop_seed, float(drop_p),
                                   _ptr(bcoef), _ptr(add1), add1_mode, _ptr(add2), add2_mode, _ptr(dx1), _ptr(dx2), code, B, H, W_,
                                   _stream()), "mad_gn_bwd_apply")
    return dx1, dx2, dgamma, dbeta


# ------------------------------------------------------------------------------ GroupNorm
def gn_nsplit(HW: int) -> int:
    return min(64, max(1, HW // 64))


def gn_partial(x1: Tensor, x2: Optional[Tensor] = None) -> Tensor:
    """Per-(sample, split, channel) sum / sum of squares: (B, nsplit, C, 2)."""
    _dev(x1, x2)
    B, H, W_, C1 = x1.shape
    C2 = 0 if x2 is None else x2.shape[3]
    HW = H * W_
    nsplit = gn_nsplit(HW)
    partial = torch.empty((B, nsplit, C1 + C2, 2), dtype=torch.float32, device=x1.device)
    check(lib.mad_gn_partial(_ptr(x1), C1, _ptr(x2), C2, _dtype_code(x1), _ptr(partial), B, HW, nsplit, _stream()),
          "mad_gn_partial")
    return partial


def gn_finalize(partial: Tensor, HW: int, gamma: Tensor, beta: Tensor, groups: int, eps: float,
                ada: Optional[Tensor] = None, ada_stride: int = 0, partial2: Optional[Tensor] = None,
                scale: float = 1.0) -> Tensor:
    """Partial sums (B, nsplit, C, 2) over HW pixels (optionally a second slab for concatenated
    channels) -> coef (B, C, 2) with y = x*a + d, every pair multiplied by ``scale``."""
    _dev(partial, partial2, gamma, beta)
    B, ns1, C1, _ = partial.shape
    ns2, C2 = (0, 0) if partial2 is None else (partial2.shape[1], partial2.shape[2])
    C = C1 + C2
    if gamma.numel() != C or beta.numel() != C:
        raise ValueError("GroupNorm gain/offset length does not match the channel count")
    coef = torch.empty((B, C, 2), dtype=torch.float32, device=partial.device)
    check(lib.mad_gn_finalize_scaled(_ptr(partial), C1, ns1, _ptr(partial2), C2, ns2, _ptr(gamma), _ptr(beta),
                                     None if ada is None else ada.data_ptr(), ada_stride, _ptr(coef), B, groups, HW,
                                     float(eps), float(scale), _stream()), "mad_gn_finalize")
    return coef


def gn_coef(x1: Tensor, gamma: Tensor, beta: Tensor, groups: int, eps: float, x2: Optional[Tensor] = None,
            ada: Optional[Tensor] = None, ada_stride: int = 0) -> Tensor:
    """Statistics + folding in one call: returns coef (B, C, 2) with y = x*a + d.  `ada` is a view
    whose data pointer is the first scale element of sample 0; rows are `ada_stride` floats apart."""
    B, H, W_, _ = x1.shape
    return gn_finalize(gn_partial(x1, x2), H * W_, gamma, beta, groups, eps, ada=ada, ada_stride=ada_stride)


def gn_coef_static(x1: Tensor, gamma: Tensor, beta: Tensor, groups: int, eps: float, x2: Optional[Tensor] = None,
                   ada: Optional[Tensor] = None, ada_stride: int = 0, out: Optional[Tensor] = None) -> Tensor:
    """``gn_coef`` through the library's entry point that allocates nothing: the partial sums stay in a device
    array inside the library.  With ``out`` given (fp32 (B, C, 2)) this wrapper allocates nothing either."""
    _dev(x1, x2, gamma, beta, out)
    B, H, W_, C1 = x1.shape
    C2 = 0 if x2 is None else x2.shape[3]
    if out is None:
        out = torch.empty((B, C1 + C2, 2), dtype=torch.float32, device=x1.device)
    elif out.dtype != torch.float32 or tuple(out.shape) != (B, C1 + C2, 2):
        raise ValueError("gn_coef_static: out must be fp32 (B, C, 2)")
    check(lib.mad_gn_coef_static(_ptr(x1), C1, _ptr(x2), C2, _dtype_code(x1), _ptr(gamma), _ptr(beta),
                                 None if ada is None else ada.data_ptr(), ada_stride, _ptr(out), B, H * W_, groups,
                                 float(eps), _stream()), "mad_gn_coef_static")
    return out


def gn_apply(x1: Tensor, coef: Tensor, act: bool, x2: Optional[Tensor] = None, pool: bool = False,
             drop_seed: int = 0, drop_p: float = 0.0) -> Tensor:
    """y = act(x*a + d) (optionally 2x2-averaged); with ``drop_p`` > 0 the result is dropped out in the same pass
    with the counter-based mask of ``drop_seed`` (training)."""
    _dev(x1, x2, coef)
    B, H, W_, C1 = x1.shape
    C2 = 0 if x2 is None else x2.shape[3]
    Ho, Wo = (H // 2, W_ // 2) if pool else (H, W_)
    y = torch.empty((B, Ho, Wo, C1 + C2), dtype=x1.dtype, device=x1.device)
    if drop_p > 0.0:
        if pool:
            raise ValueError("gn_apply: dropout and pooling are not combined")
        check(lib.mad_gn_apply_dropout(_ptr(x1), C1, _ptr(x2), C2, _ptr(coef), _ptr(y), _dtype_code(x1), B, H, W_, int(act),
                                       drop_seed, float(drop_p), _stream()), "mad_gn_apply_dropout")
        return y
    check(lib.mad_gn_apply(_ptr(x1), C1, _ptr(x2), C2, _ptr(coef), _ptr(y), _dtype_code(x1), B, H, W_, int(act),
                           int(pool), _stream()), "mad_gn_apply")
    return y


# ------------------------------------------------------------------------------ attention
def attention(qkv: Tensor, heads: int, want_lse: bool = False):
    """qkv (B,H,W,3C) with per-head [q|k|v] channel blocks -> (B,H,W,C); with ``want_lse`` also the
    per-query log-sum-exp (B, heads, H*W) that ``attention_backward`` needs."""
    _dev(qkv)
    B, H, W_, C3 = qkv.shape
    C = C3 // 3
    if C != heads * 64:
        raise ValueError(f"attention needs head_dim 64: C={C}, heads={heads}")
    out = torch.empty((B, H, W_, C), dtype=qkv.dtype, device=qkv.device)
    end = TIMER.bracket("attention", 4.0 * B * (H * W_) ** 2 * C)
    if want_lse:
        lse = torch.empty((B, heads, H * W_), dtype=torch.float32, device=qkv.device)
        check(lib.mad_attention_lse(_ptr(qkv), _ptr(out), _ptr(lse), _dtype_code(qkv), B, H * W_, heads, _stream()),
              "mad_attention_lse")
    else:
        check(lib.mad_attention(_ptr(qkv), _ptr(out), _dtype_code(qkv), B, H * W_, heads, _stream()), "mad_attention")
    if end is not None:
        end.record()
    return (out, lse) if want_lse else out


def attention_backward(qkv: Tensor, out: Tensor, dout: Tensor, lse: Tensor, heads: int) -> Tensor:
    """Gradient of ``attention`` w.r.t. qkv (same layout)."""
    _dev(qkv, out, dout, lse)
    B, H, W_, C3 = qkv.shape
    if out.shape != dout.shape or out.dtype != qkv.dtype or dout.dtype != qkv.dtype or not dout.is_contiguous():
        raise ValueError("attention_backward: out / dout must be contiguous tensors of the forward's output shape")
    dqkv = torch.empty_like(qkv)
    dsum = torch.empty_like(lse)
    end = TIMER.bracket("attention_bwd", 10.0 * B * (H * W_) ** 2 * (C3 // 3))
    check(lib.mad_attention_bwd(_ptr(qkv), _ptr(out), _ptr(dout), _ptr(lse), _ptr(dsum), _ptr(dqkv), _dtype_code(qkv),
                                B, H * W_, heads, _stream()), "mad_attention_bwd")
    if end is not None:
        end.record()
    return dqkv
