"""One training step of the denoiser on the HIP kernels: forward that keeps what the backward needs,
hand-written backward, gradient all-reduce, fused AdamW + EMA.

Mirrors the loop body of the reference's ``main.py:223-257`` (``/root/reference/main.py``):

    t ~ randint, noise ~ randn, x_t = add_noise(x0, noise, t)            main.py:235-237
    pred = model(x_t, t, class_labels=label)      (train mode: dropout)   main.py:240
    loss = mse_loss(pred, noise); loss.backward()                        main.py:243-247
    grads = nan_to_num(grads); AdamW step; EMA step                      main.py:248-257

The reference differentiates with torch autograd; here every gradient is one of the kernels declared in
``include/mad_hip.h`` (conv weight gradients on MFMA, data gradients through the forward conv kernels with
transposed and flipped weights, GroupNorm/SiLU/AdaGN backward, flash-style attention backward, linear
backward).  Activations are bf16 (or fp32), master weights / gradients / optimizer state fp32 in flat
buffers, so the data-parallel gradient sum is a handful of large RCCL all-reduces over contiguous ranges.
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Tuple

import torch
from torch import Tensor

from . import _lib, ops
from .model import DhariwalUNet
from .unet_spec import GN_EPS, BlockSpec, gn_groups


def allreduce_flat(flat: Tensor, bucket_elems: int, group=None) -> int:
    """Sum ``flat`` over the ranks of ``group`` in place, as a few large asynchronous all-reduces over
    contiguous ranges (RCCL on GPUs, gloo on CPU tensors); returns the number of collectives issued."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return 0
    n = flat.numel()
    works = [dist.all_reduce(flat[o:min(o + bucket_elems, n)], group=group, async_op=True)
             for o in range(0, n, bucket_elems)]
    for w in works:
        w.wait()
    return len(works)


def warmup_lr(base_lr: float, optimizer_steps_done: int, warmup_steps: int) -> float:
    """transformers ``get_constant_schedule_with_warmup`` (main.py:179-182): the k-th optimizer step (k from 0)
    runs at base_lr * k / warmup while k < warmup."""
    if warmup_steps > 0 and optimizer_steps_done < warmup_steps:
        return base_lr * optimizer_steps_done / max(1, warmup_steps)
    return base_lr


def _dgrad_weight(e: dict, key: str) -> Tensor:
    """(Cout,k,k,Cin) forward pack -> (Cin,k,k,Cout) with the taps flipped: the forward conv kernels then
    compute the data gradient, dx = conv(dy, w').  Kept beside the forward pack (both are rebuilt after an
    optimizer step)."""
    tk = key + ".T"
    if tk not in e:
        e[tk] = e[key].flip(1, 2).permute(3, 1, 2, 0).contiguous()
    return e[tk]


class _Saved:
    """What one UNetBlock's backward needs from its forward."""
    __slots__ = ("x", "x_part", "skip", "skip_part", "c0", "h0", "h", "h_part", "c1", "h1", "seed", "y", "y_part",
                 "c2", "hn", "qkv", "att", "lse")

    def __init__(self):
        for k in self.__slots__:
            setattr(self, k, None)


class UNetTrainer:
    """Owns flat fp32 parameter / gradient / AdamW / EMA buffers of a ``DhariwalUNet`` and runs training steps."""

    def __init__(self, unet: DhariwalUNet, lr: float = 1e-4, betas=(0.95, 0.999), eps: float = 1e-7,
                 weight_decay: float = 1e-2, dropout: float = 0.10, ema: bool = True, ema_max_decay: float = 0.9999,
                 ema_inv_gamma: float = 1.0, ema_power: float = 0.75, process_group=None, bucket_mb: float = 128,
                 seed: int = 0):
        self.unet = unet
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.dropout = float(dropout)
        self.ema_cfg = (ema_max_decay, ema_inv_gamma, ema_power)
        self.group = process_group
        self.bucket_elems = max(1, int(bucket_mb * (1 << 20)) // 4)
        self.step_count = 0
        self.seed = seed
        self._drop_calls = 0
        self._micro_index = 0
        self._accumulating = False
        self.micro_batches = 0
        params = list(unet.named_parameters())
        if not params or not params[0][1].is_cuda:
            raise RuntimeError("UNetTrainer needs the model on the GPU (the training step has no CPU path)")
        dev = params[0][1].device
        total = sum(p.numel() for _, p in params)
        self.flat = torch.empty(total, dtype=torch.float32, device=dev)
        self.gflat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.m = torch.zeros_like(self.flat)
        self.v = torch.zeros_like(self.flat)
        self.grads: Dict[str, Tensor] = {}
        self._offset: Dict[str, int] = {}                  # first element of a parameter in the flat buffers
        self._param: Dict[str, Tensor] = {}
        self._arena_of = None                              # the pack whose conv copies live in the arenas below
        off = 0
        for name, p in params:
            n = p.numel()
            self._offset[name] = off
            self.flat[off:off + n].copy_(p.data.reshape(-1).to(torch.float32))
            p.data = self.flat[off:off + n].view(p.shape)
            self._param[name] = p.data
            p.requires_grad_(False)
            self.grads[name] = self.gflat[off:off + n].view(p.shape)
            off += n
        self.ema = self.flat.clone() if ema else None
        unet.invalidate()
        # Buckets for the data-parallel sum: runs of consecutive parameters of at least `bucket_mb`.  The backward
        # writes gradients roughly from the last parameter to the first, so a bucket is complete -- and its
        # all-reduce can start, overlapping the rest of the backward -- when all of its parameters are written.
        self.buckets: List[Tuple[int, int]] = []          # [lo, hi) ranges of the flat buffer
        self._bucket_of: Dict[str, int] = {}
        self._bucket_size: List[int] = []
        lo = off = cnt = 0
        for name, p in params:
            self._bucket_of[name] = len(self.buckets)
            off += p.numel()
            cnt += 1
            if off - lo >= self.bucket_elems:
                self.buckets.append((lo, off)); self._bucket_size.append(cnt)
                lo, cnt = off, 0
        if off > lo:
            self.buckets.append((lo, off)); self._bucket_size.append(cnt)
        self._pending: List[int] = []
        self._works: list = []
        self._launched: List[int] = []                     # bucket indices in launch order (last backward)
        self._overlap = False
        self._reduce_fn = None                             # test hook: called instead of dist.all_reduce

    # ------------------------------------------------------------------ working copies of the weights
    REPACK = os.environ.get("MAD_TRAIN_REPACK", "1") != "0"
    FUSE_SKIP = os.environ.get("MAD_TRAIN_FUSE_SKIP", "1") != "0"
    FOLD_GN_SUMS = os.environ.get("MAD_TRAIN_FOLD_GN_SUMS", "1") != "0"

    def _adopt_pack(self, P: dict) -> None:
        """Move the conv working copies of a freshly built pack into two arenas (forward layout; transposed with
        flipped taps for the data gradient) that one kernel rewrites from the masters after every update."""
        dev, dt = self.flat.device, P["dtype"]
        rows, targets, off, tiles = [], [], 0, 0
        for e in P["blocks"]:
            n = e["spec"].name
            for key in ("conv0.w", "conv1.w", "skip.w", "proj.w"):
                if key not in e:
                    continue
                Cout, k, _, Cin = e[key].shape
                rows.append([self._offset[f"{n}.{key[:-1]}weight"], off, off, Cout, Cin, k, tiles, 0])
                targets.append((e, key, Cout, Cin, k, off))
                off += -(-Cout * Cin * k * k // 128) * 128
                tiles += -(-Cout // 32) * -(-Cin // 32)
        self._arena_f = torch.empty(off, dtype=dt, device=dev)
        self._arena_t = torch.empty(off, dtype=dt, device=dev)
        for e, key, Cout, Cin, k, o in targets:
            n = Cout * Cin * k * k
            e[key] = self._arena_f[o:o + n].view(Cout, k, k, Cin)
            e[key + ".T"] = self._arena_t[o:o + n].view(Cin, k, k, Cout)
            if key == "skip.w":
                e["skip.w2d"] = e[key].reshape(Cout, -1)
        self._pack_table = torch.tensor(rows, dtype=torch.int64, device=dev)
        self._pack_tiles = tiles
        self._arena_of = P
        ops.pack_conv_weights_(self.flat, self._arena_f, self._arena_t, self._pack_table, tiles)

    def _refresh_pack(self) -> None:
        """After an update: rewrite the working copies in place (model.py's pack stays valid: biases, norm and
        embedding weights alias the masters; the conv copies come from one kernel; the few regrouped tensors
        are recomputed here as `DhariwalUNet._pack` computes them)."""
        P = self.unet._packed
        if not self.REPACK or P is None or self._arena_of is not P:
            self.unet.invalidate()
            self._arena_of = None
            return
        ops.pack_conv_weights_(self.flat, self._arena_f, self._arena_t, self._pack_table, self._pack_tiles)
        W, s = self._param, self.unet.spec
        for e in P["blocks"]:
            b = e["spec"]
            n = b.name
            e.pop("conv0.wh", None)          # the inference path's own weight copies: rebuilt when it next runs
            e.pop("conv1.wh", None)
            if b.skip_conv:
                torch.add(e["conv1.b"], e["skip.b"], out=e["conv1+skip.b"])
            if b.attention:
                co, heads = b.out_channels, b.num_heads
                if "qkv.idx" not in e:
                    e["qkv.idx"] = torch.arange(3 * co, device=self.flat.device).reshape(heads, 64, 3).permute(0, 2, 1).reshape(-1)
                e["qkv.w"].copy_(W[f"{n}.qkv.weight"].permute(0, 2, 3, 1)[e["qkv.idx"]])
                e["qkv.b"].copy_(W[f"{n}.qkv.bias"][e["qkv.idx"]])
                e.pop("qkv.w.T", None)
        torch.cat([W[f"{b.name}.affine.weight"] for b in s.blocks()], 0, out=P["aff.w"])
        torch.cat([W[f"{b.name}.affine.bias"] for b in s.blocks()], 0, out=P["aff.b"])
        P["out.w"].copy_(W["out_conv.weight"].permute(0, 2, 3, 1))
        P["out.w32"][:s.out_channels].copy_(P["out.w"])
        self.unet._pack_version += 1
        P["version"] = self.unet._pack_version

    # ------------------------------------------------------------------ helpers
    def _set(self, name: str, g: Tensor) -> None:
        """Store (first micro-batch) or add (gradient accumulation) one parameter's gradient."""
        dst = self.grads[name]
        if self._accumulating:
            dst.add_(g.reshape(dst.shape))
        else:
            dst.copy_(g.reshape(dst.shape))
        self._mark(name)

    def _mark(self, name: str) -> None:
        """Parameter ``name``'s gradient is final for this backward: start its bucket's all-reduce when the bucket is whole."""
        if not self._overlap:
            return
        b = self._bucket_of[name]
        self._pending[b] -= 1
        if self._pending[b] == 0:
            self._launch(b)

    def _launch(self, b: int) -> None:
        lo, hi = self.buckets[b]
        self._launched.append(b)
        if self._reduce_fn is not None:
            self._reduce_fn(lo, hi)
            return
        import torch.distributed as dist
        # the collective is ordered after the kernels already queued on the current stream and runs on the
        # process group's own stream, beside the rest of the backward
        self._works.append(dist.all_reduce(self.gflat[lo:hi], group=self.group, async_op=True))

    def _wgrad(self, name: str, x: Tensor, dy: Tensor, ksize: int, upsample: bool = False,
               bias: Optional[str] = None, same_bias: Optional[str] = None) -> None:
        """Conv weight gradient written (or, when accumulating micro-batches, added) straight into the flat buffer;
        ``bias`` names the layer's bias, whose gradient comes out of the same pass over dy, and ``same_bias`` a second
        bias that sees the same dy (a block's conv1 and its skip projection) and therefore has the same gradient."""
        ops.conv_wgrad(x, dy, ksize, upsample=upsample, out=self.grads[name], accumulate=self._accumulating,
                       dbias=self.grads[bias] if bias else None, bias_accumulate=self._accumulating)
        if same_bias:
            self.grads[same_bias].copy_(self.grads[bias])
            self._mark(same_bias)
        self._mark(name)
        if bias:
            self._mark(bias)

    @staticmethod
    def _conv3(x: Tensor, w: Tensor, b: Optional[Tensor], upsample: bool = False, residual: Optional[Tensor] = None,
               res_mode: int = _lib.RES_NONE, want_stats: bool = True):
        """3x3 conv of a materialised input on whichever kernel covers the shape; returns (y, partials or None)."""
        B, Hs, Ws, C = x.shape
        H, W_ = (2 * Hs, 2 * Ws) if upsample else (Hs, Ws)
        Cout = w.shape[0]
        cm = 32 if x.dtype == torch.bfloat16 else 16
        if ops.fused_conv_ok(H, W_, Cout) and C % cm == 0:
            r = ops.conv3x3_fused(x, w, b, coef=None, upsample=upsample, residual=residual, res_mode=res_mode,
                                  want_stats=want_stats)
            return r if want_stats else (r, None)
        stats = want_stats and (H * W_) % 64 == 0
        r = ops.conv2d(x, w, b, residual=residual, res_mode=res_mode, upsample=upsample, want_stats=stats)
        return r if stats else (r, None)

    @staticmethod
    def _dgrad3(dy: Tensor, wT: Tensor) -> Tensor:
        """Data gradient of a 3x3 / 1x1 stride-1 conv: the forward kernels on the flipped, transposed weights."""
        if wT.shape[1] == 3:
            return UNetTrainer._conv3(dy, wT, None, want_stats=False)[0]
        return ops.conv2d(dy, wT, None)

    def _dgrad3_sums(self, dy: Tensor, wT: Tensor, gx1: Tensor, gcoef: Tensor, gx2: Optional[Tensor] = None,
                     drop_seed: int = 0, drop_p: float = 0.0):
        """Data gradient of a 3x3 conv whose input was silu(GroupNorm([gx1, gx2])) at the same resolution; returns
        (dh, partial sums of that GroupNorm's backward, or None when the shape is not covered by the fused launch)."""
        B, H, W_, C = dy.shape
        Cg = wT.shape[0]
        cm = 32 if dy.dtype == torch.bfloat16 else 16
        vec = 8 if dy.dtype == torch.bfloat16 else 4
        if (self.FOLD_GN_SUMS and wT.shape[1] == 3 and ops.fused_conv_ok(H, W_, Cg) and C % cm == 0
                and tuple(gx1.shape[:3]) == (B, H, W_) and gx1.shape[3] % vec == 0):
            return ops.conv3x3_dgrad_gn_sums(dy, wT, gx1, gcoef, True, gx2=gx2, drop_seed=drop_seed, drop_p=drop_p)
        return self._dgrad3(dy, wT), None

    def _part(self, t: Tensor, part: Optional[Tensor]) -> Tensor:
        return part if part is not None else ops.gn_partial(t)

    # ------------------------------------------------------------------ forward (train mode)
    def _block_fwd(self, e: dict, x: Tensor, x_part: Optional[Tensor], skip: Optional[Tensor],
                   skip_part: Optional[Tensor], ada: Tensor, ada_total: int):
        b: BlockSpec = e["spec"]
        sv = _Saved()
        hw_in, hw_out = b.res_in ** 2, b.res_out ** 2
        sv.x, sv.skip = x, skip
        sv.x_part = self._part(x, x_part)
        sv.skip_part = None if skip is None else self._part(skip, skip_part)
        sv.c0 = ops.gn_finalize(sv.x_part, hw_in, e["norm0.g"], e["norm0.b"], gn_groups(b.in_channels), GN_EPS,
                                partial2=sv.skip_part)
        ada_v = ada[:, e["ada_off"]:]
        sv.h0 = ops.gn_apply(x, sv.c0, act=True, x2=skip, pool=b.down)
        sv.h, sv.h_part = self._conv3(sv.h0, e["conv0.w"], e["conv0.b"], upsample=b.up)
        sv.h_part = self._part(sv.h, sv.h_part)
        sv.c1 = ops.gn_finalize(sv.h_part, hw_out, e["norm1.g"], e["norm1.b"], gn_groups(b.out_channels), GN_EPS,
                                ada=ada_v, ada_stride=ada_total)
        if self.dropout > 0.0:                   # dropout fused into the apply pass; the backward rebuilds the mask
            self._drop_calls += 1
            # one mask per (optimizer step, micro-batch, layer): accumulated micro-batches draw independently
            sv.seed = (self.seed << 32) ^ (self.step_count << 16) ^ (self._micro_index << 10) ^ self._drop_calls
        sv.h1 = ops.gn_apply(sv.h, sv.c1, act=True, drop_seed=sv.seed or 0, drop_p=self.dropout if sv.seed is not None else 0.0)
        cm = 32 if x.dtype == torch.bfloat16 else 16
        if (b.skip_conv and self.FUSE_SKIP and ops.fused_conv_ok(b.res_out, b.res_out, b.out_channels)
                and b.out_channels % cm == 0 and x.shape[3] % cm == 0 and (skip is None or skip.shape[3] % cm == 0)):
            # the 1x1 projection of the block input rides in conv1's launch (more reduction steps of the same
            # accumulators) instead of a pass of its own that writes a residual tensor conv1 then re-reads
            sv.y, sv.y_part = ops.conv3x3_fused(sv.h1, e["conv1.w"], e["conv1+skip.b"], coef=None, want_stats=True,
                                                skip=(x, skip, e["skip.w2d"]))
            res = mode = None
        elif b.skip_conv:
            res, mode = ops.conv2d(x, e["skip.w"], e["skip.b"], x2=skip), _lib.RES_SAME
        elif b.up:
            res, mode = x, _lib.RES_UP
        elif b.down:
            res, mode = x, _lib.RES_DOWN
        else:
            res, mode = x, _lib.RES_SAME
        if mode is not None:
            sv.y, sv.y_part = self._conv3(sv.h1, e["conv1.w"], e["conv1.b"], residual=res, res_mode=mode)
        out, out_part = sv.y, sv.y_part
        if b.attention:
            sv.y_part = self._part(sv.y, sv.y_part)
            sv.c2 = ops.gn_finalize(sv.y_part, hw_out, e["norm2.g"], e["norm2.b"], gn_groups(b.out_channels), GN_EPS)
            sv.hn = ops.gn_apply(sv.y, sv.c2, act=False)
            sv.qkv = ops.conv2d(sv.hn, e["qkv.w"], e["qkv.b"])
            sv.att, sv.lse = ops.attention(sv.qkv, b.num_heads, want_lse=True)
            stats = hw_out % 64 == 0
            r = ops.conv2d(sv.att, e["proj.w"], e["proj.b"], residual=sv.y, res_mode=_lib.RES_SAME, want_stats=stats)
            out, out_part = r if stats else (r, None)
        return out, out_part, sv

    def forward(self, x: Tensor, t: Tensor, class_labels: Optional[Tensor]):
        """Train-mode ``DhariwalUNet.forward`` (model.py:553-590); returns (pred NCHW fp32, tape)."""
        u, s = self.unet, self.unet.spec
        P = u._pack(x.device)
        if self.REPACK and self._arena_of is not P:
            self._adopt_pack(P)
        dt = P["dtype"]
        B = x.shape[0]
        x = x.to(torch.float32).contiguous()
        t = t.to(device=x.device, dtype=torch.float32).reshape(-1).contiguous()
        tape: dict = {"x": x, "P": P}
        emb0 = ops.pos_embedding(t, P["freqs"], B)
        pre1 = ops.linear_f32(emb0, P["map_layer0.weight"], P["map_layer0.bias"])
        pre2 = ops.linear_f32(pre1, P["map_layer1.weight"], P["map_layer1.bias"], act_in=True)
        lab = None
        if s.label_dim:
            lab = class_labels.to(device=x.device, dtype=torch.float32).reshape(-1, s.label_dim)
            if lab.shape[0] == 1 and B > 1:
                lab = lab.expand(B, -1)
            lab = lab.contiguous()
            ops.linear_f32(lab, P["map_label.weight"], None, out=pre2, accumulate=True)
        ada = ops.linear_f32(pre2, P["aff.w"], P["aff.b"], act_in=True)
        total = P["aff.total"]
        tape.update(emb0=emb0, pre1=pre1, pre2=pre2, lab=lab, ada=ada)

        h = ops.conv_in(x, P["in.w"], P["in.b"], dt)
        h_part = None
        skips = [(h, None)]
        saved: List[_Saved] = []
        n_enc = len(s.enc)
        for i, e in enumerate(P["blocks"]):
            b: BlockSpec = e["spec"]
            if i < n_enc:
                h, h_part, sv = self._block_fwd(e, h, h_part, None, None, ada, total)
                skips.append((h, h_part))
            else:
                sk, sk_part = skips.pop() if b.cat_skip else (None, None)
                h, h_part, sv = self._block_fwd(e, h, h_part, sk, sk_part, ada, total)
            saved.append(sv)
        R = s.img_resolution
        h_part = self._part(h, h_part)
        c = ops.gn_finalize(h_part, R * R, P["out_norm.g"], P["out_norm.b"], gn_groups(s.final_channels), GN_EPS)
        hact = ops.gn_apply(h, c, act=True)
        fused_out = R % 16 == 0 and s.out_channels <= 32 and s.final_channels % 32 == 0
        if fused_out:
            pred = ops.conv_out_fused(hact, P["out.w32"], P["out.b"], s.out_channels, coef=None)
        else:
            pred = ops.conv_out(hact, P["out.w"], P["out.b"])
        tape.update(saved=saved, h_last=h, h_last_part=h_part, c_out=c, hact=hact)
        return pred, tape

    # ------------------------------------------------------------------ backward
    def _block_bwd(self, e: dict, sv: _Saved, dout: Tensor, dskip_of_x: Optional[Tensor], dada: Tensor,
                   ada: Tensor, ada_total: int):
        """Gradients of one block's parameters; returns (dx, dskip): the gradients w.r.t. the block input and
        w.r.t. the concatenated encoder skip (None without one).  ``dskip_of_x`` is the gradient that reaches
        the block INPUT through its other use as a decoder skip."""
        b: BlockSpec = e["spec"]
        n = b.name
        hw_out = b.res_out ** 2
        ada_v, dada_v = ada[:, e["ada_off"]:], dada[:, e["ada_off"]:]
        dy = dout
        if b.attention:
            self._wgrad(f"{n}.proj.weight", sv.att, dout, 1, bias=f"{n}.proj.bias")
            datt = ops.conv2d(dout, _dgrad_weight(e, "proj.w"), None)
            dqkv = ops.attention_backward(sv.qkv, sv.att, datt, sv.lse, b.num_heads)
            # the packed qkv rows are a permutation of the reference's (model.py:359-363): undo it for the masters
            co, heads = b.out_channels, b.num_heads
            idx = torch.arange(3 * co, device=dout.device).reshape(heads, 64, 3).permute(0, 2, 1).reshape(-1)
            gb = torch.empty((3 * co,), dtype=torch.float32, device=dout.device)
            gw = ops.conv_wgrad(sv.hn, dqkv, 1, dbias=gb)
            if self._accumulating:
                self.grads[f"{n}.qkv.weight"].index_add_(0, idx, gw)
                self.grads[f"{n}.qkv.bias"].index_add_(0, idx, gb)
            else:
                self.grads[f"{n}.qkv.weight"].index_copy_(0, idx, gw)
                self.grads[f"{n}.qkv.bias"].index_copy_(0, idx, gb)
            self._mark(f"{n}.qkv.weight")
            self._mark(f"{n}.qkv.bias")
            dhn = ops.conv2d(dqkv, _dgrad_weight(e, "qkv.w"), None)
            dy, _, dg, db = ops.gn_backward(sv.y, dhn, sv.c2, sv.y_part, e["norm2.g"], e["norm2.b"],
                                            gn_groups(b.out_channels), GN_EPS, act=False, add1=dout,
                                            dgamma=self.grads[f"{n}.norm2.weight"], dbeta=self.grads[f"{n}.norm2.bias"],
                                            accumulate=self._accumulating)
            self._mark(f"{n}.norm2.weight")
            self._mark(f"{n}.norm2.bias")
        # conv1 and the AdaGN in front of it
        self._wgrad(f"{n}.conv1.weight", sv.h1, dy, 3, bias=f"{n}.conv1.bias",
                    same_bias=f"{n}.skip.bias" if b.skip_conv else None)
        drop_p = self.dropout if sv.seed is not None else 0.0
        dh1, gs1 = self._dgrad3_sums(dy, _dgrad_weight(e, "conv1.w"), sv.h, sv.c1, drop_seed=sv.seed or 0, drop_p=drop_p)
        dh, _, dg, db = ops.gn_backward(sv.h, dh1, sv.c1, sv.h_part, e["norm1.g"], e["norm1.b"],
                                        gn_groups(b.out_channels), GN_EPS, act=True, ada=ada_v, ada_stride=ada_total,
                                        dada=dada_v, drop_seed=sv.seed or 0, drop_p=drop_p,
                                        dgamma=self.grads[f"{n}.norm1.weight"], dbeta=self.grads[f"{n}.norm1.bias"],
                                        accumulate=self._accumulating, gsum=gs1)
        self._mark(f"{n}.norm1.weight")
        self._mark(f"{n}.norm1.bias")
        # conv0 (input possibly resampled after norm0 + SiLU)
        self._wgrad(f"{n}.conv0.weight", sv.h0, dh, 3, upsample=b.up, bias=f"{n}.conv0.bias")
        mode0 = ops.GRAD_FROM_DOUBLE if b.up else (ops.GRAD_FROM_HALF if b.down else ops.GRAD_SAME)
        if mode0 == ops.GRAD_SAME:
            dh0, gs0 = self._dgrad3_sums(dh, _dgrad_weight(e, "conv0.w"), sv.x, sv.c0, gx2=sv.skip)
        else:                                # norm0 runs at the other resolution: its sums need the resampled gradient
            dh0, gs0 = self._dgrad3(dh, _dgrad_weight(e, "conv0.w")), None
        # the residual branch's gradient w.r.t. the block input
        if b.skip_conv:
            if sv.skip is None:
                self._wgrad(f"{n}.skip.weight", sv.x, dy, 1)
            else:                            # the projection's input is the concatenation: one gradient per source
                gsk, c1 = self.grads[f"{n}.skip.weight"], sv.x.shape[3]
                for sl, src in ((slice(0, c1), sv.x), (slice(c1, None), sv.skip)):
                    g = ops.conv_wgrad(src, dy, 1)
                    if self._accumulating:
                        gsk[:, sl].add_(g)
                    else:
                        gsk[:, sl].copy_(g)
                self._mark(f"{n}.skip.weight")
            dres, res_mode = ops.conv2d(dy, _dgrad_weight(e, "skip.w"), None), ops.GRAD_SAME
        else:
            dres, res_mode = dy, mode0
        add2 = None
        if dskip_of_x is not None:
            if sv.skip is not None:          # the extra gradient covers x's channels only: widen it with zeros
                pad = torch.zeros_like(sv.skip)
                add2 = torch.cat([dskip_of_x, pad], dim=-1)
            else:
                add2 = dskip_of_x
        dx, dskip, dg, db = ops.gn_backward(sv.x, dh0, sv.c0, sv.x_part, e["norm0.g"], e["norm0.b"],
                                            gn_groups(b.in_channels), GN_EPS, act=True, x2=sv.skip,
                                            stat2=sv.skip_part, dy_mode=mode0, add1=dres, add1_mode=res_mode,
                                            add2=add2,
                                            dgamma=self.grads[f"{n}.norm0.weight"], dbeta=self.grads[f"{n}.norm0.bias"],
                                            accumulate=self._accumulating, gsum=gs0)
        self._mark(f"{n}.norm0.weight")
        self._mark(f"{n}.norm0.bias")
        return dx, dskip

    def backward(self, tape: dict, dpred: Tensor) -> None:
        """Fill ``self.grads`` from the gradient of the loss w.r.t. the prediction (NCHW fp32)."""
        u, s = self.unet, self.unet.spec
        P = tape["P"]
        dt = P["dtype"]
        B = dpred.shape[0]
        total = P["aff.total"]
        dev = dpred.device
        # out_conv: weight gradient on the generic kernel, data gradient through the 3x3 kernel on a
        # channel-padded NHWC copy of dpred
        self._set("out_conv.bias", dpred.sum((0, 2, 3)))
        Cf, Co = s.final_channels, s.out_channels
        pad_c = 32 if dt == torch.bfloat16 else 16
        dp = torch.zeros((B, dpred.shape[2], dpred.shape[3], pad_c), dtype=dt, device=dev)
        dp[..., :Co] = dpred.permute(0, 2, 3, 1).to(dt)
        if dt == torch.bfloat16:
            # the 3 image channels zero-padded to one 32-channel MFMA block: the general weight-gradient kernel
            self._set("out_conv.weight", ops.conv_wgrad(tape["hact"], dp, 3)[:Co])
        else:
            self._set("out_conv.weight", ops.conv_wgrad_edge(tape["hact"], False, dpred, True, 3))
        wT = torch.zeros((Cf, 3, 3, pad_c), dtype=dt, device=dev)
        wT[..., :Co] = P["out.w"].flip(1, 2).permute(3, 1, 2, 0).to(dt)
        dhact, gs = self._dgrad3_sums(dp, wT, tape["h_last"], tape["c_out"])
        dh, _, dg, db = ops.gn_backward(tape["h_last"], dhact, tape["c_out"], tape["h_last_part"], P["out_norm.g"],
                                        P["out_norm.b"], gn_groups(Cf), GN_EPS, act=True,
                                        dgamma=self.grads["out_norm.weight"], dbeta=self.grads["out_norm.bias"],
                                        accumulate=self._accumulating, gsum=gs)
        self._mark("out_norm.weight")
        self._mark("out_norm.bias")

        dada = torch.zeros_like(tape["ada"])
        saved: List[_Saved] = tape["saved"]
        n_enc = len(s.enc)
        # skip gradients: the decoder (walked backwards) appends d a_0, d a_1, ... for the encoder chain's
        # activations a_0 = first conv, a_i = encoder block i-1; the consumers then take them from the end
        pending: List[Tensor] = []
        blocks = P["blocks"]
        for i in range(len(blocks) - 1, -1, -1):
            e, sv = blocks[i], saved[i]
            b: BlockSpec = e["spec"]
            # the input of encoder block i>0 (and of the first decoder block) is also a decoder skip
            extra = pending.pop() if self._input_is_skip(i, n_enc) else None
            dh, dskip = self._block_bwd(e, sv, dh, extra, dada, tape["ada"], total)
            if dskip is not None:
                pending.append(dskip)
        # what is left is the skip gradient of the first conv's output
        if pending:
            dh = dh + pending.pop(0)
        assert not pending, "unconsumed skip gradients"
        if dt == torch.bfloat16:
            xin = tape["x"]
            xp = torch.zeros((B, xin.shape[2], xin.shape[3], 32), dtype=dt, device=dev)
            xp[..., :xin.shape[1]] = xin.permute(0, 2, 3, 1).to(dt)
            self._set(f"{s.enc_conv_name}.weight", ops.conv_wgrad(xp, dh, 3)[:, :xin.shape[1]])
        else:
            self._set(f"{s.enc_conv_name}.weight", ops.conv_wgrad_edge(tape["x"], True, dh, False, 3))
        self._set(f"{s.enc_conv_name}.bias", ops.bias_grad(dh))

        # embedding MLP: ada = aff(silu(pre2)), pre2 = lin1(silu(pre1)) + label(lab), pre1 = lin0(emb0)
        dpre2, gw, gb = ops.linear_backward(dada, P["aff.w"], tape["pre2"], act_in=True)
        off = 0
        for e in blocks:
            n, co2 = e["spec"].name, 2 * e["spec"].out_channels
            self._set(f"{n}.affine.weight", gw[off:off + co2])
            self._set(f"{n}.affine.bias", gb[off:off + co2])
            off += co2
        if s.label_dim:
            _, gl, _ = ops.linear_backward(dpre2, P["map_label.weight"], tape["lab"], act_in=False, need_dx=False)
            self._set("map_label.weight", gl)
        dpre1, gw, gb = ops.linear_backward(dpre2, P["map_layer1.weight"], tape["pre1"], act_in=True)
        self._set("map_layer1.weight", gw)
        self._set("map_layer1.bias", gb)
        _, gw, gb = ops.linear_backward(dpre1, P["map_layer0.weight"], tape["emb0"], act_in=False, need_dx=False)
        self._set("map_layer0.weight", gw)
        self._set("map_layer0.bias", gb)

    @staticmethod
    def _input_is_skip(i: int, n_enc: int) -> bool:
        """Block i's input is output i-1 of the encoder chain (index 0 = the first conv); every encoder
        output, the last one included, is popped as a skip by some decoder block (model.py:578-588)."""
        return 1 <= i <= n_enc

    # ------------------------------------------------------------------ step
    def forward_backward(self, x_t: Tensor, t: Tensor, class_labels: Optional[Tensor], target: Tensor,
                         accumulate: bool = False, last: bool = True) -> Tensor:
        """Loss of one micro-batch and its gradients; ``accumulate`` adds to the gradients already held
        (``accelerator.accumulate``, main.py:239).  On the ``last`` micro-batch of an optimizer step each bucket's
        all-reduce is started as soon as the backward has written the bucket (data-parallel runs only)."""
        import torch.distributed as dist
        self._drop_calls = 0
        self._micro_index = self.micro_batches if accumulate else 0      # position inside the optimizer step
        self._accumulating = bool(accumulate) and self.micro_batches > 0
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1
        self._overlap = bool(last) and (multi or self._reduce_fn is not None)
        self._pending = list(self._bucket_size)
        self._launched, self._works = [], []
        pred, tape = self.forward(x_t, t, class_labels)
        loss, dpred = ops.mse_loss_grad(pred, target.to(torch.float32))
        self.backward(tape, dpred)
        self.micro_batches = self.micro_batches + 1 if accumulate else 1
        self._accumulating = False
        return loss

    def sync_grads(self) -> None:
        """Finish the data-parallel sum of the gradients: buckets the backward did not start (no overlap requested)
        are reduced now, then every collective is awaited.  The division by the world size (and by the number of
        accumulated micro-batches) is folded into the update."""
        if self._overlap:
            assert all(p == 0 for p in self._pending), "a parameter's gradient was never written"
            for w in self._works:
                w.wait()
            self._works, self._overlap = [], False
            return
        allreduce_flat(self.gflat, self.bucket_elems, self.group)

    def ema_decay(self) -> float:
        """diffusers ``EMAModel.get_decay`` with warm-up (main.py:153-159): 1 - (1 + step/inv_gamma)^-power."""
        max_decay, inv_gamma, power = self.ema_cfg
        step = max(0, self.step_count - 1)
        if step <= 0:
            return 0.0
        return max(0.0, min(max_decay, 1.0 - (1.0 + step / inv_gamma) ** -power))

    def step(self, lr: Optional[float] = None, world_size: int = 1) -> None:
        """nan_to_num + AdamW + EMA on the mean gradient over ranks and accumulated micro-batches."""
        self.step_count += 1
        scale = 1.0 / (world_size * max(1, self.micro_batches))
        ops.adamw_step_(self.flat, self.gflat, self.m, self.v, self.ema, self.lr if lr is None else lr, self.betas[0],
                        self.betas[1], self.eps, self.weight_decay, self.step_count, grad_scale=scale,
                        ema_decay=self.ema_decay())
        self.micro_batches = 0
        self._refresh_pack()

    def train_step(self, x0: Tensor, class_labels: Optional[Tensor], alphas_cumprod: Tensor,
                   generator: Optional[torch.Generator] = None, num_timesteps: int = 1000, accumulation: int = 1,
                   warmup_steps: int = 0) -> Tensor:
        """main.py:235-257 for one optimizer step: ``accumulation`` micro-batches split from x0 (NCHW fp32 in
        [-1, 1]), gradient all-reduce, update.  Returns the mean loss (device scalar)."""
        import torch.distributed as dist
        alphas = alphas_cumprod.to(x0.device)
        total = None
        chunks = list(zip(x0.chunk(accumulation), (class_labels.chunk(accumulation) if class_labels is not None
                                                    else [None] * accumulation)))
        for i, (xs, ls) in enumerate(chunks):
            B = xs.shape[0]
            t = torch.randint(0, num_timesteps, (B,), device=x0.device, generator=generator)
            noise = torch.randn(xs.shape, device=x0.device, dtype=torch.float32, generator=generator)
            a = alphas[t].to(torch.float32)
            x_t = a.sqrt().view(B, 1, 1, 1) * xs + (1.0 - a).sqrt().view(B, 1, 1, 1) * noise     # add_noise, main.py:237
            loss = self.forward_backward(x_t, t.to(torch.float32), ls, noise, accumulate=True, last=i == len(chunks) - 1)
            total = loss if total is None else total + loss
        self.sync_grads()
        ws = dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1
        self.step(lr=warmup_lr(self.lr, self.step_count, warmup_steps), world_size=ws)
        return total / accumulation
