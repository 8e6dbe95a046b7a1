"""Kernel-level parity: each C-ABI entry point against the CPU oracle / golden vectors (GPU only)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def _rand(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


def _nhwc(x, dt, dev):
    return x.permute(0, 2, 3, 1).contiguous().to(dt).to(dev)


def _nchw(y):
    return y.float().cpu().permute(0, 3, 1, 2)


# ------------------------------------------------------------------------------ sampler
@pytest.mark.parametrize("kind", ["ddpm", "ddim"])
def test_sampler_golden(dev, kind):
    """Every sampler golden (t in {0,1,179,699,999}, 1000- and 50-step tables, masked / clipped /
    compute_eps / inversion / add_noise).  Bit-exact is expected; 2 ulp-ish slack is allowed."""
    from mad_amd.scheduler import SCHEDULER_FUNC
    g = load_golden("sampler.npz")
    s = SCHEDULER_FUNC[kind]()
    np.testing.assert_array_equal(s.alphas_cumprod.numpy(), g[f"{kind}/alphas_cumprod"])
    n_exact = n_total = 0

    def cmp(got, key, cond=1.0):
        # cond: conditioning of the output w.r.t. a 1-ulp change of a host-side coefficient.  The
        # predicted x0 = (x_t - sqrt(1-a)*eps)/sqrt(a) amplifies such a change by 1/sqrt(a) (157x at
        # t=999), and torch's CPU `** 0.5` differs by 1 ulp between CPU types, so the goldens (made
        # on another host) can only be met to that conditioning.
        nonlocal n_exact, n_total
        want = torch.from_numpy(g[key])
        got = got.cpu()
        n_total += 1
        n_exact += int(torch.equal(got, want))
        fin = torch.isfinite(want)          # "sample"-type inversion at t=0 divides by sqrt(1-1) = 0
        assert torch.equal(torch.isfinite(got), fin), key
        if fin.any():
            scale = float(want[fin].abs().max())
            assert float((got[fin] - want[fin]).abs().max()) <= 4e-7 * max(scale, 1.0) * cond, key

    for n in (1000, 50):
        s.set_timesteps(n)
        np.testing.assert_array_equal(s.timesteps.numpy(), g[f"{kind}/timesteps_{n}"])
        s.set_inverse_timesteps(n)
        np.testing.assert_array_equal(s.timesteps.numpy(), g[f"{kind}/inverse_timesteps_{n}"])
        s.set_timesteps(n)
        ratio = 1000 // n
        for t in sorted({(t // ratio) * ratio for t in (0, 1, 179, 699, 999)}):
            key = f"{kind}/n{n}/t{t}"
            T = lambda nm: torch.from_numpy(g[f"{key}/{nm}"]).to(dev)
            for pt in ("epsilon", "sample", "v_prediction"):
                s.config.prediction_type = pt
                r = s.step(T("eps"), t, T("sample"), eta=0.01, variance_noise=T("noise"), return_pred_original=True)
                cmp(r.prev_sample, f"{key}/{pt}/step")
                cmp(r.pred_original_sample, f"{key}/{pt}/x0", cond=1.0 + 1.0 / s.sqrt_alpha_pair(t)[0])
                r = s.step(T("eps"), t, T("sample"), eta=0.01, variance_noise=T("noise"), original_image=T("orig"),
                           mask=T("mask"))
                cmp(r.prev_sample, f"{key}/{pt}/step_masked")
                if kind == "ddim":
                    r = s.step(T("eps"), t, T("sample"), eta=1.0, variance_noise=T("noise"),
                               use_clipped_model_output=True)
                    cmp(r.prev_sample, f"{key}/{pt}/step_eta1_clipped")
                if t > 0:
                    cmp(s.compute_eps(T("eps"), t, T("sample"), T("next"), eta=0.01), f"{key}/{pt}/compute_eps")
                cmp(s.ddim_inverse_step(T("eps"), t, T("sample")).prev_sample, f"{key}/{pt}/inverse")
            s.config.prediction_type = "epsilon"
            cmp(s.add_noise(T("orig"), T("noise"), torch.tensor([t])), f"{key}/add_noise")
    print(f"{kind}: {n_exact}/{n_total} sampler goldens bit-exact")
    # measured on the MI355X boxes: 53/95 (DDPM), 61/113 (DDIM) identical to the last bit; the rest differ by a
    # 1-ulp host coefficient (see cmp) and stay inside its 4e-7 bound
    assert n_exact >= 0.5 * n_total


def test_sampler_errors(dev):
    from mad_amd.scheduler import CustomDDIMScheduler, CustomDDPMScheduler
    z = torch.zeros(1, 3, 4, 4, device=dev)
    d = CustomDDIMScheduler()
    with pytest.raises(ValueError):
        d.step(z, 10, z)
    p = CustomDDPMScheduler()
    p.set_timesteps(1000)
    with pytest.raises(AssertionError):
        p.compute_eps(z, 0, z, z)
    p.config.prediction_type = "bogus"
    with pytest.raises(ValueError):
        p.step(z, 10, z)
    with pytest.raises(RuntimeError):
        from mad_amd import ops
        ops.add_noise(torch.zeros(1, 3, 4, 4), torch.zeros(1, 3, 4, 4), 1.0, 0.0)   # CPU tensor: no CPU path


def test_masked_batch_and_finalize(dev):
    from mad_amd import ops
    from mad_amd.scheduler import CustomDDPMScheduler
    from oracle.scheduler_ref import RefScheduler
    s, r = CustomDDPMScheduler(), RefScheduler("ddpm")
    s.set_timesteps(1000); r.set_timesteps(1000)
    B = 3
    eps, x, z, o = (_rand((B, 3, 16, 16), i) for i in range(4))
    mask = (_rand((B, 1, 16, 16), 9) > 0).float()
    got = s.step(eps.to(dev), 500, x.to(dev), variance_noise=z.to(dev), original_image=o.to(dev), mask=mask.to(dev))
    want = r.step(eps, 500, x, variance_noise=z, original_image=o, mask=mask)
    assert float((got.prev_sample.cpu() - want.prev_sample).abs().max()) < 1e-6
    y = ops.finalize_image((x * 2).to(dev), True).cpu()
    assert torch.equal(y, (x * 2).clamp(-1, 1) / 2 + 0.5)
    dst = x.clone().to(dev)
    ops.masked_copy_(dst, o.to(dev), mask.to(dev))
    assert torch.equal(dst.cpu(), torch.where(mask.expand(-1, 3, -1, -1) > 0, o, x))


def test_keep_mask_golden(dev):
    from mad_amd import ops
    from mad_amd.translation import COMP_PRIORITY, TranslationDiffusion
    g = load_golden("masks.npz")
    lab = torch.from_numpy(g["label_map"]).to(torch.uint8).to(dev)
    basic = sorted(TranslationDiffusion.BASIC_COMPONENT, key=lambda c: COMP_PRIORITY[c])
    assert torch.equal(ops.keep_mask(lab, basic, True).cpu(), torch.from_numpy(g["keep_all"]))
    assert torch.equal(ops.keep_mask(lab, basic, False).cpu(), torch.from_numpy(g["keep_nodilate"]))
    for k in g.files:
        if k.startswith("keep_") and k[5].isdigit():
            comps = sorted([int(c) for c in k[5:].split("_")], key=lambda c: COMP_PRIORITY[c])
            assert torch.equal(ops.keep_mask(lab, comps, True).cpu(), torch.from_numpy(g[k])), k
    # a stack of maps in one launch: every row equals the single-map result (the reference map, its mirror image and
    # its transpose)
    stack = torch.stack([lab, lab.flip(1), lab.t().contiguous()]).contiguous()
    both = ops.keep_mask(stack, basic, True)
    assert tuple(both.shape) == (3, 1, 64, 64)
    for i in range(3):
        assert torch.equal(both[i], ops.keep_mask(stack[i].contiguous(), basic, True)), i
    assert torch.equal(both[0].cpu(), torch.from_numpy(g["keep_all"]))
    td = TranslationDiffusion.__new__(TranslationDiffusion)
    td.device = dev
    assert torch.equal(td.process_mask(stack, (64, 64)), both)


# ------------------------------------------------------------------------------ embedding
def test_embedding_and_linear(dev):
    from mad_amd import ops
    from oracle.unet_ref import positional_embedding
    t = torch.tensor([0.0, 1.0, 333.0, 999.0])
    half = 96
    freqs = (1.0 / 10000) ** (torch.arange(half, dtype=torch.float32) / half)
    got = ops.pos_embedding(t.to(dev), freqs.to(dev), 4).cpu()
    want = positional_embedding(t, 2 * half)
    assert float((got - want).abs().max()) < 2e-5        # sin/cos of arguments up to ~1000 rad
    got1 = ops.pos_embedding(t[2:3].to(dev), freqs.to(dev), 4).cpu()       # one timestep broadcast
    assert torch.equal(got1[0], got1[3]) and torch.equal(got1[0], got[2])
    x, W, b = _rand((5, 192), 1), _rand((300, 192), 2, 0.1), _rand((300,), 3)
    y = ops.linear_f32(x.to(dev), W.to(dev), b.to(dev), act_out=True)
    assert rel_err(y.cpu(), F.silu(x @ W.t() + b)) < 1e-5
    y2 = ops.linear_f32(x.to(dev), W.to(dev), None, out=y.clone(), accumulate=True, act_in=True)
    assert rel_err(y2.cpu(), F.silu(x @ W.t() + b) + F.silu(x) @ W.t()) < 1e-5


# ------------------------------------------------------------------------------ convolutions
CONV_CASES = [
    # B, H, C1, C2, Cout, k, up, res
    (2, 8, 64, 0, 64, 3, False, "none"),
    (3, 8, 128, 0, 128, 3, False, "same"),      # M = 192: partial last tile
    (1, 16, 64, 64, 192, 3, False, "none"),     # concat of two sources, 64-wide n tiles
    (2, 16, 128, 0, 128, 3, True, "up"),        # nearest x2 in the loader and on the residual
    (2, 16, 64, 0, 128, 3, False, "down"),      # 2x2-mean residual
    (2, 8, 128, 64, 256, 1, False, "none"),     # 1x1 over a concat
    (1, 32, 192, 0, 576, 1, False, "same"),
    (2, 64, 128, 0, 128, 3, False, "same"),     # several tiles, several rows per tile
    (16, 8, 256, 0, 256, 3, False, "same"),     # few tiles, long K: the dispatcher cuts K into shares by itself
    (3, 8, 128, 64, 128, 1, False, "same"),     # 1x1 over a concat, M = 192: a whole tile (scalar-base loader) + a partial one
    (2, 32, 192, 64, 256, 1, False, "same"),    # 1x1, eight 256-pixel tiles, two sources
]


@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("tile", ["auto", "128x128x2", "128x64x2", "64x128x3", "split3", "gemm"])
def test_conv2d(dev, case, dt, tile, monkeypatch):
    """Every block-tile variant the dispatcher can pick (small inputs alone would only reach 64x64), the
    split-K form (three shares of the K-tiles, some of them empty for the short layers), and ("gemm") the
    256-pixel-tile GEMM of the 1x1 layers forced wherever it fits."""
    from mad_amd import _lib, ops
    gemm = tile.startswith("gemm")
    if gemm:
        B, H, C1, C2, Cout, k, up, res = case
        if dt != torch.bfloat16 or k != 1 or up or res not in ("none", "same") or Cout % 128:
            pytest.skip("1x1 GEMM form: bf16, 1x1, no resampling, 128 output channels per workgroup")
        monkeypatch.setenv("MAD_CONV1X1", "2")
    elif tile == "split3":
        monkeypatch.setenv("MAD_IGEMM_KSPLIT", "3")
    elif tile != "auto":
        monkeypatch.setenv("MAD_IGEMM_TILE", tile)
    B, H, C1, C2, Cout, k, up, res = case
    Hs = H // 2 if up else H
    x1 = _rand((B, C1, Hs, Hs), 1)
    x2 = _rand((B, C2, Hs, Hs), 2) if C2 else None
    w = _rand((Cout, C1 + C2, k, k), 3, 1.0 / math.sqrt((C1 + C2) * k * k))
    b = _rand((Cout,), 4)
    rshape = {"none": None, "same": (B, Cout, H, H), "up": (B, Cout, H // 2, H // 2), "down": (B, Cout, 2 * H, 2 * H)}[res]
    r = _rand(rshape, 5) if rshape else None
    q = (lambda t: t.to(dt).float()) if dt == torch.bfloat16 else (lambda t: t)
    xin = q(x1) if x2 is None else torch.cat([q(x1), q(x2)], 1)
    if up:
        xin = xin.repeat_interleave(2, 2).repeat_interleave(2, 3)
    want = F.conv2d(xin.double(), q(w).double(), b.double(), padding=k // 2)
    if res == "same":
        want = want + q(r).double()
    elif res == "up":
        want = want + q(r).double().repeat_interleave(2, 2).repeat_interleave(2, 3)
    elif res == "down":
        want = want + F.avg_pool2d(q(r).double(), 2)
    mode = {"none": _lib.RES_NONE, "same": _lib.RES_SAME, "up": _lib.RES_UP, "down": _lib.RES_DOWN}[res]
    y = ops.conv2d(_nhwc(x1, dt, dev), w.permute(0, 2, 3, 1).contiguous().to(dt).to(dev), b.to(dev),
                   x2=None if x2 is None else _nhwc(x2, dt, dev),
                   residual=None if r is None else _nhwc(r, dt, dev), res_mode=mode, upsample=up)
    assert tuple(y.shape) == (B, H, H, Cout)
    tol = 2e-5 if dt == torch.float32 else 6e-3       # bf16: inputs are pre-rounded, only the output rounds
    assert rel_err(_nchw(y), want) < tol
    # the same launch with the epilogue's GroupNorm partial sums: identical output, sums of what was stored
    y2, stats = ops.conv2d(_nhwc(x1, dt, dev), w.permute(0, 2, 3, 1).contiguous().to(dt).to(dev), b.to(dev),
                           x2=None if x2 is None else _nhwc(x2, dt, dev),
                           residual=None if r is None else _nhwc(r, dt, dev), res_mode=mode, upsample=up,
                           want_stats=True)
    assert torch.equal(y2, y) and tuple(stats.shape) == (B, H * H // 64, Cout, 2)
    runs = y.float().reshape(B, H * H // 64, 64, Cout)
    # (the bf16 GEMM form keeps the sums of a channel pair in its even slot: a GroupNorm group holds whole pairs)
    cpair = (lambda t: t.reshape(*t.shape[:-1], Cout // 2, 2).sum(-1)) if gemm else (lambda t: t)
    assert rel_err(cpair(stats[..., 0].cpu()), cpair(runs.sum(2).cpu())) < 1e-4
    assert rel_err(cpair(stats[..., 1].cpu()), cpair((runs * runs).sum(2).cpu())) < 1e-4


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_conv_in_out(dev, dt):
    from mad_amd import ops
    # 16x32 and 24x16 take the matrix-pipe kernel (several tiles per image), 12x12 the direct one
    for shape, cout in [((2, 3, 16, 32), 64), ((3, 3, 24, 16), 128), ((2, 3, 12, 12), 64), ((1, 3, 32, 32), 192)]:
        x = _rand(shape, 1)
        w, b = _rand((cout, 3, 3, 3), 2, 0.2), _rand((cout,), 3)
        y = ops.conv_in(x.to(dev), w.to(dev), b.to(dev), dt)
        assert rel_err(_nchw(y), F.conv2d(x, w, b, padding=1)) < (1e-5 if dt == torch.float32 else 6e-3), (shape, cout)
        # the same layer with the epilogue's GroupNorm partial sums (per 8x16 tile) where the tiled kernel runs
        y2, part = ops.conv_in(x.to(dev), w.to(dev), b.to(dev), dt, want_stats=True)
        assert torch.equal(y2, y)
        B_, _, H_, W_ = shape
        if H_ % 8 == 0 and W_ % 16 == 0:
            tiles = y.float().reshape(B_, H_ // 8, 8, W_ // 16, 16, cout).permute(0, 1, 3, 2, 4, 5).reshape(B_, -1, 128, cout)
            assert tuple(part.shape) == (B_, (H_ // 8) * (W_ // 16), cout, 2)
            assert rel_err(part[..., 0].cpu(), tiles.sum(2).cpu()) < 1e-4
            assert rel_err(part[..., 1].cpu(), (tiles * tiles).sum(2).cpu()) < 1e-4
        else:
            assert part is None
    h = _rand((2, 64, 16, 16), 4)
    wo, bo = _rand((3, 64, 3, 3), 5, 0.05), _rand((3,), 6)
    hq = h.to(dt).float()
    yo = ops.conv_out(_nhwc(h, dt, dev), wo.permute(0, 2, 3, 1).contiguous().to(dev), bo.to(dev))
    assert rel_err(yo.cpu(), F.conv2d(hq, wo, bo, padding=1)) < 1e-5
    # the MFMA form of the last layer, with norm + SiLU folded into its loader
    coef = torch.stack([1 + 0.3 * _rand((2, 64), 7), 0.3 * _rand((2, 64), 8)], dim=-1)
    w32 = torch.zeros(32, 3, 3, 64)
    w32[:3] = wo.permute(0, 2, 3, 1)
    hin = F.silu(hq.double() * coef[..., 0].double()[:, :, None, None] + coef[..., 1].double()[:, :, None, None])
    hin = hin.float().to(dt).double()
    want = F.conv2d(hin, w32[:3].permute(0, 3, 1, 2).to(dt).double(), bo.double(), padding=1)
    yf = ops.conv_out_fused(_nhwc(h, dt, dev), w32.to(dt).to(dev), bo.to(dev), 3, coef=coef.contiguous().to(dev))
    assert tuple(yf.shape) == (2, 3, 16, 16)
    assert rel_err(yf.cpu(), want) < 2e-5


# ------------------------------------------------------------------------------ GroupNorm
GN_CASES = [
    # B, H, C1, C2, ada, act, pool
    (2, 16, 64, 0, False, True, False),        # 16 groups of 4
    (2, 16, 128, 64, False, True, False),      # 192 channels, groups of 6 straddle the two sources
    (3, 8, 256, 0, True, True, False),         # AdaGN modulation
    (2, 16, 128, 0, False, True, True),        # fused 2x2 mean (encoder downsample)
    (1, 32, 512, 256, False, False, False),    # no activation (norm2), 24 channels per group
    (2, 8, 1536, 0, True, True, False),        # widest layer of the 64x64 model
]


@pytest.mark.parametrize("case", GN_CASES)
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_groupnorm(dev, case, dt):
    from mad_amd import ops
    B, H, C1, C2, use_ada, act, pool = case
    C = C1 + C2
    x1, x2 = _rand((B, C1, H, H), 1, 1.5) + 0.3, (_rand((B, C2, H, H), 2, 0.7) - 0.2 if C2 else None)
    gamma, beta = 1 + 0.2 * _rand((C,), 3), 0.2 * _rand((C,), 4)
    groups = min(32, C // 4)
    q = (lambda t: t.to(dt).float()) if dt == torch.bfloat16 else (lambda t: t)
    xin = q(x1) if x2 is None else torch.cat([q(x1), q(x2)], 1)
    want = F.group_norm(xin.double(), groups, gamma.double(), beta.double(), 1e-5)
    ada = None
    if use_ada:
        ada_full = _rand((B, 7 + 2 * C), 5, 0.3)             # an offset view into a wider table
        scale, shift = ada_full[:, 7:7 + C], ada_full[:, 7 + C:]
        want = shift.double()[:, :, None, None] + want * (scale.double()[:, :, None, None] + 1)
        ada_dev = ada_full.to(dev)
        ada = ada_dev[:, 7:]
    if act:
        want = F.silu(want)
    if pool:
        want = F.avg_pool2d(want, 2)
    a1 = _nhwc(x1, dt, dev)
    a2 = None if x2 is None else _nhwc(x2, dt, dev)
    coef = ops.gn_coef(a1, gamma.to(dev), beta.to(dev), groups, 1e-5, x2=a2, ada=ada,
                       ada_stride=(7 + 2 * C) if use_ada else 0)
    y = ops.gn_apply(a1, coef, act=act, x2=a2, pool=pool)
    assert rel_err(_nchw(y), want) < (2e-5 if dt == torch.float32 else 6e-3)


# ------------------------------------------------------------------------------ attention
@pytest.mark.parametrize("B,L,heads", [(2, 64, 1), (1, 256, 2), (2, 1024, 4), (3, 64, 12), (8, 256, 3)])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("qb", ["1", "2"])
def test_attention(dev, B, L, heads, dt, qb, monkeypatch):
    """qb: one or two 32-query blocks per wave of the bf16 kernel (two needs L a multiple of 256); (8, 256, 3) has a
    multiple of eight (sample, head) pairs, the case whose grid is mapped XCD by XCD."""
    from mad_amd import ops
    if qb == "2" and (dt != torch.bfloat16 or L % 256):
        pytest.skip("two query blocks per wave: bf16, L a multiple of 256")
    monkeypatch.setenv("MAD_ATTN_QB", qb)
    side = int(math.isqrt(L))
    q, k, v = (_rand((B, heads, L, 64), s, 1.3) for s in (1, 2, 3))
    k[0, 0, 5] *= 6.0                                        # one dominant key: forces max rescaling mid-loop
    qq = (lambda t: t.to(dt).float()) if dt == torch.bfloat16 else (lambda t: t)
    w = torch.softmax(qq(q).double() @ qq(k).double().transpose(-1, -2) / 8.0, dim=-1)
    want = (w @ qq(v).double()).permute(0, 2, 1, 3).reshape(B, L, heads * 64)
    qkv = torch.cat([q, k, v], dim=-1).permute(0, 2, 1, 3).reshape(B, side, side, heads * 192)
    out = ops.attention(qkv.contiguous().to(dt).to(dev), heads)
    assert rel_err(out.float().cpu().reshape(B, L, heads * 64), want) < (2e-5 if dt == torch.float32 else 1.2e-2)


# ------------------------------------------------------------------------------ fused halo conv
FUSED_CASES = [
    # B, H, C1, C2, Cout, fuse, up, res, stats
    (2, 16, 64, 0, 128, True, False, "none", True),
    (1, 32, 128, 0, 128, True, False, "same", True),
    (2, 32, 64, 64, 256, True, False, "none", False),     # concat, two n tiles
    (1, 32, 128, 0, 128, True, True, "up", True),         # nearest x2 of a 16x16 source
    (1, 32, 64, 0, 128, False, False, "down", False),     # plain input, 2x2-mean residual
    (1, 64, 160, 0, 128, True, False, "same", True),      # 5 bf16 chunks / 10 fp32 chunks
    (1, 32, 96, 0, 192, True, False, "same", True),       # Cout an odd multiple of 64
    (2, 32, 64, 32, 320, True, False, "same", True),      # 2 x 128 + 64 output channels, two sources
    (1, 32, 64, 0, 256, True, False, "same", True),       # one 256-channel workgroup column ("256")
    (1, 32, 96, 32, 512, True, True, "up", True),         # two of them, two sources, nearest x2
    (2, 16, 128, 0, 128, False, False, "same", True),     # plain input, one tile per image
    (1, 32, 64, 64, 192, False, True, "up", True),        # plain, two sources, nearest x2, odd multiple of 64 channels
]


@pytest.mark.parametrize("case", FUSED_CASES)
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("bn", ["128", "64", "192", "128w", "192w", "256", "ks2", "ks2n"])
@pytest.mark.parametrize("fmt", ["plain", "tiled", "tiled+pre"])
def test_conv3x3_fused(dev, case, dt, bn, fmt, monkeypatch):
    """Both workgroup widths (128 and 64 output channels) of the halo-tile kernel, ("192") the split of an
    odd multiple of 64 channels over a wide and a narrow launch, ("w") the 16x32-pixel eight-wave tiles
    forced wherever they fit (bf16, width a multiple of 32, 128-channel workgroups), and ("256") the eight-wave
    256-channel workgroups of the prescaled loader-transform form forced likewise.  fmt: the weights as
    (Cout,3,3,Cin), tile by tile, and tile by tile with the loader transform in its prescaled form (coefficients
    times -log2(e), weights times -ln(2)): the same convolution every time.  ("ks2" / "ks2n") the eight-wave form
    whose two sets of waves split the channel chunks, at 128 / 64 output channels per workgroup, forced wherever it
    fits (bf16; plain input or the prescaled transform at 128 channels; whole pairs of chunks); every other value
    of bn runs with that form switched off, so that the small shapes here still reach the four-wave forms."""
    from mad_amd import _lib, ops
    B, H, C1, C2, Cout, fuse, up, res, want_stats = case
    monkeypatch.setenv("MAD_HALO_KS", "2" if bn.startswith("ks2") else "0")
    if bn.startswith("ks2"):
        if dt != torch.bfloat16 or (C1 + C2) % 64 or (fuse and (fmt != "tiled+pre" or bn == "ks2n")):
            pytest.skip("two K shares: bf16, whole chunk pairs, plain input or the prescaled transform at 128 channels")
        if bn == "ks2" and Cout % 128:
            pytest.skip("128-channel workgroups")
        bn = "128" if bn == "ks2" else "64"
    wide = bn.endswith("w")
    bn = bn.rstrip("w")
    monkeypatch.setenv("MAD_HALO_BN256", "2" if bn == "256" else "0")
    if bn == "256":
        if dt != torch.bfloat16 or fmt != "tiled+pre" or not fuse or Cout % 256:
            pytest.skip("256-channel workgroups: bf16, prescaled loader transform, whole 256-channel columns")
        bn = "128"
    monkeypatch.setenv("MAD_HALO_BN", bn)
    monkeypatch.setenv("MAD_HALO_WIDE", "2" if wide else "0")
    if bn == "192" and Cout % 128 != 64:
        pytest.skip("the split applies to odd multiples of 64 channels")
    if wide and (dt != torch.bfloat16 or H % 32):
        pytest.skip("16x32 tiles: bf16, width a multiple of 32")
    C = C1 + C2
    Hs = H // 2 if up else H
    x1 = _rand((B, C1, Hs, Hs), 1, 1.3) + 0.2
    x2 = _rand((B, C2, Hs, Hs), 2) if C2 else None
    w = _rand((Cout, C, 3, 3), 3, 1.0 / math.sqrt(C * 9))
    bias = _rand((Cout,), 4)
    coef = torch.stack([1 + 0.3 * _rand((B, C), 5), 0.3 * _rand((B, C), 6)], dim=-1) if fuse else None
    rshape = {"none": None, "same": (B, Cout, H, H), "up": (B, Cout, H // 2, H // 2), "down": (B, Cout, 2 * H, 2 * H)}[res]
    r = _rand(rshape, 7) if rshape else None
    q = (lambda t: t.to(dt).float()) if dt == torch.bfloat16 else (lambda t: t)
    xin = q(x1) if x2 is None else torch.cat([q(x1), q(x2)], 1)
    xin = xin.double()
    if fuse:
        xin = F.silu(xin * coef[..., 0].double()[:, :, None, None] + coef[..., 1].double()[:, :, None, None])
        xin = q(xin.float()).double()                      # the loader rounds the activated value to the storage type
    if up:
        xin = xin.repeat_interleave(2, 2).repeat_interleave(2, 3)
    want = F.conv2d(xin, q(w).double(), bias.double(), padding=1)
    if res == "same":
        want = want + q(r).double()
    elif res == "up":
        want = want + q(r).double().repeat_interleave(2, 2).repeat_interleave(2, 3)
    elif res == "down":
        want = want + F.avg_pool2d(q(r).double(), 2)
    mode = {"none": _lib.RES_NONE, "same": _lib.RES_SAME, "up": _lib.RES_UP, "down": _lib.RES_DOWN}[res]
    pre = fmt == "tiled+pre" and fuse
    wk = w.permute(0, 2, 3, 1).contiguous().to(dev)
    wk = ops.halo_tile_weights(wk, dt, _lib.PRESCALE_W if pre else 1.0) if fmt != "plain" else wk.to(dt)
    ck = None if coef is None else (coef * _lib.PRESCALE_COEF if pre else coef).contiguous().to(dev)
    out = ops.conv3x3_fused(_nhwc(x1, dt, dev), wk, bias.to(dev), coef=ck, act=True,
                            x2=None if x2 is None else _nhwc(x2, dt, dev),
                            residual=None if r is None else _nhwc(r, dt, dev), res_mode=mode, upsample=up,
                            want_stats=want_stats, prescaled=pre, w_tiled=fmt != "plain")
    y, stats = out if want_stats else (out, None)
    tol = 3e-5 if dt == torch.float32 else 8e-3
    assert rel_err(_nchw(y), want) < tol
    if want_stats:
        yy = _nchw(y).double()                            # statistics are taken over the stored values
        tiles = yy.reshape(B, Cout, H // 16, 16, H // 16, 16).permute(0, 2, 4, 1, 3, 5).reshape(B, -1, Cout, 256)
        s0, s1, st = tiles.sum(-1), (tiles ** 2).sum(-1), stats.cpu()
        if wide:      # a 16x32 tile writes its sums to the row of its left half and zeros to the right half's
            pair = lambda t: t.reshape(B, H // 16, H // 32, 2, Cout).sum(3)
            s0, s1, st = pair(s0), pair(s1), st.reshape(B, H // 16, H // 32, 2, Cout, 2).sum(3)
        # bf16 launches hold the sums of a channel pair in its even slot (a GroupNorm group holds whole pairs)
        cpair = lambda t: t.reshape(*t.shape[:-1], Cout // 2, 2).sum(-1)
        assert rel_err(cpair(st[..., 0]), cpair(s0)) < 1e-4
        assert rel_err(cpair(st[..., 1]), cpair(s1)) < 1e-4
        if dt == torch.float32:
            assert rel_err(st[..., 0], s0) < 1e-4 and rel_err(st[..., 1], s1) < 1e-4
        # and they feed the finalize kernel exactly like the partial-sum kernel's output does
        gamma, beta = 1 + 0.2 * _rand((Cout,), 8), 0.2 * _rand((Cout,), 9)
        c1 = ops.gn_finalize(stats, H * H, gamma.to(dev), beta.to(dev), 32, 1e-5)
        c2 = ops.gn_coef(y, gamma.to(dev), beta.to(dev), 32, 1e-5)
        assert rel_err(c1.cpu(), c2.cpu()) < 1e-4
        # the scaled form writes the same pairs times the factor
        c3 = ops.gn_finalize(stats, H * H, gamma.to(dev), beta.to(dev), 32, 1e-5, scale=_lib.PRESCALE_COEF)
        assert torch.equal(c3, c1 * torch.tensor(_lib.PRESCALE_COEF, dtype=torch.float32))


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("bn", ["128", "64", "128w", "256", "ks2", "ks2n"])
@pytest.mark.parametrize("case", [(2, 32, 64, 64, 32, 128), (1, 16, 128, 96, 0, 192), (1, 64, 64, 32, 64, 256),
                                  (2, 32, 128, 64, 64, 128), (2, 16, 192, 128, 0, 192)])
@pytest.mark.parametrize("fmt", ["plain", "tiled+pre", "nocoef"])
def test_conv3x3_fused_projection(dev, case, dt, bn, fmt, monkeypatch):
    """conv1 + the residual branch's 1x1 projection of the (two-source) block input in one launch (fmt and bn as in
    test_conv3x3_fused; the projection's weights are never scaled; "nocoef": a plain conv input, as the blocks
    below the loader transform's resolution pass it)."""
    from mad_amd import _lib, ops
    B, H, C, S1, S2, Cout = case
    monkeypatch.setenv("MAD_HALO_KS", "2" if bn.startswith("ks2") else "0")
    if bn.startswith("ks2"):
        if dt != torch.bfloat16 or C % 64 or (S1 + S2) % 64 or fmt == "plain" or (fmt == "tiled+pre" and bn == "ks2n"):
            pytest.skip("two K shares: bf16, whole chunk pairs, plain input or the prescaled transform at 128 channels")
        if bn == "ks2" and Cout % 128:
            pytest.skip("128-channel workgroups")
        bn = "128" if bn == "ks2" else "64"
    monkeypatch.setenv("MAD_HALO_WIDE", "2" if bn.endswith("w") else "0")
    bn = bn.rstrip("w")
    monkeypatch.setenv("MAD_HALO_BN256", "2" if bn == "256" else "0")
    if bn == "256":
        if dt != torch.bfloat16 or fmt != "tiled+pre" or Cout % 256:
            pytest.skip("256-channel workgroups: bf16, prescaled loader transform, whole 256-channel columns")
        bn = "128"
    monkeypatch.setenv("MAD_HALO_BN", bn)
    h = _rand((B, C, H, H), 1, 1.2)
    s1, s2 = _rand((B, S1, H, H), 2), (_rand((B, S2, H, H), 3) if S2 else None)
    w = _rand((Cout, C, 3, 3), 4, 1.0 / math.sqrt(C * 9))
    sw = _rand((Cout, S1 + S2, 1, 1), 5, 1.0 / math.sqrt(S1 + S2))
    bias = _rand((Cout,), 6)
    coef = torch.stack([1 + 0.3 * _rand((B, C), 7), 0.3 * _rand((B, C), 8)], dim=-1)
    q = (lambda t: t.to(dt).float()) if dt == torch.bfloat16 else (lambda t: t)
    nocoef = fmt == "nocoef"
    hin = q(h) if nocoef else F.silu(q(h) * coef[:, :, 0, None, None] + coef[:, :, 1, None, None])
    if dt == torch.bfloat16:
        hin = q(hin)                                           # the loader rounds the activated value
    sin = q(s1) if s2 is None else torch.cat([q(s1), q(s2)], 1)
    want = F.conv2d(hin.double(), q(w).double(), bias.double(), padding=1) + F.conv2d(sin.double(), q(sw).double())
    pre = fmt == "tiled+pre"
    wk = w.permute(0, 2, 3, 1).contiguous().to(dev)
    wk = ops.halo_tile_weights(wk, dt, _lib.PRESCALE_W) if pre else wk.to(dt)
    y, stats = ops.conv3x3_fused(_nhwc(h, dt, dev), wk, bias.to(dev),
                                 coef=None if nocoef else (coef * _lib.PRESCALE_COEF if pre else coef).to(dev),
                                 act=True, want_stats=True,
                                 skip=(_nhwc(s1, dt, dev), None if s2 is None else _nhwc(s2, dt, dev),
                                       sw.reshape(Cout, -1).to(dt).to(dev)), prescaled=pre, w_tiled=pre)
    assert rel_err(_nchw(y), want) < (2e-5 if dt == torch.float32 else 8e-3)
    yf = y.float()
    got = stats.sum(1).cpu()
    cpair = lambda t: t.reshape(*t.shape[:-1], Cout // 2, 2).sum(-1)      # bf16: sums per channel pair
    assert rel_err(cpair(got[..., 0]), cpair(yf.sum((1, 2)).cpu())) < 1e-4
    assert rel_err(cpair(got[..., 1]), cpair((yf * yf).sum((1, 2)).cpu())) < 1e-4


@pytest.mark.parametrize("shape", [(2, 64, 16, 32, 3), (1, 192, 32, 16, 2), (3, 32, 16, 16, 1), (2, 128, 48, 32, 3)])
@pytest.mark.parametrize("fuse", [True, False])
def test_conv_out_tap_major(dev, shape, fuse, monkeypatch):
    """Last layer, bf16: the tap-major kernel (nine taps as the N dimension of one product per halo pixel) against
    fp64 on the rounded operands and against the halo-tile form of the same entry point; image borders, several
    tiles per image, 1-3 output channels, with and without the folded norm + SiLU."""
    from mad_amd import ops
    B, Cin, H, W_, Cout = shape
    dt = torch.bfloat16
    h = _rand((B, Cin, H, W_), 11)
    wo, bo = _rand((Cout, Cin, 3, 3), 12, 0.05), _rand((Cout,), 13)
    hq = h.to(dt).float()
    w32 = torch.zeros(32, 3, 3, Cin)
    w32[:Cout] = wo.permute(0, 2, 3, 1)
    coef = torch.stack([1 + 0.3 * _rand((B, Cin), 14), 0.3 * _rand((B, Cin), 15)], dim=-1).contiguous() if fuse else None
    hin = hq.double()
    if fuse:
        hin = F.silu(hin * coef[..., 0].double()[:, :, None, None] + coef[..., 1].double()[:, :, None, None])
        hin = hin.float().to(dt).double()
    want = F.conv2d(hin, w32[:Cout].permute(0, 3, 1, 2).to(dt).double(), bo.double(), padding=1)
    args = (_nhwc(h, dt, dev), w32.to(dt).to(dev), bo.to(dev), Cout)
    kw = dict(coef=None if coef is None else coef.to(dev), act=True)
    y = ops.conv_out_fused(*args, **kw)
    assert tuple(y.shape) == (B, Cout, H, W_)
    # the transform's exp / reciprocal are 1-ulp instructions and its result rounds to bf16: an element of the
    # normalised input can land on the neighbouring bf16 value
    assert rel_err(y.cpu(), want) < (3e-3 if fuse else 2e-5)
    monkeypatch.setenv("MAD_CONV_OUT_HALO", "1")
    y_halo = ops.conv_out_fused(*args, **kw)
    assert rel_err(y.cpu(), y_halo.cpu().double()) < (3e-3 if fuse else 2e-5)


IMG8_CASES = [
    # B, C1, C2, Cout, fuse, res
    (3, 64, 0, 64, False, "none"),        # one K-tile (bf16) / two (fp32) per tap, a single slab
    (2, 512, 512, 128, True, "same"),     # two full bf16 slabs from two sources
    (2, 192, 128, 192, True, "none"),     # sources meet inside a slab; ragged last slab in fp32
    (5, 768, 768, 64, True, "same"),      # three slabs, second source starts mid-slab
    (2, 320, 0, 128, False, "down"),      # residual averaged from 16x16
    (1, 576, 0, 64, True, "down"),        # second slab of one K-tile (bf16): ring restarts with nk = 9
]


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", IMG8_CASES)
def test_conv3x3_img8(dev, case, dt):
    """Image-resident 8x8 conv against an fp64 reference; statistics against the stored output."""
    from mad_amd import _lib, ops
    B, C1, C2, Cout, fuse, res = case
    C = C1 + C2
    x = _rand((B, C, 8, 8), 1, 1.2)
    w = _rand((Cout, C, 3, 3), 2, 1.0 / math.sqrt(C * 9))
    bias = _rand((Cout,), 3)
    coef = torch.stack([1 + 0.3 * _rand((B, C), 4), 0.3 * _rand((B, C), 5)], dim=-1) if fuse else None
    q = (lambda t: t.to(dt).float()) if dt == torch.bfloat16 else (lambda t: t)
    hin = q(x)
    if fuse:
        hin = q(F.silu(hin * coef[:, :, 0, None, None] + coef[:, :, 1, None, None]))
    want = F.conv2d(hin.double(), q(w).double(), bias.double(), padding=1)
    r, mode = None, _lib.RES_NONE
    if res == "same":
        r, mode = _rand((B, Cout, 8, 8), 6), _lib.RES_SAME
        want = want + q(r).double()
    elif res == "down":
        r, mode = _rand((B, Cout, 16, 16), 6), _lib.RES_DOWN
        want = want + F.avg_pool2d(q(r).double(), 2)
    xn = _nhwc(x, dt, dev)
    y, stats = ops.conv3x3_img8(xn[..., :C1].contiguous(), w.permute(0, 2, 3, 1).contiguous().to(dt).to(dev),
                                bias.to(dev), coef=None if coef is None else coef.to(dev), act=True,
                                x2=xn[..., C1:].contiguous() if C2 else None,
                                residual=None if r is None else _nhwc(r, dt, dev), res_mode=mode, want_stats=True)
    assert rel_err(_nchw(y), want) < (2e-5 if dt == torch.float32 else 8e-3)
    yf = y.float()
    got = stats.sum(1).cpu()
    assert tuple(stats.shape) == (B, 1, Cout, 2)
    assert rel_err(got[..., 0], yf.sum((1, 2)).cpu()) < 1e-4
    assert rel_err(got[..., 1], (yf * yf).sum((1, 2)).cpu()) < 1e-4
    # same values as the implicit-GEMM path on the same (already activated) input, up to summation order
    hin_n = _nhwc(hin, dt, dev)
    y2 = ops.conv2d(hin_n, w.permute(0, 2, 3, 1).contiguous().to(dt).to(dev), bias.to(dev),
                    residual=None if r is None else _nhwc(r, dt, dev), res_mode=mode)
    assert rel_err(y.float(), y2.float()) < (2e-5 if dt == torch.float32 else 6e-3)
