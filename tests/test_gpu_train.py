"""The training step (forward with tape, hand-written backward, AdamW + EMA) against autograd through the
CPU restatement of the reference's denoiser (GPU only)."""
import os

import pytest
import torch
import torch.nn.functional as F

from conftest import REPO, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _grad_on():
    with torch.enable_grad():
        yield


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _setup(cfg_file, precision, dev, B, seed=3):
    from mad_amd.config import load_cfg
    from mad_amd.model import build_model
    from mad_amd.synthetic import synthetic_faces
    cfg = load_cfg(os.path.join(REPO, "configs", cfg_file))
    m = build_model(cfg, precision=precision)
    m.unet.init_seeded(seed)
    sd = {k: v.detach().clone().float() for k, v in m.unet.state_dict().items()}
    m = m.to(dev).train()
    R = m.unet.spec.img_resolution
    g = torch.Generator().manual_seed(11)
    x = synthetic_faces(B, R, seed=5) * 0.7 + 0.5 * torch.randn(B, 3, R, R, generator=g)
    t = torch.tensor([37.0, 611.0, 250.0, 900.0][:B])
    lab = torch.eye(2)[torch.arange(B) % 2]
    target = torch.randn(B, 3, R, R, generator=g)
    return cfg, m, sd, x, t, lab, target


def _oracle_grads(cfg, sd, x, t, lab, target):
    from oracle import unet_ref
    from mad_amd.unet_spec import spec_from_cfg
    s = spec_from_cfg(cfg)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if "resample_filter" not in k}
    full = dict(sd)
    full.update(params)
    pred = unet_ref.unet_forward(full, x, t, lab, img_resolution=s.img_resolution, model_channels=s.model_channels,
                                 channel_mult=s.channel_mult, num_blocks=s.num_blocks, label_dim=s.label_dim)
    loss = F.mse_loss(pred, target)
    loss.backward()
    return float(loss.detach()), pred.detach(), {k: v.grad for k, v in params.items()}


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_train_gradients_match_autograd(dev, precision):
    from mad_amd.train import UNetTrainer
    cfg, m, sd, x, t, lab, target = _setup("model_tiny.yaml", precision, dev, B=3)
    want_loss, want_pred, want = _oracle_grads(cfg, sd, x, t, lab, target)
    tr = UNetTrainer(m.unet, dropout=0.0)
    pred, tape = tr.forward(x.to(dev), t.to(dev), lab.to(dev))
    tol_out = 1e-4 if precision == "fp32" else 3e-2
    assert rel_err(pred.cpu(), want_pred) < tol_out
    loss = tr.forward_backward(x.to(dev), t.to(dev), lab.to(dev), target.to(dev))
    assert abs(float(loss) - want_loss) < (1e-4 if precision == "fp32" else 2e-2) * want_loss
    tol = 2e-3 if precision == "fp32" else 6e-2
    worst = ("", 0.0)
    assert set(tr.grads) == set(want)
    for k, g in want.items():
        assert g is not None, k
        err = rel_err(tr.grads[k].cpu(), g)
        if err > worst[1]:
            worst = (k, err)
    print(f"{precision}: worst gradient rel err {worst[1]:.3e} at {worst[0]}")
    assert worst[1] < tol, worst


@pytest.mark.parametrize("cfg_file", ["model_64_64.yaml", "model_256_256.yaml"])
def test_train_gradients_real_configs(dev, cfg_file):
    """The real configurations, one image, bf16 activations: loss and per-tensor gradient direction against
    fp32 autograd through the oracle (cosine similarity: bf16 rounding through ~40 blocks leaves a few per
    cent of noise on individual elements)."""
    from mad_amd.train import UNetTrainer
    cfg, m, sd, x, t, lab, target = _setup(cfg_file, "bf16", dev, B=1)
    want_loss, want_pred, want = _oracle_grads(cfg, sd, x, t, lab, target)
    tr = UNetTrainer(m.unet, dropout=0.0)
    loss = tr.forward_backward(x.to(dev), t.to(dev), lab.to(dev), target.to(dev))
    assert abs(float(loss) - want_loss) < 3e-2 * want_loss
    worst = ("", 1.0)
    for k, g in want.items():
        got = tr.grads[k].cpu().double().reshape(-1)
        ref = g.double().reshape(-1)
        assert torch.isfinite(got).all(), k
        cos = float((got @ ref) / (got.norm() * ref.norm()).clamp_min(1e-30))
        ratio = float(got.norm() / ref.norm().clamp_min(1e-30))
        if cos < worst[1]:
            worst = (k, cos)
        assert 0.9 < ratio < 1.1, (k, ratio)
    print(f"{cfg_file}: worst gradient cosine {worst[1]:.5f} at {worst[0]}")
    assert worst[1] > 0.99, worst


def test_adamw_ema_step_matches_torch(dev):
    """Flat fused update against torch.optim.AdamW (+ nan_to_num and the EMA rule) on random buffers."""
    from mad_amd import ops
    g = torch.Generator().manual_seed(1)
    n = 100003
    p0, grad = torch.randn(n, generator=g), torch.randn(n, generator=g) * 0.1
    grad[5], grad[6], grad[7] = float("nan"), float("inf"), float("-inf")
    ref = torch.nn.Parameter(p0.clone().double())
    opt = torch.optim.AdamW([ref], lr=1e-3, betas=(0.95, 0.999), eps=1e-7, weight_decay=1e-2)
    p, m, v, ema = p0.clone().to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev), p0.clone().to(dev)
    ema_ref = p0.clone().double()
    for step in range(1, 4):
        gd = torch.nan_to_num(grad.double() * 0.5, nan=0.0, posinf=1e5, neginf=-1e5) * step
        ref.grad = gd.clone()
        opt.step()
        decay = 0.9 if step > 1 else 0.0
        ema_ref -= (1 - decay) * (ema_ref - ref.detach())
        ops.adamw_step_(p, (grad * step).to(dev), m, v, ema, 1e-3, 0.95, 0.999, 1e-7, 1e-2, step, grad_scale=0.5,
                        ema_decay=decay)
    assert rel_err(p.cpu(), ref.detach()) < 1e-4
    assert rel_err(ema.cpu(), ema_ref) < 1e-4


def test_training_reduces_loss(dev):
    """Full steps (dropout on) on one fixed noised batch: the loss goes down and the EMA follows."""
    from mad_amd.train import UNetTrainer
    cfg, m, sd, x0, _, lab, noise = _setup("model_tiny.yaml", "bf16", dev, B=4)
    t = torch.tensor([500.0, 700.0, 900.0, 999.0])
    betas = torch.linspace(1e-4, 2e-2, 1000)
    a = torch.cumprod(1 - betas, 0)[t.long()].view(4, 1, 1, 1)
    x_t = a.sqrt() * x0.clamp(-1, 1) + (1 - a).sqrt() * noise
    tr = UNetTrainer(m.unet, lr=5e-4, dropout=0.1)
    losses = []
    for _ in range(30):
        losses.append(float(tr.forward_backward(x_t.to(dev), t.to(dev), lab.to(dev), noise.to(dev))))
        tr.step()
    print("losses", [round(v, 4) for v in losses])
    assert min(losses[-5:]) < 0.8 * losses[0]
    assert torch.isfinite(tr.flat).all() and torch.isfinite(tr.ema).all()
    assert float((tr.ema - tr.flat).abs().max()) > 0


def test_accumulated_micro_batches_draw_their_own_dropout_masks(dev):
    """The same micro-batch twice inside one optimizer step: with dropout on the two passes must see different
    masks (the reference draws a fresh mask per forward, main.py:239-246), so their losses differ; with the
    position inside the step held fixed the pass is reproducible to the last bit."""
    from mad_amd.train import UNetTrainer
    cfg, m, sd, x, t, lab, target = _setup("model_tiny.yaml", "fp32", dev, B=2)
    tr = UNetTrainer(m.unet, lr=1e-4, dropout=0.3)
    args = (x.to(dev), t.to(dev), lab.to(dev), target.to(dev))
    first = float(tr.forward_backward(*args))
    again = float(tr.forward_backward(*args))                    # not accumulating: position 0 both times
    second = float(tr.forward_backward(*args, accumulate=True))  # position 1 of the same step
    assert again == first
    assert abs(second - first) > 1e-6 * abs(first)


@pytest.mark.parametrize("cfg_file,precision", [("model_tiny.yaml", "fp32"), ("model_tiny.yaml", "bf16"),
                                                ("model_64_64.yaml", "bf16")])
def test_refreshed_working_copies_equal_a_fresh_pack(dev, cfg_file, precision):
    """After an update the working copies of the weights are rewritten in place (one kernel for the convs'
    forward and data-gradient layouts, a few torch ops for the regrouped tensors): every entry must equal
    what packing the updated masters from scratch gives."""
    from mad_amd.train import UNetTrainer, _dgrad_weight
    cfg, m, sd, x, t, lab, target = _setup(cfg_file, precision, dev, B=2)
    tr = UNetTrainer(m.unet, lr=1e-2, dropout=0.0)
    assert tr.REPACK
    for _ in range(2):
        tr.forward_backward(x[:2].to(dev), t[:2].to(dev), lab[:2].to(dev), target[:2].to(dev))
        tr.step()
    P = m.unet._packed
    assert P is not None and tr._arena_of is P
    got_blocks = [{k: v.clone() for k, v in e.items() if torch.is_tensor(v)} for e in P["blocks"]]
    got_top = {k: v.clone() for k, v in P.items() if torch.is_tensor(v)}
    m.unet.invalidate()
    fresh = m.unet._pack(dev)
    assert fresh is not P
    for k, v in fresh.items():
        if torch.is_tensor(v):
            assert torch.equal(got_top[k], v), k
    n_t = 0
    for e_got, e in zip(got_blocks, fresh["blocks"]):
        for k, v in e.items():
            if torch.is_tensor(v):
                assert torch.equal(e_got[k], v), (e["spec"].name, k)
        for k in ("conv0.w", "conv1.w", "skip.w", "proj.w"):
            if k in e:
                assert torch.equal(e_got[k + ".T"], _dgrad_weight(dict(e), k)), (e["spec"].name, k + ".T")
                n_t += 1
    assert n_t >= 2 * len(got_blocks)


def test_data_parallel_and_accumulated_steps_equal_the_big_batch(dev):
    """What the all-reduce computes, done by hand (the collective itself is covered on gloo): two replicas on
    half batches, gradients summed, update scaled by 1/world -- and the same through gradient accumulation --
    both equal one step on the whole batch."""
    from mad_amd.train import UNetTrainer
    cfg, m, sd, x, t, lab, target = _setup("model_tiny.yaml", "fp32", dev, B=4)
    x, t, lab, target = (v.to(dev) for v in (x, t, lab, target))

    def fresh():
        _, mm, *_ = _setup("model_tiny.yaml", "fp32", dev, B=4)
        return UNetTrainer(mm.unet, lr=1e-3, dropout=0.0)

    whole = fresh()
    whole.forward_backward(x, t, lab, target)
    whole.step()
    r0, r1 = fresh(), fresh()
    r0.forward_backward(x[:2], t[:2], lab[:2], target[:2])
    r1.forward_backward(x[2:], t[2:], lab[2:], target[2:])
    total = r0.gflat + r1.gflat                      # the all-reduce (sum)
    r0.gflat.copy_(total)
    r0.step(world_size=2)
    assert rel_err(r0.flat.cpu(), whole.flat.cpu()) < 1e-5
    acc = fresh()
    acc.forward_backward(x[:2], t[:2], lab[:2], target[:2], accumulate=True)
    acc.forward_backward(x[2:], t[2:], lab[2:], target[2:], accumulate=True)
    assert acc.micro_batches == 2
    acc.step()
    assert rel_err(acc.flat.cpu(), whole.flat.cpu()) < 1e-5
    assert rel_err(acc.ema.cpu(), whole.ema.cpu()) < 1e-5


def test_buckets_are_reduced_once_and_only_when_complete(dev):
    """The overlap bookkeeping, with the collective replaced by a recorder: every bucket of the flat gradient is
    handed over exactly once, in roughly reverse parameter order, and nothing writes into it afterwards."""
    from mad_amd.train import UNetTrainer
    cfg, m, sd, x, t, lab, target = _setup("model_tiny.yaml", "bf16", dev, B=2)
    tr = UNetTrainer(m.unet, dropout=0.1, bucket_mb=0.5)
    assert len(tr.buckets) > 4 and tr.buckets[0][0] == 0 and tr.buckets[-1][1] == tr.gflat.numel()
    assert all(a[1] == b[0] for a, b in zip(tr.buckets, tr.buckets[1:]))
    seen = []
    tr._reduce_fn = lambda lo, hi: seen.append((lo, hi, tr.gflat[lo:hi].clone()))
    x, t, lab, target = (v.to(dev) for v in (x, t, lab, target))
    tr.forward_backward(x, t, lab, target, accumulate=True, last=False)        # first micro-batch: nothing handed over
    assert not seen
    tr.forward_backward(x, t, lab, target, accumulate=True, last=True)
    tr.sync_grads()
    assert sorted((lo, hi) for lo, hi, _ in seen) == tr.buckets
    for lo, hi, snap in seen:
        assert torch.equal(snap, tr.gflat[lo:hi])
    order = tr._launched
    assert order[0] > order[-1]                         # the tail of the parameter list completes first
    assert seen[0][2].abs().max() > 0
