#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the reference's own Python classes.

Run in the build container only (it reads ``/root/reference``); the GPU box never runs it:
    python tests/golden/make_golden.py

What is imported from the reference, unmodified: ``modeling/model.py`` (``DhariwalUNet`` with
``eff_attn=False``), ``modeling/scheduler.py`` (the overridden step / compute_eps / inversion
methods) and ``modeling/translation.py`` (``encode``, ``generate``, ``sample_xt``,
``process_label``, ``tensor_dilate``, ``process_mask``).  Third-party modules that are absent
here are replaced by empty placeholders before the import; the two diffusers scheduler base
classes get minimal stand-ins restating the members the reference uses from
``diffusers==0.24.0`` (parity at that boundary is unpinned; see ``oracle/scheduler_ref.py``).
Weights come from ``mad_amd.init.seeded_state_dict`` loaded by ``state_dict`` name, so only
inputs and expected outputs are stored.
"""
import os
import sys
import types
import warnings
from dataclasses import dataclass

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)
warnings.filterwarnings("ignore")


# ------------------------------------------------------------------ placeholders
def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _SchedulerConfig(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


@dataclass
class _SchedOut:
    prev_sample: torch.Tensor
    pred_original_sample: torch.Tensor = None


class _Base:
    """Members of the diffusers 0.24.0 scheduler bases that the reference touches."""

    def __init__(self, num_train_timesteps=1000, prediction_type="epsilon", beta_schedule="linear",
                 beta_start=1e-4, beta_end=0.02):
        assert beta_schedule == "linear"
        self.config = _SchedulerConfig(num_train_timesteps=num_train_timesteps,
                                       prediction_type=prediction_type, clip_sample=True,
                                       clip_sample_range=1.0, thresholding=False,
                                       timestep_spacing="leading", steps_offset=0)
        self.betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.one = torch.tensor(1.0)
        self.final_alpha_cumprod = torch.tensor(1.0)
        self.num_inference_steps = None
        self.custom_timesteps = False
        self.variance_type = "fixed_small"
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy())

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = num_inference_steps
        ratio = self.config.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
        self.timesteps = torch.from_numpy(ts + self.config.steps_offset).to(device)

    def add_noise(self, original_samples, noise, timesteps):
        ac = self.alphas_cumprod.to(dtype=original_samples.dtype)
        sa = ac[timesteps] ** 0.5
        sa = sa.flatten()
        while len(sa.shape) < len(original_samples.shape):
            sa = sa.unsqueeze(-1)
        sb = (1 - ac[timesteps]) ** 0.5
        sb = sb.flatten()
        while len(sb.shape) < len(original_samples.shape):
            sb = sb.unsqueeze(-1)
        return sa * original_samples + sb * noise


class _DDPMBase(_Base):
    def previous_timestep(self, timestep):
        n = self.num_inference_steps if self.num_inference_steps else self.config.num_train_timesteps
        return timestep - self.config.num_train_timesteps // n

    def _get_variance(self, t, predicted_variance=None, variance_type=None):
        prev_t = self.previous_timestep(t)
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
        cur_beta = 1 - a_t / a_p
        variance = (1 - a_p) / (1 - a_t) * cur_beta
        return torch.clamp(variance, min=1e-20)


class _DDIMBase(_Base):
    def _get_variance(self, timestep, prev_timestep):
        a_t = self.alphas_cumprod[timestep]
        a_p = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        return ((1 - a_p) / (1 - a_t)) * (1 - a_t / a_p)


def _randn_tensor(shape, generator=None, device=None, dtype=None, layout=None):
    return torch.randn(shape, generator=generator, device=device, dtype=dtype)


def install_placeholders():
    import transformers  # noqa: F401  (real package; must be imported before the placeholders)
    _mod("xformers")
    sys.modules["xformers"].ops = _mod("xformers.ops", MemoryEfficientAttentionFlashAttentionOp=None,
                                       memory_efficient_attention=None)
    d = _mod("diffusers", UNet2DConditionModel=object, DDIMScheduler=_DDIMBase, DDPMScheduler=_DDPMBase)
    _mod("diffusers.schedulers")
    _mod("diffusers.schedulers.scheduling_ddim", DDIMSchedulerOutput=_SchedOut)
    _mod("diffusers.schedulers.scheduling_ddpm", DDPMSchedulerOutput=_SchedOut)
    _mod("diffusers.utils")
    _mod("diffusers.utils.torch_utils", randn_tensor=_randn_tensor)
    _mod("cv2", INTER_LINEAR=1, INTER_NEAREST=0)
    _mod("dlib")
    _mod("mediapipe")
    _mod("skimage")
    _mod("skimage.exposure", match_histograms=None)
    _mod("skimage.draw")
    _mod("torchvision")
    _mod("torchvision.transforms")
    _mod("torchvision.transforms.functional", gaussian_blur=None)
    return d


# ------------------------------------------------------------------ helpers
def build_ref_unet(model_mod, spec, seed):
    from mad_amd.init import seeded_state_dict
    net = model_mod.DhariwalUNet(img_resolution=spec.img_resolution, in_channels=spec.in_channels,
                                 out_channels=spec.out_channels, label_dim=spec.label_dim,
                                 augment_dim=0, model_channels=spec.model_channels,
                                 channel_mult=list(spec.channel_mult), channel_mult_emb=4,
                                 num_blocks=spec.num_blocks, dropout=0.10, label_dropout=0,
                                 eff_attn=False)
    sd = seeded_state_dict(spec, seed)
    ref_keys = list(net.state_dict().keys())
    assert sorted(ref_keys) == sorted(sd.keys()), "state_dict keys differ from the reference"
    for k, v in net.state_dict().items():
        assert tuple(v.shape) == tuple(sd[k].shape), k
    net.load_state_dict(sd, strict=True)
    return net.eval()


def seeded_inputs(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def onehot(idx, n):
    return torch.nn.functional.one_hot(torch.tensor(idx), n).float()


class Adapter:
    """What the loops need around ``DhariwalUNet`` (text kwarg ignored, float timestep, [0])."""

    def __init__(self, net):
        self.net = net

    def __call__(self, x, t, text=None, class_labels=None):
        return self.net(x, t.to(torch.float32), class_labels)[0]


def sampler_goldens(ref_sched):
    """sampler.npz: ``step`` / ``compute_eps`` / ``ddim_inverse_step`` / ``add_noise`` of both reference schedulers
    at t in {0, 1, 179, 699, 999} on the 1000- and the 50-step table, for every prediction type the reference
    accepts (scheduler.py:52-71, 157-176, 251-270, 348-362, 455-469, 547-566), masked and unmasked."""
    samp = {}
    T_LIST = [0, 1, 179, 699, 999]
    shape = (2, 3, 8, 8)
    for kind, cls in (("ddpm", ref_sched.CustomDDPMScheduler), ("ddim", ref_sched.CustomDDIMScheduler)):
        sch = cls(num_train_timesteps=1000, prediction_type="epsilon", beta_schedule="linear",
                  beta_start=1e-4, beta_end=0.02)
        samp[f"{kind}/alphas_cumprod"] = sch.alphas_cumprod.numpy()
        for n_steps in (1000, 50):
            sch.set_timesteps(n_steps)
            samp[f"{kind}/timesteps_{n_steps}"] = sch.timesteps.numpy()
            sch.set_inverse_timesteps(n_steps)
            samp[f"{kind}/inverse_timesteps_{n_steps}"] = sch.timesteps.numpy()
        for n_steps in (1000, 50):
            sch.set_timesteps(n_steps)
            ratio = 1000 // n_steps
            for t in T_LIST:
                t = (t // ratio) * ratio
                key = f"{kind}/n{n_steps}/t{t}"
                sample = seeded_inputs(shape, 7 * t + 1, 1.5)
                eps = seeded_inputs(shape, 7 * t + 2)
                noise = seeded_inputs(shape, 7 * t + 3)
                orig = seeded_inputs(shape, 7 * t + 4, 0.7)
                nxt = seeded_inputs(shape, 7 * t + 5, 1.2)
                mask = (seeded_inputs((1, 8, 8), 7 * t + 6) > 0).float()
                for nm, v in (("sample", sample), ("eps", eps), ("noise", noise), ("orig", orig),
                              ("next", nxt), ("mask", mask)):
                    samp[f"{key}/{nm}"] = v.numpy()
                for pt in ("epsilon", "sample", "v_prediction"):
                    sch.config.prediction_type = pt
                    tt = torch.tensor(t)
                    r = sch.step(eps, tt, sample, eta=0.01, variance_noise=noise)
                    samp[f"{key}/{pt}/step"] = r.prev_sample.numpy()
                    samp[f"{key}/{pt}/x0"] = r.pred_original_sample.numpy()
                    r = sch.step(eps, tt, sample, eta=0.01, variance_noise=noise,
                                 original_image=orig, mask=mask)
                    samp[f"{key}/{pt}/step_masked"] = r.prev_sample.numpy()
                    if kind == "ddim":
                        r = sch.step(eps, tt, sample, eta=1.0, variance_noise=noise,
                                     use_clipped_model_output=True)
                        samp[f"{key}/{pt}/step_eta1_clipped"] = r.prev_sample.numpy()
                    if t > 0:
                        samp[f"{key}/{pt}/compute_eps"] = sch.compute_eps(eps, tt, sample, nxt, eta=0.01).numpy()
                    if t >= ratio or True:
                        r = sch.ddim_inverse_step(eps, tt, sample)
                        samp[f"{key}/{pt}/inverse"] = r.prev_sample.numpy()
                sch.config.prediction_type = "epsilon"
                samp[f"{key}/add_noise"] = sch.add_noise(orig, noise, torch.tensor([t])).numpy()
        if kind == "ddpm":
            sch.set_timesteps(1000)
            samp["ddpm/var_1"] = sch._get_variance(1).numpy()
            samp["ddpm/var_999"] = sch._get_variance(999).numpy()
    np.savez_compressed(os.path.join(HERE, "sampler.npz"), **samp)
    print("sampler: keys", len(samp))



def main():
    install_placeholders()
    sys.path.insert(0, REF)
    import modeling.model as ref_model
    import modeling.scheduler as ref_sched
    import modeling.translation as ref_trans
    from mad_amd.config import load_cfg
    from mad_amd.unet_spec import spec_from_cfg

    torch.set_grad_enabled(False)
    torch.set_num_threads(8)

    # ---------------------------------------------------------------- 1. UNet
    for tag, cfg_file, B, with_taps in (("tiny", "model_tiny.yaml", 2, True),
                                        ("64", "model_64_64.yaml", 1, False),
                                        ("256", "model_256_256.yaml", 1, False)):
        cfg = load_cfg(os.path.join(REPO, "configs", cfg_file))
        spec = spec_from_cfg(cfg)
        net = build_ref_unet(ref_model, spec, seed=1)
        R = spec.img_resolution
        x = seeded_inputs((B, 3, R, R), 100 + B)
        t = torch.tensor([333.0])
        labels = onehot([i % 2 for i in range(B)], 2)
        out = {"x": x.numpy(), "t": t.numpy(), "labels": labels.numpy(), "seed": np.int64(1)}
        if with_taps:
            taps = {}
            hooks = []
            for name, m in list(net.enc.items()) + list(net.dec.items()):
                prefix = "enc." if name in net.enc and m is net.enc[name] else "dec."
                hooks.append(m.register_forward_hook(
                    lambda mod, i, o, key=prefix + name: taps.__setitem__(key, o.detach().clone())))
            y = net(x, t, labels)[0]
            for h in hooks:
                h.remove()
            for k, v in taps.items():
                out["tap:" + k] = v.numpy()
            # per-sample timesteps (B,) as well
            t2 = torch.tensor([10.0, 900.0])
            out["t_per_sample"] = t2.numpy()
            out["y_per_sample"] = net(x, t2, labels)[0].numpy()
        else:
            y = net(x, t, labels)[0]
        out["y"] = y.numpy()
        np.savez_compressed(os.path.join(HERE, f"unet_{tag}.npz"), **out)
        print(f"unet_{tag}: out absmax {y.abs().max():.4f} std {y.std():.4f}")

    sampler_goldens(ref_sched)

    # ---------------------------------------------------------------- 3. masks
    rng = np.random.RandomState(5)
    lab = np.zeros((64, 64), dtype=np.int64)
    yy, xx = np.mgrid[0:64, 0:64]

    def ell(cy, cx, ry, rx):
        return ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1

    lab[ell(34, 32, 26, 20)] = 4
    lab[ell(58, 32, 8, 12)] = 10
    lab[ell(36, 32, 7, 4)] = 8
    lab[ell(26, 22, 2, 5)] = 1
    lab[ell(26, 42, 2, 5)] = 6
    lab[ell(21, 22, 1.2, 6)] = 2
    lab[ell(21, 42, 1.2, 6)] = 7
    lab[ell(47, 32, 1.5, 7)] = 9
    lab[ell(50, 32, 1.5, 6)] = 13
    lab[ell(6, 32, 6, 22)] = 12     # hair: outside BASIC_COMPONENT
    lab[0:3, 0:5] = 1               # eye pixels at the border exercise the reflect padding
    td = ref_trans.TranslationDiffusion.__new__(ref_trans.TranslationDiffusion)
    td.device = "cpu"
    masks = {"label_map": lab}
    masks["keep_all"] = td.process_mask(lab.astype(np.uint8), (64, 64)).numpy()
    masks["keep_nodilate"] = td.process_mask(lab.astype(np.uint8), (64, 64), dilate=False).numpy()
    for comps in ([9, 13, 2, 7], [9, 13, 2, 7, 1, 6], [4, 8, 10], [1, 6]):
        masks["keep_" + "_".join(map(str, comps))] = td.process_mask(
            lab.astype(np.uint8), (64, 64), component_list=list(comps)).numpy()
    b = torch.from_numpy((rng.rand(1, 20, 20) > 0.9).astype(np.float32))
    masks["dilate_in"] = b.numpy()
    masks["dilate_5"] = ref_trans.tensor_dilate(b, 5).numpy()
    masks["dilate_11"] = ref_trans.tensor_dilate(b, 11).numpy()
    np.savez_compressed(os.path.join(HERE, "masks.npz"), **masks)
    print("masks: keep fraction", float(masks["keep_all"].mean()))

    # ---------------------------------------------------------------- 4. trajectories
    cfg = load_cfg(os.path.join(REPO, "configs", "model_tiny.yaml"))
    spec = spec_from_cfg(cfg)
    net = Adapter(build_ref_unet(ref_model, spec, seed=1))
    traj = {}
    for kind, cls, sample_steps, start in (("ddpm", ref_sched.CustomDDPMScheduler, 1000, 24),
                                           ("ddim", ref_sched.CustomDDIMScheduler, 50, -1)):
        td = ref_trans.TranslationDiffusion.__new__(ref_trans.TranslationDiffusion)
        td.device = "cpu"
        td.num_classes = 2
        td.sample_steps = sample_steps
        td.refine_iterations = 0
        td.refine_steps = 0
        td.eta = 0.01
        td.null_text = None
        td.scheduler = cls(num_train_timesteps=1000, prediction_type="epsilon",
                           beta_schedule="linear", beta_start=1e-4, beta_end=0.02)
        image = (seeded_inputs((2, 3, 16, 16), 42) * 0.4).clamp(-0.9, 0.9)
        src, tgt = td.process_label(0, 2), td.process_label(1, 2)
        mask = (seeded_inputs((1, 16, 16), 9) > 0.3).float()
        # record every random draw in order
        draws = []
        real_randn, real_randn_like = torch.randn, torch.randn_like

        def rec_randn(*a, **k):
            z = real_randn(*a, **k)
            draws.append(z.clone())
            return z

        def rec_randn_like(x, **k):
            z = real_randn_like(x, **k)
            draws.append(z.clone())
            return z

        torch.manual_seed(1234)
        torch.randn, torch.randn_like = rec_randn, rec_randn_like
        try:
            eps_list = td.encode(image, src, net, start_from_step=start)
            n_enc = len(draws)
            out_same = td.generate(net, eps_list=eps_list, class_label=src, source_image=image.clone(),
                                   source_mask=None, start_from_step=start)
            out_tgt = td.generate(net, eps_list=eps_list, class_label=tgt, source_image=image.clone(),
                                  source_mask=mask, start_from_step=start)
        finally:
            torch.randn, torch.randn_like = real_randn, real_randn_like
        traj[f"{kind}/image"] = image.numpy()
        traj[f"{kind}/mask"] = mask.numpy()
        traj[f"{kind}/sample_steps"] = np.int64(sample_steps)
        traj[f"{kind}/start"] = np.int64(start)
        traj[f"{kind}/enc_draws"] = torch.stack(draws[:n_enc]).numpy()
        traj[f"{kind}/gen_draws"] = torch.stack(draws[n_enc:]).numpy()
        traj[f"{kind}/eps_list"] = eps_list.numpy()
        traj[f"{kind}/out_same_label"] = out_same.numpy()
        traj[f"{kind}/out_target_masked"] = out_tgt.numpy()
        print(f"traj {kind}: eps_list {tuple(eps_list.shape)} draws enc {n_enc} gen {len(draws) - n_enc}"
              f" |out-img| {(out_same * 2 - 1 - image).abs().max():.4f}")
    np.savez_compressed(os.path.join(HERE, "trajectory.npz"), **traj)


def refine_and_inversion_goldens():
    """trajectory_refine.npz: the refine loop of ``generate`` (translation.py:842-868) and the
    DDIM-inversion start (``inverse_diffusion`` + ``noise_initialize``, translation.py:282-301, 666-672,
    778-789) run by the reference's own loops on a tiny denoiser, every random draw recorded."""
    install_placeholders()
    sys.path.insert(0, REF)
    import modeling.model as ref_model
    import modeling.scheduler as ref_sched
    import modeling.translation as ref_trans
    from mad_amd.config import load_cfg
    from mad_amd.unet_spec import spec_from_cfg

    torch.set_grad_enabled(False)
    cfg = load_cfg(os.path.join(REPO, "configs", "model_tiny.yaml"))
    spec = spec_from_cfg(cfg)
    net = Adapter(build_ref_unet(ref_model, spec, seed=1))
    out = {}

    def make_td(cls, sample_steps, refine_iterations, refine_steps):
        td = ref_trans.TranslationDiffusion.__new__(ref_trans.TranslationDiffusion)
        td.device, td.num_classes, td.sample_steps = "cpu", 2, sample_steps
        td.refine_iterations, td.refine_steps, td.eta, td.null_text = refine_iterations, refine_steps, 0.01, None
        td.scheduler = cls(num_train_timesteps=1000, prediction_type="epsilon", beta_schedule="linear",
                           beta_start=1e-4, beta_end=0.02)
        return td

    def recorded(fn):
        draws = []
        real_randn, real_randn_like = torch.randn, torch.randn_like

        def rec_randn(*a, **k):
            z = real_randn(*a, **k)
            draws.append(z.clone())
            return z

        def rec_randn_like(x, **k):
            z = real_randn_like(x, **k)
            draws.append(z.clone())
            return z

        torch.randn, torch.randn_like = rec_randn, rec_randn_like
        try:
            res = fn()
        finally:
            torch.randn, torch.randn_like = real_randn, real_randn_like
        return res, draws

    image = (seeded_inputs((2, 3, 16, 16), 43) * 0.4).clamp(-0.9, 0.9)
    mask = (seeded_inputs((1, 16, 16), 10) > 0.3).float()
    out["image"], out["mask"] = image.numpy(), mask.numpy()

    # ---- refine loop, DDPM table, with and without a keep-mask
    td = make_td(ref_sched.CustomDDPMScheduler, 1000, refine_iterations=2, refine_steps=3)
    src, tgt = td.process_label(0, 2), td.process_label(1, 2)
    torch.manual_seed(4321)
    start = 10
    eps_list, d_enc = recorded(lambda: td.encode(image, src, net, start_from_step=start))
    out["refine/start"] = np.int64(start)
    out["refine/eps_list"] = eps_list.numpy()
    for tag, m in (("nomask", None), ("mask", mask)):
        res, d = recorded(lambda: td.generate(net, eps_list=eps_list, class_label=tgt, source_image=image.clone(),
                                              source_mask=m, start_from_step=start))
        out[f"refine/{tag}/draws"] = torch.stack(d).numpy()
        out[f"refine/{tag}/out"] = res.numpy()
        print(f"refine {tag}: {len(d)} draws, out range {float(res.min()):.3f}..{float(res.max()):.3f}")

    # ---- DDIM inversion as the starting latent; the reference asks for device="cuda" there
    # (translation.py:290), which this CPU-only container maps to the CPU
    td = make_td(ref_sched.CustomDDIMScheduler, 50, 0, 0)
    orig_set_inv = td.scheduler.set_inverse_timesteps
    td.scheduler.set_inverse_timesteps = lambda n, device=None: orig_set_inv(n, device="cpu")
    inverse_step = 20
    torch.manual_seed(999)
    latent, d0 = recorded(lambda: td.inverse_diffusion(net, image, tgt, inverse_step=inverse_step))
    assert not d0
    out["inversion/inverse_step"] = np.int64(inverse_step)
    out["inversion/latent"] = latent.numpy()
    # (a) latent alone as the start of a full 50-step generation with fresh noise
    res, d = recorded(lambda: td.generate(net, eps_list=None, class_label=tgt, source_image=image.clone(),
                                          source_mask=mask, noise_initialize=latent, start_from_step=-1))
    out["inversion/latent_only/draws"] = torch.stack(d).numpy() if d else np.zeros((0,), np.float32)
    out["inversion/latent_only/out"] = res.numpy()
    # (b) use_eps and use_inversion together (domain_translation's combination), start_from_step = 30
    start = 30
    eps_list, d_enc = recorded(lambda: td.encode(image, src, net, start_from_step=start))
    res, d = recorded(lambda: td.generate(net, eps_list=eps_list, class_label=tgt, source_image=image.clone(),
                                          source_mask=mask, noise_initialize=latent, start_from_step=start))
    out["inversion/both/start"] = np.int64(start)
    out["inversion/both/eps_list"] = eps_list.numpy()
    out["inversion/both/draws"] = torch.stack(d).numpy()
    out["inversion/both/out"] = res.numpy()
    print(f"inversion: latent |x| {float(latent.abs().max()):.3f}, draws latent_only "
          f"{len(out['inversion/latent_only/draws'])}, both {len(d)}")
    np.savez_compressed(os.path.join(HERE, "trajectory_refine.npz"), **out)


def long_chain_goldens():
    """trajectory_s700.npz: the loop at the length the removal workload runs it (generate_translation.py:25,
    ``--num-of-step 700``): ``encode`` with start_from_step = 700 on the 1000-step DDPM table (699 chained
    sampler steps + denoiser evaluations + compute_eps divisions by sigma_t, translation.py:708-754) followed by
    ``generate`` over the 700 steps (translation.py:756-840), run by the reference's own loops on the tiny
    denoiser.  Random draws come from ``numpy.random.RandomState(seed).standard_normal`` (a frozen stream), one
    call per draw in call order, so a test regenerates them instead of storing 8 MB of noise.  Stored: inputs,
    the two end images, the encoded noise at a few steps and its per-step sum / sum of squares."""
    install_placeholders()
    sys.path.insert(0, REF)
    import modeling.model as ref_model
    import modeling.scheduler as ref_sched
    import modeling.translation as ref_trans
    from mad_amd.config import load_cfg
    from mad_amd.unet_spec import spec_from_cfg

    torch.set_grad_enabled(False)
    cfg = load_cfg(os.path.join(REPO, "configs", "model_tiny.yaml"))
    spec = spec_from_cfg(cfg)
    net = Adapter(build_ref_unet(ref_model, spec, seed=1))
    td = ref_trans.TranslationDiffusion.__new__(ref_trans.TranslationDiffusion)
    td.device, td.num_classes, td.sample_steps = "cpu", 2, 1000
    td.refine_iterations, td.refine_steps, td.eta, td.null_text = 0, 0, 0.01, None
    td.scheduler = ref_sched.CustomDDPMScheduler(num_train_timesteps=1000, prediction_type="epsilon",
                                                 beta_schedule="linear", beta_start=1e-4, beta_end=0.02)
    start, seed = 700, 20261112
    image = (seeded_inputs((2, 3, 16, 16), 44) * 0.4).clamp(-0.9, 0.9)
    mask = (seeded_inputs((1, 16, 16), 11) > 0.3).float()
    src, tgt = td.process_label(0, 2), td.process_label(1, 2)
    rs = np.random.RandomState(seed)
    n_draws = [0]
    real_randn, real_randn_like = torch.randn, torch.randn_like

    def draw(shape):
        n_draws[0] += 1
        return torch.from_numpy(rs.standard_normal(size=tuple(shape)).astype(np.float32))

    def fake_randn(*a, **k):
        shape = a[0] if len(a) == 1 and not isinstance(a[0], int) else a
        return draw(shape)

    torch.randn, torch.randn_like = fake_randn, (lambda x, **k: draw(x.shape))
    try:
        eps_list = td.encode(image, src, net, start_from_step=start)
        n_enc = n_draws[0]
        out_same = td.generate(net, eps_list=eps_list, class_label=src, source_image=image.clone(),
                               source_mask=None, start_from_step=start)
        n_same = n_draws[0]
        out_tgt = td.generate(net, eps_list=eps_list, class_label=tgt, source_image=image.clone(),
                              source_mask=mask, start_from_step=start)
    finally:
        torch.randn, torch.randn_like = real_randn, real_randn_like
    keep = [0, 1, 2, 100, 350, 600, 697, 698, 699]
    e64 = eps_list.double()
    out = {"image": image.numpy(), "mask": mask.numpy(), "start": np.int64(start), "seed": np.int64(seed),
           "n_draws": np.array([n_enc, n_same - n_enc, n_draws[0] - n_same], dtype=np.int64),
           "eps_steps": np.array(keep, dtype=np.int64), "eps_at_steps": eps_list[:, keep].numpy(),
           "eps_sum": e64.sum(dim=(0, 2, 3, 4)).numpy(), "eps_sumsq": (e64 * e64).sum(dim=(0, 2, 3, 4)).numpy(),
           "out_same_label": out_same.numpy(), "out_target_masked": out_tgt.numpy()}
    np.savez_compressed(os.path.join(HERE, "trajectory_s700.npz"), **out)
    print(f"s700: eps_list {tuple(eps_list.shape)}, draws {out['n_draws'].tolist()}, "
          f"|out_same - image| {float((out_same * 2 - 1 - image).abs().max()):.4f}, "
          f"noise rms first/last step {float(eps_list[:, 1].std()):.3f} / {float(eps_list[:, -1].std()):.3f}")


def synthetic_label_map():
    """The 64x64 face-parsing map of the mask fixtures (ellipses with the labels of BASIC_COMPONENT, hair as a
    label outside it, eye pixels at the border for the reflect padding)."""
    lab = np.zeros((64, 64), dtype=np.int64)
    yy, xx = np.mgrid[0:64, 0:64]

    def ell(cy, cx, ry, rx):
        return ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1

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
    lab[0:3, 0:5] = 1
    return lab


def transfer_goldens():
    """trajectory_transfer.npz: the makeup-transfer form of ``generate`` -- per-step component activation
    (``comp_step``), the keep-mask rebuilt from ``source_mask_param`` whenever a component switches on, and the
    activated source regions replaced by the guide image (translation.py:799-821) -- run by the reference's own
    ``encode`` and ``generate`` with the arguments ``image_translation`` passes them (translation.py:576-621),
    on a 64x64 label map with a small denoiser, every random draw recorded.  The guide image stands in for the
    host-side morphing result.  Component steps are the defaults (translation.py:37-47) divided by ten."""
    install_placeholders()
    sys.path.insert(0, REF)
    import modeling.model as ref_model
    import modeling.scheduler as ref_sched
    import modeling.translation as ref_trans
    from mad_amd.config import load_cfg
    from mad_amd.unet_spec import spec_from_cfg

    torch.set_grad_enabled(False)
    cfg = load_cfg(os.path.join(REPO, "configs", "model_tiny64.yaml"))
    spec = spec_from_cfg(cfg)
    net = Adapter(build_ref_unet(ref_model, spec, seed=2))
    lab = synthetic_label_map().astype(np.uint8)
    comp_setup = {1: (0.8, 10), 6: (0.8, 10), 9: (0.8, 8), 13: (0.8, 8), 10: (0.35, 18), 4: (0.4, 18),
                  8: (0.4, 18), 2: (0.8, 8), 7: (0.8, 8)}
    start = max(s for _, s in comp_setup.values())

    def smooth(seed):
        z = seeded_inputs((1, 3, 64, 64), seed)
        z = torch.nn.functional.avg_pool2d(z, 9, stride=1, padding=4) * 4.0
        return z.clamp(-0.9, 0.9)

    source, guide = smooth(51), smooth(52)
    out = {"label_map": lab, "source": source.numpy(), "guide": guide.numpy(), "seed": np.int64(2),
           "start": np.int64(start), "comp": np.array([[c, s] for c, (_, s) in comp_setup.items()], dtype=np.int64),
           "comp_scale": np.array([w for (w, _) in comp_setup.values()], dtype=np.float32)}

    td = ref_trans.TranslationDiffusion.__new__(ref_trans.TranslationDiffusion)
    td.device, td.num_classes, td.sample_steps = "cpu", 2, 1000
    td.refine_iterations, td.refine_steps, td.eta, td.null_text = 0, 0, 0.01, None
    td.scheduler = ref_sched.CustomDDPMScheduler(num_train_timesteps=1000, prediction_type="epsilon",
                                                 beta_schedule="linear", beta_start=1e-4, beta_end=0.02)
    src, tgt = td.process_label(1, 1), td.process_label(0, 1)        # transfer: makeup -> non-makeup label order

    def recorded(fn):
        draws = []
        real_randn, real_randn_like = torch.randn, torch.randn_like

        def rec_randn(*a, **k):
            z = real_randn(*a, **k)
            draws.append(z.clone())
            return z

        def rec_randn_like(x, **k):
            z = real_randn_like(x, **k)
            draws.append(z.clone())
            return z

        torch.randn, torch.randn_like = rec_randn, rec_randn_like
        try:
            res = fn()
        finally:
            torch.randn, torch.randn_like = real_randn, real_randn_like
        return res, draws

    torch.manual_seed(2024)
    eps_list, d_enc = recorded(lambda: td.encode(source, src, net, start_from_step=start))
    out["enc_draws"] = torch.stack(d_enc).numpy()
    out["eps_list"] = eps_list.numpy()
    for tag, use_cam, inpainting in (("cam_inpaint", True, True), ("nocam_inpaint", False, True),
                                     ("cam_noinpaint", True, False)):
        param = {"mask": lab.copy(), "size": (64, 64), "landmarks": None}
        comp_step = {c: (s if use_cam else ref_trans.MAX_STEP) for c, (_, s) in comp_setup.items()}
        src_img = source.clone()
        res, d = recorded(lambda: td.generate(net, eps_list=eps_list, class_label=tgt, source_image=src_img,
                                              source_mask_param=param if inpainting else None,
                                              noise_initialize=guide, start_from_step=start, comp_step=comp_step,
                                              extra_image=guide))
        out[f"{tag}/draws"] = torch.stack(d).numpy()
        out[f"{tag}/out"] = res.numpy()
        out[f"{tag}/source_after"] = src_img.numpy()       # the reference edits source_image in place
        print(f"transfer {tag}: {len(d)} draws, |out - source| {float((res * 2 - 1 - source).abs().mean()):.4f}, "
              f"source pixels replaced {float((src_img != source).float().mean()):.3f}")
    np.savez_compressed(os.path.join(HERE, "trajectory_transfer.npz"), **out)


def param_order_golden():
    """param_order.json: ``named_parameters()`` order of the reference module per config.  The
    reference restores EMA weights BY POSITION (``copy_parameters(shadow_params, model.parameters())``,
    app.py:15-32), so a drop-in module must register its parameters in exactly this order."""
    import json
    install_placeholders()
    sys.path.insert(0, REF)
    import modeling.model as ref_model
    from mad_amd.config import load_cfg
    from mad_amd.unet_spec import spec_from_cfg
    out = {}
    for cfg_file in ("model_tiny.yaml", "model_64_64.yaml", "model_256_256.yaml"):
        spec = spec_from_cfg(load_cfg(os.path.join(REPO, "configs", cfg_file)))
        net = build_ref_unet(ref_model, spec, seed=1)
        out[cfg_file] = [[n, list(p.shape)] for n, p in net.named_parameters()]
    with open(os.path.join(HERE, "param_order.json"), "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "refine":
        refine_and_inversion_goldens()
    elif len(sys.argv) > 1 and sys.argv[1] == "params":
        param_order_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "transfer":
        transfer_goldens()
    elif len(sys.argv) > 1 and sys.argv[1] == "s700":
        long_chain_goldens()
    elif len(sys.argv) > 1 and sys.argv[1] == "sampler":
        install_placeholders()
        sys.path.insert(0, REF)
        import modeling.scheduler as ref_sched_only
        torch.set_grad_enabled(False)
        sampler_goldens(ref_sched_only)
    else:
        main()
        refine_and_inversion_goldens()
        transfer_goldens()
        long_chain_goldens()
        param_order_golden()
