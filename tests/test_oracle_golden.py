"""The CPU oracle against the golden vectors taken from the reference's own classes."""
import os

import numpy as np
import pytest
import torch

from conftest import REPO, load_golden, rel_err
from mad_amd.config import load_cfg
from mad_amd.init import seeded_state_dict
from mad_amd.unet_spec import spec_from_cfg
from oracle.scheduler_ref import RefScheduler
from oracle.translation_ref import RefTranslation, keep_mask, tensor_dilate
from oracle.unet_ref import unet_forward

torch.set_grad_enabled(False)


def _oracle_unet(cfg_file, seed=1):
    cfg = load_cfg(os.path.join(REPO, "configs", cfg_file))
    spec = spec_from_cfg(cfg)
    sd = seeded_state_dict(spec, seed)
    kw = dict(img_resolution=spec.img_resolution, model_channels=spec.model_channels,
              channel_mult=spec.channel_mult, num_blocks=spec.num_blocks, label_dim=spec.label_dim)
    return spec, sd, kw


@pytest.mark.parametrize("tag,cfg_file", [("tiny", "model_tiny.yaml"), ("64", "model_64_64.yaml"),
                                          ("256", "model_256_256.yaml")])
def test_unet_matches_reference(tag, cfg_file):
    g = load_golden(f"unet_{tag}.npz")
    spec, sd, kw = _oracle_unet(cfg_file, int(g["seed"]))
    taps = {}
    y = unet_forward(sd, torch.from_numpy(g["x"]), torch.from_numpy(g["t"]),
                     torch.from_numpy(g["labels"]), taps=taps, **kw)
    assert rel_err(y, g["y"]) < 2e-5
    for k in g.files:
        if k.startswith("tap:"):
            assert rel_err(taps[k[4:]], g[k]) < 2e-5, k
    if "y_per_sample" in g.files:
        y2 = unet_forward(sd, torch.from_numpy(g["x"]), torch.from_numpy(g["t_per_sample"]),
                          torch.from_numpy(g["labels"]), **kw)
        assert rel_err(y2, g["y_per_sample"]) < 2e-5


def test_schedule_anchors():
    s = RefScheduler("ddpm")
    ac = s.alphas_cumprod
    np.testing.assert_allclose(ac[[0, 179, 699, 999]].numpy(),
                               [0.9999, 0.7122501, 6.96611e-3, 4.03583e-5], rtol=2e-6)
    s.set_timesteps(1000)
    np.testing.assert_allclose(float(s.variance(1)), 5.453479e-5, rtol=2e-6)
    np.testing.assert_allclose(float(s.variance(999)), 1.999996e-2, rtol=2e-6)
    d = RefScheduler("ddim")
    d.set_timesteps(50)
    assert d.timesteps.tolist() == list(range(980, -1, -20))
    d.set_inverse_timesteps(50)
    assert d.timesteps.tolist() == list(range(0, 1000, 20))


@pytest.mark.parametrize("kind", ["ddpm", "ddim"])
def test_sampler_matches_reference(kind):
    g = load_golden("sampler.npz")
    s = RefScheduler(kind)
    np.testing.assert_array_equal(s.alphas_cumprod.numpy(), g[f"{kind}/alphas_cumprod"])
    for n in (1000, 50):
        s.set_timesteps(n)
        np.testing.assert_array_equal(s.timesteps.numpy(), g[f"{kind}/timesteps_{n}"])
        s.set_inverse_timesteps(n)
        np.testing.assert_array_equal(s.timesteps.numpy(), g[f"{kind}/inverse_timesteps_{n}"])
    checked = 0
    for n in (1000, 50):
        s.set_timesteps(n)
        ratio = 1000 // n
        for t in sorted({(t // ratio) * ratio for t in (0, 1, 179, 699, 999)}):
            key = f"{kind}/n{n}/t{t}"
            T = lambda nm: torch.from_numpy(g[f"{key}/{nm}"])
            for pt in ("epsilon", "sample", "v_prediction"):
                s.config.prediction_type = pt
                r = s.step(T("eps"), t, T("sample"), eta=0.01, variance_noise=T("noise"))
                np.testing.assert_array_equal(r.prev_sample.numpy(), g[f"{key}/{pt}/step"])
                np.testing.assert_array_equal(r.pred_original_sample.numpy(), g[f"{key}/{pt}/x0"])
                r = s.step(T("eps"), t, T("sample"), eta=0.01, variance_noise=T("noise"),
                           original_image=T("orig"), mask=T("mask"))
                np.testing.assert_array_equal(r.prev_sample.numpy(), g[f"{key}/{pt}/step_masked"])
                if kind == "ddim":
                    r = s.step(T("eps"), t, T("sample"), eta=1.0, variance_noise=T("noise"),
                               use_clipped_model_output=True)
                    np.testing.assert_array_equal(r.prev_sample.numpy(), g[f"{key}/{pt}/step_eta1_clipped"])
                if t > 0:
                    z = s.compute_eps(T("eps"), t, T("sample"), T("next"), eta=0.01)
                    np.testing.assert_array_equal(z.numpy(), g[f"{key}/{pt}/compute_eps"])
                r = s.ddim_inverse_step(T("eps"), t, T("sample"))
                np.testing.assert_array_equal(r.prev_sample.numpy(), g[f"{key}/{pt}/inverse"])
                checked += 1
            s.config.prediction_type = "epsilon"
            np.testing.assert_array_equal(
                s.add_noise(T("orig"), T("noise"), torch.tensor([t])).numpy(), g[f"{key}/add_noise"])
    assert checked >= 24
    # the clamp is exercised: some predicted x0 sit exactly on +-1
    assert (np.abs(g[f"{kind}/n1000/t699/epsilon/x0"]) == 1.0).any()


def test_sampler_errors():
    d = RefScheduler("ddim")
    with pytest.raises(ValueError):
        d.step(torch.zeros(1), 10, torch.zeros(1))
    p = RefScheduler("ddpm")
    p.set_timesteps(1000)
    with pytest.raises(AssertionError):
        p.compute_eps(torch.zeros(1), 0, torch.zeros(1), torch.zeros(1))
    p.config.prediction_type = "bogus"
    with pytest.raises(ValueError):
        p.step(torch.zeros(1), 10, torch.zeros(1))


def test_masks_match_reference():
    g = load_golden("masks.npz")
    lab = torch.from_numpy(g["label_map"])
    np.testing.assert_array_equal(keep_mask(lab).numpy(), g["keep_all"])
    np.testing.assert_array_equal(keep_mask(lab, dilate=False).numpy(), g["keep_nodilate"])
    for k in g.files:
        if k.startswith("keep_") and k[5].isdigit():
            comps = [int(c) for c in k[5:].split("_")]
            np.testing.assert_array_equal(keep_mask(lab, comps).numpy(), g[k], err_msg=k)
    b = torch.from_numpy(g["dilate_in"])
    np.testing.assert_array_equal(tensor_dilate(b, 5).numpy(), g["dilate_5"])
    np.testing.assert_array_equal(tensor_dilate(b, 11).numpy(), g["dilate_11"])


@pytest.mark.parametrize("kind", ["ddpm", "ddim"])
def test_trajectory_matches_reference(kind):
    g = load_golden("trajectory.npz")
    spec, sd, kw = _oracle_unet("model_tiny.yaml", 1)
    model = lambda x, t, labels: unet_forward(sd, x, t, labels, **kw)
    steps, start = int(g[f"{kind}/sample_steps"]), int(g[f"{kind}/start"])
    image = torch.from_numpy(g[f"{kind}/image"])
    draws = iter(torch.from_numpy(np.concatenate([g[f"{kind}/enc_draws"], g[f"{kind}/gen_draws"]])))
    tr = RefTranslation(RefScheduler(kind), 2, sample_steps=steps, eta=0.01,
                        randn=lambda shape: next(draws))
    src, tgt = tr.process_label(0, 2), tr.process_label(1, 2)
    eps_list = tr.encode(image, src, model, start_from_step=start)
    ref_eps = torch.from_numpy(g[f"{kind}/eps_list"])
    assert eps_list.shape == ref_eps.shape
    # compute_eps divides by sigma_t (down to 1e-4 for DDIM eta=0.01): compare per step, relative
    for i in range(eps_list.shape[1]):
        assert rel_err(eps_list[:, i], ref_eps[:, i]) < 5e-3, i
    # 24 (DDPM) / 50 (DDIM) chained denoiser calls: fp32 summation-order noise of the conv
    # backend compounds along the chain, so the end images agree to ~1e-4, not to the last bit
    tol = 2e-4
    out = tr.generate(model, eps_list=ref_eps, class_label=src, source_image=image, start_from_step=start)
    assert float((out - torch.from_numpy(g[f"{kind}/out_same_label"])).abs().max()) < tol
    out = tr.generate(model, eps_list=ref_eps, class_label=tgt, source_image=image,
                      source_mask=torch.from_numpy(g[f"{kind}/mask"]), start_from_step=start)
    assert float((out - torch.from_numpy(g[f"{kind}/out_target_masked"])).abs().max()) < tol


def rs_draws(seed):
    """The frozen numpy stream the S = 700 golden was recorded with: one standard_normal call per draw."""
    rs = np.random.RandomState(seed)
    return lambda shape: torch.from_numpy(rs.standard_normal(size=tuple(shape)).astype(np.float32))


def test_long_chain_matches_reference():
    """The loop at the length the removal workload runs it (start_from_step = 700 on the 1000-step DDPM table:
    699 + 700 denoiser evaluations, /root/reference/generate_translation.py:25, modeling/translation.py:708-840)
    against the reference's own run of it: encoded noise at the stored steps, its per-step sum of squares over all
    700 steps, and both end images."""
    g = load_golden("trajectory_s700.npz")
    spec, sd, kw = _oracle_unet("model_tiny.yaml", 1)
    model = lambda x, t, labels: unet_forward(sd, x, t, labels, **kw)
    start = int(g["start"])
    image, mask = torch.from_numpy(g["image"]), torch.from_numpy(g["mask"])
    tr = RefTranslation(RefScheduler("ddpm"), 2, sample_steps=1000, eta=0.01, randn=rs_draws(int(g["seed"])))
    src, tgt = tr.process_label(0, 2), tr.process_label(1, 2)
    eps_list = tr.encode(image, src, model, start_from_step=start)
    assert tuple(eps_list.shape) == (2, start, 3, 16, 16)
    ref_at = torch.from_numpy(g["eps_at_steps"])
    worst = max(rel_err(eps_list[:, int(i)], ref_at[:, k]) for k, i in enumerate(g["eps_steps"]))
    sq = (eps_list.double() ** 2).sum(dim=(0, 2, 3, 4)).numpy()
    worst_sq = float(np.abs(sq / g["eps_sumsq"] - 1).max())
    out_same = tr.generate(model, eps_list=eps_list, class_label=src, source_image=image, start_from_step=start)
    out_tgt = tr.generate(model, eps_list=eps_list, class_label=tgt, source_image=image, source_mask=mask,
                          start_from_step=start)
    e_same = float((out_same - torch.from_numpy(g["out_same_label"])).abs().max())
    e_tgt = float((out_tgt - torch.from_numpy(g["out_target_masked"])).abs().max())
    print(f"S=700 oracle vs reference: noise rel err at stored steps {worst:.2e}, per-step sum of squares "
          f"{worst_sq:.2e}, end images {e_same:.2e} / {e_tgt:.2e}")
    # measured 1.6e-6 / 4e-8 / 9e-7 in the build container; the margins cover another host's fp32 summation order
    assert worst < 2e-3 and worst_sq < 1e-4
    assert e_same < 2e-4 and e_tgt < 2e-4


def test_refine_loop_matches_reference():
    """generate() with REFINE_ITERATIONS=2, REFINE_STEPS=3 (translation.py:842-868), draws replayed."""
    g = load_golden("trajectory_refine.npz")
    spec, sd, kw = _oracle_unet("model_tiny.yaml", 1)
    model = lambda x, t, labels: unet_forward(sd, x, t, labels, **kw)
    image, start = torch.from_numpy(g["image"]), int(g["refine/start"])
    eps = torch.from_numpy(g["refine/eps_list"])
    for tag, mask in (("nomask", None), ("mask", torch.from_numpy(g["mask"]))):
        draws = iter(torch.from_numpy(g[f"refine/{tag}/draws"]))
        tr = RefTranslation(RefScheduler("ddpm"), 2, sample_steps=1000, eta=0.01, refine_steps=3,
                            refine_iterations=2, randn=lambda shape: next(draws))
        out = tr.generate(model, eps_list=eps, class_label=tr.process_label(1, 2), source_image=image,
                          source_mask=mask, start_from_step=start)
        assert next(draws, None) is None                       # every recorded draw was consumed
        assert float((out - torch.from_numpy(g[f"refine/{tag}/out"])).abs().max()) < 2e-4, tag


def test_inversion_start_matches_reference():
    """inverse_diffusion (translation.py:282-301) and generate(noise_initialize=...) alone and together
    with eps_list (the use_eps + use_inversion combination of domain_translation, :666-672, 778-789)."""
    g = load_golden("trajectory_refine.npz")
    spec, sd, kw = _oracle_unet("model_tiny.yaml", 1)
    model = lambda x, t, labels: unet_forward(sd, x, t, labels, **kw)
    image, mask = torch.from_numpy(g["image"]), torch.from_numpy(g["mask"])
    none = iter(())
    tr = RefTranslation(RefScheduler("ddim"), 2, sample_steps=50, eta=0.01, randn=lambda shape: next(none))
    tgt = tr.process_label(1, 2)
    latent = tr.inverse_diffusion(model, image, tgt, inverse_step=int(g["inversion/inverse_step"]))
    ref_latent = torch.from_numpy(g["inversion/latent"])
    assert rel_err(latent, ref_latent) < 1e-4
    draws = iter(torch.from_numpy(g["inversion/latent_only/draws"]))
    tr = RefTranslation(RefScheduler("ddim"), 2, sample_steps=50, eta=0.01, randn=lambda shape: next(draws))
    out = tr.generate(model, eps_list=None, class_label=tgt, source_image=image, source_mask=mask,
                      noise_initialize=ref_latent, start_from_step=-1)
    assert float((out - torch.from_numpy(g["inversion/latent_only/out"])).abs().max()) < 2e-4
    draws = iter(torch.from_numpy(g["inversion/both/draws"]))
    tr = RefTranslation(RefScheduler("ddim"), 2, sample_steps=50, eta=0.01, randn=lambda shape: next(draws))
    out = tr.generate(model, eps_list=torch.from_numpy(g["inversion/both/eps_list"]), class_label=tgt,
                      source_image=image, source_mask=mask, noise_initialize=ref_latent,
                      start_from_step=int(g["inversion/both/start"]))
    assert float((out - torch.from_numpy(g["inversion/both/out"])).abs().max()) < 2e-4


TRANSFER_CASES = [("cam_inpaint", True, True), ("nocam_inpaint", False, True), ("cam_noinpaint", True, False)]


@pytest.mark.parametrize("tag,use_cam,inpainting", TRANSFER_CASES)
def test_transfer_loop_matches_reference(tag, use_cam, inpainting):
    """The makeup-transfer form of generate() (translation.py:799-821 with the arguments of :608-621): per-step
    component activation, keep-mask rebuilt from the label map, activated source regions replaced by the guide."""
    from oracle.translation_ref import MAX_STEP
    g = load_golden("trajectory_transfer.npz")
    spec, sd, kw = _oracle_unet("model_tiny64.yaml", int(g["seed"]))
    model = lambda x, t, labels: unet_forward(sd, x, t, labels, **kw)
    source, guide = torch.from_numpy(g["source"]), torch.from_numpy(g["guide"])
    lab, start = torch.from_numpy(g["label_map"]).long(), int(g["start"])
    comp_step = {int(c): (int(s) if use_cam else MAX_STEP) for c, s in g["comp"]}
    draws = iter(torch.from_numpy(g[f"{tag}/draws"]))
    tr = RefTranslation(RefScheduler("ddpm"), 2, sample_steps=1000, eta=0.01, randn=lambda shape: next(draws))
    after = []
    out = tr.generate(model, eps_list=torch.from_numpy(g["eps_list"]), class_label=tr.process_label(0, 1),
                      source_image=source, label_map=lab if inpainting else None, noise_initialize=guide,
                      start_from_step=start, comp_step=comp_step, extra_image=guide, source_after=after)
    assert next(draws, None) is None
    np.testing.assert_array_equal(after[0].numpy(), g[f"{tag}/source_after"])
    assert float((out - torch.from_numpy(g[f"{tag}/out"])).abs().max()) < 2e-4


def test_transfer_encode_matches_reference():
    g = load_golden("trajectory_transfer.npz")
    spec, sd, kw = _oracle_unet("model_tiny64.yaml", int(g["seed"]))
    model = lambda x, t, labels: unet_forward(sd, x, t, labels, **kw)
    draws = iter(torch.from_numpy(g["enc_draws"]))
    tr = RefTranslation(RefScheduler("ddpm"), 2, sample_steps=1000, eta=0.01, randn=lambda shape: next(draws))
    eps = tr.encode(torch.from_numpy(g["source"]), tr.process_label(1, 1), model, start_from_step=int(g["start"]))
    ref = torch.from_numpy(g["eps_list"])
    assert eps.shape == ref.shape
    for i in range(eps.shape[1]):
        assert rel_err(eps[:, i], ref[:, i]) < 5e-3, i
