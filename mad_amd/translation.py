"""Translation loops: noise encoding, deterministic inversion and guided regeneration.

Mirrors the tensor part of ``TranslationDiffusion`` (``/root/reference/modeling/translation.py``):
``process_label`` :275-280, ``inverse_diffusion`` :282-301, ``domain_translation`` :636-697,
``image_translation`` :510-634, ``sample_xt`` :699-706, ``encode`` :708-754, ``generate`` :756-873,
and the tensor branch of ``process_mask`` :213-273 with ``tensor_dilate`` :54-58.  Method names,
argument names and return conventions follow the reference (CPU float tensor in [0, 1] out of
``generate``); every tensor operation runs in HIP kernels through ``mad_amd.ops``.

Extensions, all optional: images may be batched (B > 1; the reference adds a batch dimension of 1),
parsing masks may be a (B,H,W) stack, the encoded noise is stored step-major so each step reads one
contiguous slice, and ``noise_source`` lets a caller inject the random draws (used by the parity
tests, which replay the reference's draws).

Out of scope here (host-side computer-vision preprocessing on libraries that are not part of the
denoising path): face morphing, histogram matching, landmark detection and copy-paste blur
(translation.py:66-144, 303-508).  ``image_translation`` takes their result as ``inverse_image``.
"""
from __future__ import annotations

import warnings
from collections import namedtuple
from typing import Any, Callable, Dict, List, Optional, Tuple, Union

import numpy as np
import torch

from . import ops
from .scheduler import SCHEDULER_FUNC, CustomDDIMScheduler, CustomDDPMScheduler

Tensor = torch.Tensor

COMP_INFO = namedtuple("COMP_INFO", "scale, step")
DEFAULT_COMP_SETUP = {                      # translation.py:37-47
    1: COMP_INFO(scale=0.8, step=100), 6: COMP_INFO(scale=0.8, step=100),
    9: COMP_INFO(scale=0.8, step=80), 13: COMP_INFO(scale=0.8, step=80),
    10: COMP_INFO(scale=0.35, step=180), 4: COMP_INFO(scale=0.4, step=180),
    8: COMP_INFO(scale=0.4, step=180), 2: COMP_INFO(scale=0.8, step=80), 7: COMP_INFO(scale=0.8, step=80),
}
MAX_STEP = max(ci.step for ci in DEFAULT_COMP_SETUP.values())
COMP_PRIORITY = {4: 1, 8: 1, 10: 1, 9: 2, 13: 2, 2: 3, 7: 3, 1: 4, 6: 4}   # translation.py:51


class TranslationDiffusion:
    BASIC_COMPONENT = [9, 13, 4, 8, 10, 2, 7, 1, 6]    # translation.py:151

    def __init__(self, cfg, device):
        self.device = torch.device(device)
        self.cfg = cfg
        self.num_classes = cfg.MODEL.LABEL_DIM
        self.sample_steps = cfg.EVAL.SAMPLE_STEPS
        self.refine_iterations = cfg.EVAL.REFINE_ITERATIONS
        self.refine_steps = cfg.EVAL.REFINE_STEPS
        self.eta = cfg.EVAL.ETA
        self.image_size = cfg.TRAIN.IMAGE_SIZE
        self.scheduler: Union[CustomDDIMScheduler, CustomDDPMScheduler] = SCHEDULER_FUNC[cfg.EVAL.SCHEDULER](
            num_train_timesteps=cfg.TRAIN.TIME_STEPS,
            prediction_type=cfg.TRAIN.NOISE_SCHEDULER.PRED_TYPE,
            beta_schedule=cfg.TRAIN.NOISE_SCHEDULER.TYPE,
            beta_start=cfg.TRAIN.NOISE_SCHEDULER.BETA_START,
            beta_end=cfg.TRAIN.NOISE_SCHEDULER.BETA_END,
        )
        self.null_text = None          # the in-tree UNet takes no text; kept for call compatibility
        self.generator: Optional[torch.Generator] = None
        self.noise_source: Optional[Callable[[Tuple[int, ...]], Tensor]] = None

    # ------------------------------------------------------------------ inputs
    def randn(self, shape) -> Tensor:
        if self.noise_source is not None:
            z = self.noise_source(tuple(shape))
            return z.to(device=self.device, dtype=torch.float32).contiguous()
        return torch.randn(tuple(shape), generator=self.generator, device=self.device, dtype=torch.float32)

    def process_image(self, image):
        """Path / PIL / ndarray / tensor -> (B,3,R,R) fp32 in [-1,1] on the device (translation.py:201-211)."""
        if isinstance(image, str):
            from PIL import Image
            image = Image.open(image).convert("RGB")
        if isinstance(image, np.ndarray):
            from PIL import Image
            image = Image.fromarray(image)
        if not isinstance(image, torch.Tensor):
            # T.Resize's default interpolation is bilinear (translation.py:193-198); PIL's default is bicubic
            from PIL import Image
            image = image.resize((self.image_size, self.image_size), Image.BILINEAR)
            arr = np.asarray(image, dtype=np.float32) / 255.0
            image = torch.from_numpy(arr).permute(2, 0, 1)
            image = (image - 0.5) / 0.5
        image = image.to(device=self.device, dtype=torch.float32)
        if image.ndim == 3:
            image = image.unsqueeze(0)
        return image.contiguous()

    def _label_map(self, mask, size) -> Tensor:
        """-> (B,H,W) uint8 label maps on the device."""
        if isinstance(mask, str):
            from PIL import Image
            mask = Image.open(mask)
        if isinstance(mask, np.ndarray):
            mask = torch.from_numpy(mask)
        if not isinstance(mask, torch.Tensor):
            from PIL import Image
            mask = torch.from_numpy(np.array(mask.convert("L").resize(size, Image.Resampling.NEAREST)))
        if mask.ndim == 2:
            mask = mask.unsqueeze(0)
        return mask.to(device=self.device, dtype=torch.uint8).contiguous()

    def process_mask(self, mask, size: Tuple[int, int], return_mask_value: bool = False, dilate: bool = True,
                     component_list: List[int] = None, landmarks=None):
        """Face-parsing label map -> keep-mask, 1 = keep the source pixel (translation.py:213-273).
        Returns (1,H,W) for one map, (B,1,H,W) for a stack."""
        if mask is None:
            return (None, None) if return_mask_value else None
        if landmarks is not None:
            raise NotImplementedError("the landmark polygon branch needs mediapipe landmarks (host preprocessing)")
        maps = self._label_map(mask, size)
        comps = list(component_list or self.BASIC_COMPONENT)
        comps.sort(key=lambda c: COMP_PRIORITY[c])
        keep = ops.keep_mask(maps[0], comps, dilate) if maps.shape[0] == 1 else ops.keep_mask(maps, comps, dilate)
        if return_mask_value:
            return (maps[0] if maps.shape[0] == 1 else maps), keep
        return keep

    def process_label(self, label, batch_size: int):
        if label is not None and isinstance(label, int):
            label = torch.nn.functional.one_hot(torch.tensor(label), self.num_classes).float()
            # (B, num_classes) as in the reference, but ONE row expanded over the batch (row stride 0) instead of B
            # copies: the denoiser sees from the stride that every row is the same class and embeds it once
            label = label.unsqueeze(0).to(self.device).expand(batch_size, -1)
        return label

    def sample_xt(self, input_: Tensor, timesteps, mask: Optional[Tensor] = None) -> Tensor:
        noise = self.randn(input_.shape)
        ts = torch.as_tensor(timesteps).reshape(-1).tolist()
        if len(set(ts)) != 1:
            raise NotImplementedError("sample_xt needs one noise level for the whole batch")
        sa, sb = self.scheduler.sqrt_alpha_pair(ts[0])
        m = None if mask is None else mask.to(device=self.device, dtype=torch.float32).contiguous()
        return ops.add_noise(input_.contiguous(), noise, sa, sb, mask=m)

    # ------------------------------------------------------------------ loops
    @torch.no_grad()
    def inverse_diffusion(self, model, input_image: Tensor, class_label: Optional[Tensor] = None,
                          inverse_step: int = 300) -> Tensor:
        noise_output = input_image.clone()
        self.scheduler.set_inverse_timesteps(self.sample_steps, device=self.device)
        for timestep in self.scheduler.timesteps[:inverse_step]:
            model_output = model(noise_output, timestep.reshape(-1), text=self.null_text, class_labels=class_label)
            noise_output = self.scheduler.ddim_inverse_step(model_output, timestep, noise_output).prev_sample
        return noise_output

    @torch.no_grad()
    def encode(self, image: Tensor, class_label: Optional[Tensor], model, start_from_step: int = -1,
               return_noise: bool = False):
        """x_T followed by the noise z_t that reproduces a sampled trajectory of `image` under `model`;
        shape (B, S, 3, H, W), stored step-major."""
        self.scheduler.set_timesteps(self.sample_steps, device=self.device)
        initial_noise_steps = start_from_step if start_from_step != -1 else self.sample_steps
        x_T = self.sample_xt(image, [initial_noise_steps - 1])
        start = -start_from_step if start_from_step != -1 else 0
        steps = self.scheduler.timesteps[start:-1]          # the final step has no noise inserted
        latents = torch.empty((len(steps) + 1,) + tuple(image.shape), dtype=torch.float32, device=self.device)
        latents[0] = x_T
        noises = [] if return_noise else None
        noise_output = x_T
        for i, timestep in enumerate(steps):
            self.scheduler.config.prediction_type = "sample"
            noise_output_next = self.scheduler.step(image, timestep, noise_output, eta=self.eta,
                                                    variance_noise=self.randn(image.shape)).prev_sample
            model_output = model(noise_output, timestep.reshape(-1), text=self.null_text,
                                 class_labels=class_label).float()
            if return_noise:
                noises.append(model_output)
            self.scheduler.config.prediction_type = "epsilon"
            # the kernel writes the recovered noise straight into its slice of the step-major slab
            self.scheduler.compute_eps(model_output, timestep, noise_output, noise_output_next, eta=self.eta,
                                       out=latents[i + 1])
            noise_output = noise_output_next
        latents = latents.permute(1, 0, 2, 3, 4)             # (B, S, 3, H, W) view, step-major storage
        if return_noise:
            return torch.stack(noises, dim=1), latents
        return latents

    @torch.no_grad()
    def generate(self, model, eps_list: Optional[Tensor] = None, class_label: Optional[Tensor] = None,
                 source_image: Optional[Tensor] = None, source_mask: Optional[Tensor] = None,
                 source_mask_param: Optional[Dict[str, Any]] = None, noise_initialize: Optional[Tensor] = None,
                 start_from_step: int = -1, comp_step: Dict[int, int] = None,
                 extra_image: Optional[Tensor] = None, normalize: bool = True):
        if source_mask_param is not None and source_mask is not None:
            warnings.warn("source_mask and source_mask_param are both provided, source_mask will be overwrite")
        if eps_list is not None:
            assert eps_list.shape[1] == (self.sample_steps if start_from_step == -1 else start_from_step)
            if noise_initialize is None:
                noise_output = eps_list[:, 0].contiguous()
            else:
                level = start_from_step if start_from_step != -1 else self.sample_steps
                noise_output = self.scheduler.add_noise(noise_initialize, self.randn(noise_initialize.shape), [level])
            eps_list = eps_list[:, 1:]
        elif noise_initialize is not None:
            noise_output = noise_initialize
        else:
            raise ValueError("generate needs eps_list or noise_initialize")

        added_component = set()
        self.scheduler.config.prediction_type = "epsilon"
        self.scheduler.set_timesteps(self.sample_steps, device=self.device)
        start = -start_from_step if start_from_step != -1 else 0
        steps = self.scheduler.timesteps[start:]
        total_step = len(steps)
        for it, timestep in enumerate(steps):
            if comp_step is not None:
                changed_component = [comp for comp, step in comp_step.items() if step >= (total_step - it)]
                if source_mask_param is not None and len(changed_component) != len(added_component):
                    source_mask = self.process_mask(**{**source_mask_param, "component_list": changed_component})
                    added_component.update(changed_component)
                    # the activated regions of the source are replaced by the guide image (translation.py:820-821)
                    ops.masked_copy_(source_image, extra_image.contiguous(), source_mask.contiguous())
            model_output = model(noise_output, timestep.reshape(-1), text=self.null_text,
                                 class_labels=class_label).float()
            if eps_list is not None and timestep > 0:
                z = eps_list[:, it].contiguous()
            else:
                z = self.randn(noise_output.shape)
            noise_output = self.scheduler.step(model_output, timestep, noise_output, variance_noise=z, eta=self.eta,
                                               original_image=source_image, mask=source_mask).prev_sample

        if self.refine_iterations != 0 and self.refine_steps != 0:
            for _ in range(self.refine_iterations):
                self.scheduler.set_timesteps(1000, device=self.device)
                tail = self.scheduler.timesteps[-self.refine_steps:]
                noise_output = self.sample_xt(noise_output, [int(tail[0])], mask=source_mask)
                for timestep in tail:
                    model_output = model(noise_output, timestep.reshape(-1), text=self.null_text,
                                         class_labels=class_label).float()
                    noise_output = self.scheduler.step(model_output, timestep, noise_output,
                                                       eta=1.0 if timestep > 0 else 0,
                                                       variance_noise=self.randn(noise_output.shape),
                                                       original_image=source_image, mask=source_mask).prev_sample

        out = ops.finalize_image(noise_output.contiguous(), normalize)
        return out.cpu() if normalize else out

    # ------------------------------------------------------------------ entry points
    def domain_translation(self, source_model, target_model, source_image, source_class_label=None,
                           target_class_label=None, parsing_mask=None, return_latent: bool = False,
                           use_eps: bool = True, use_inversion: bool = False, inverse_step: int = 300,
                           dilate: bool = True, start_from_step: int = -1):
        """Makeup removal / beauty filter (translation.py:636-697)."""
        assert use_eps or use_inversion, "At least 'eps' or 'inversion' should be used"
        source_image = self.process_image(source_image)
        parsing_mask = self.process_mask(parsing_mask, tuple(source_image.shape[-2:][::-1]), dilate=dilate)
        source_class_label = self.process_label(source_class_label, source_image.shape[0])
        target_class_label = self.process_label(target_class_label, source_image.shape[0])
        eps_list, inverse_latent = None, None
        if use_eps:
            eps_list = self.encode(source_image, source_class_label, source_model, start_from_step=start_from_step)
        if use_inversion:
            inverse_latent = self.inverse_diffusion(target_model, source_image, target_class_label,
                                                    inverse_step=inverse_step)
        output = self.generate(target_model, eps_list=eps_list, class_label=target_class_label,
                               source_image=source_image, source_mask=parsing_mask,
                               noise_initialize=inverse_latent if use_inversion else None,
                               start_from_step=start_from_step)
        if return_latent:
            return (eps_list.cpu() if use_eps else eps_list), output
        return output

    def image_translation(self, source_model, target_model, source_image, target_image,
                          source_class_label, target_class_label, inverse_step: int = 0,
                          use_encode_eps: bool = True, use_his_matching: bool = False, use_morphing: bool = True,
                          use_copy_paste: bool = False,
                          comp_weight_step: Dict[int, Union[COMP_INFO, Tuple[float, int]]] = DEFAULT_COMP_SETUP,
                          source_parsing_mask=None, target_parsing_mask=None, use_cam: bool = True,
                          inpainting: bool = True, return_res: bool = False, inverse_image=None):
        """Makeup transfer (translation.py:510-634).  ``inverse_image`` is the guide image the
        reference builds on the host by morphing / histogram matching / copy-paste."""
        assert use_morphing or use_his_matching or use_copy_paste, \
            "At least 'his_matching', 'copy_paste' or 'use_morphing' should be used"
        if use_morphing:
            assert (not use_his_matching) and (not use_copy_paste), "Cannot use morphing with others"
        if inverse_image is None:
            raise NotImplementedError(
                "pass inverse_image=: face morphing / histogram matching / copy-paste are host-side "
                "preprocessing outside the denoising path")
        comp_weight_step = {comp: COMP_INFO(*ci) if isinstance(ci, tuple) else ci for comp, ci in comp_weight_step.items()}
        inverse_image = self.process_image(inverse_image)
        source_image = self.process_image(source_image).clone()
        if return_res:
            return ops.finalize_image(inverse_image, True).cpu()
        source_mask_param = {"mask": source_parsing_mask, "size": tuple(source_image.shape[-2:][::-1]),
                             "landmarks": None}
        source_class_label = self.process_label(source_class_label, source_image.shape[0])
        target_class_label = self.process_label(target_class_label, source_image.shape[0])
        source_eps_list = None
        start_from_step = -1
        if self.cfg.EVAL.SCHEDULER != "ddim":
            for ci in comp_weight_step.values():
                start_from_step = max(start_from_step, ci.step)
        if use_encode_eps:
            source_eps_list = self.encode(source_image, source_class_label, source_model,
                                          start_from_step=start_from_step)
        else:
            inverse_image = self.sample_xt(inverse_image, [start_from_step - 1])
        if inverse_step > 0:
            inverse_image = self.inverse_diffusion(target_model, inverse_image, target_class_label,
                                                   inverse_step=inverse_step)
        return self.generate(target_model, eps_list=source_eps_list, class_label=target_class_label,
                             source_image=source_image,
                             source_mask_param=source_mask_param if inpainting else None,
                             noise_initialize=inverse_image, start_from_step=start_from_step,
                             comp_step={comp: (ci.step if use_cam else MAX_STEP) for comp, ci in comp_weight_step.items()},
                             extra_image=inverse_image)
