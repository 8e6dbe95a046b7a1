"""TEST INFRASTRUCTURE -- CPU oracle for the translation loops.  Never imported by the product path.

Plain-PyTorch fp32 restatement of the tensor parts of ``TranslationDiffusion``
(``/root/reference/modeling/translation.py``): ``tensor_dilate`` :54-58, the tensor branch of
``process_mask`` :231-273, ``process_label`` :275-280, ``inverse_diffusion`` :282-301,
``sample_xt`` :699-706, ``encode`` :708-754 and ``generate`` :756-873.  The denoiser is any
callable ``model(x, t, class_labels) -> Tensor``; random draws go through ``self.randn`` in the
reference's draw order, so a run seeded like a reference run reproduces it.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional

import torch
import torch.nn.functional as F

from .scheduler_ref import RefScheduler

Tensor = torch.Tensor

BASIC_COMPONENT = [9, 13, 4, 8, 10, 2, 7, 1, 6]                       # translation.py:151
COMP_PRIORITY = {4: 1, 8: 1, 10: 1, 9: 2, 13: 2, 2: 3, 7: 3, 1: 4, 6: 4}   # translation.py:51
DEFAULT_COMP_STEP = {1: 100, 6: 100, 9: 80, 13: 80, 10: 180, 4: 180, 8: 180, 2: 80, 7: 80}  # :37-47
MAX_STEP = 180


def tensor_dilate(bin_img: Tensor, ksize: int) -> Tensor:
    pad = (ksize - 1) // 2
    x = F.pad(bin_img, [pad, pad, pad, pad], mode="reflect")
    return F.max_pool2d(x, kernel_size=ksize, stride=1, padding=0)


def keep_mask(label_map: Tensor, component_list: Optional[List[int]] = None,
              dilate: bool = True) -> Tensor:
    """Label map (H,W) integer -> keep-mask (1,H,W) float: 1 keeps the source pixel, 0 regenerates
    it (translation.py:234-273, without the landmark polygon branch)."""
    comps = sorted(list(component_list or BASIC_COMPONENT), key=lambda c: COMP_PRIORITY[c])
    edit = torch.zeros(label_map.shape, dtype=torch.float32)
    for val in comps:
        if dilate and val in (1, 6):
            eye = (label_map == val).float()
            ring = tensor_dilate(eye[None], 21)[0]
            ring[eye > 0] = 0.0
            edit[ring > 0] = 1.0
        elif val in (4, 8):
            edit[label_map == val] = 1.0
            for num in (1, 6):
                eye = tensor_dilate((label_map == num).float()[None], 11)[0]
                edit[eye > 0] = 0.0
        else:
            edit[label_map == val] = 1.0
    always_hide = ~torch.isin(label_map, torch.as_tensor(BASIC_COMPONENT, dtype=label_map.dtype))
    edit[always_hide] = 0.0
    return (1 - edit)[None]


class RefTranslation:
    def __init__(self, scheduler: RefScheduler, num_classes: int, sample_steps: int = 1000,
                 eta: float = 0.01, refine_steps: int = 0, refine_iterations: int = 0,
                 randn: Optional[Callable] = None):
        self.scheduler = scheduler
        self.num_classes = num_classes
        self.sample_steps = sample_steps
        self.eta = eta
        self.refine_steps = refine_steps
        self.refine_iterations = refine_iterations
        self.randn = randn or (lambda shape: torch.randn(shape))

    def process_label(self, label, batch_size: int):
        if label is not None and isinstance(label, int):
            label = F.one_hot(torch.tensor(label), self.num_classes).float()
            label = label[None].repeat(batch_size, 1)
        return label

    def sample_xt(self, x: Tensor, timesteps: Tensor, mask: Optional[Tensor] = None) -> Tensor:
        noise = self.randn(x.shape)
        out = self.scheduler.add_noise(x, noise, timesteps)
        if mask is not None:
            out = mask * out + (1 - mask) * x
        return out

    def inverse_diffusion(self, model, image: Tensor, class_label, inverse_step: int = 300) -> Tensor:
        x = image.clone()
        self.scheduler.set_inverse_timesteps(self.sample_steps)
        for t in self.scheduler.timesteps[:inverse_step]:
            out = model(x, t.reshape(-1).float(), class_label)
            x = self.scheduler.ddim_inverse_step(out, t, x).prev_sample
        return x

    def encode(self, image: Tensor, class_label, model, start_from_step: int = -1,
               trace: Optional[list] = None) -> Tensor:
        self.scheduler.set_timesteps(self.sample_steps)
        first = start_from_step if start_from_step != -1 else self.sample_steps
        x = self.sample_xt(image, torch.tensor([first - 1]))
        latents = [x]
        start = -start_from_step if start_from_step != -1 else 0
        for t in self.scheduler.timesteps[start:-1]:
            self.scheduler.config.prediction_type = "sample"
            x_next = self.scheduler.step(image, t, x, eta=self.eta, variance_noise=self.randn(x.shape)).prev_sample
            out = model(x, t.reshape(-1).float(), class_label).float()
            self.scheduler.config.prediction_type = "epsilon"
            latents.append(self.scheduler.compute_eps(out, t, x, x_next, eta=self.eta))
            if trace is not None:
                trace.append(x_next)
            x = x_next
        return torch.stack(latents, dim=1)

    def generate(self, model, eps_list: Optional[Tensor] = None, class_label=None,
                 source_image: Optional[Tensor] = None, source_mask: Optional[Tensor] = None,
                 label_map: Optional[Tensor] = None, noise_initialize: Optional[Tensor] = None,
                 start_from_step: int = -1, comp_step: Optional[Dict[int, int]] = None,
                 extra_image: Optional[Tensor] = None, normalize: bool = True,
                 trace: Optional[list] = None, source_after: Optional[list] = None) -> Tensor:
        """``label_map`` plays the role of ``source_mask_param`` with a pre-decoded label map.  The reference
        edits ``source_image`` in place when components switch on (translation.py:820-821); here a copy is
        edited and, when ``source_after`` is a list, appended to it at the end."""
        if eps_list is not None:
            assert eps_list.shape[1] == (self.sample_steps if start_from_step == -1 else start_from_step)
            if noise_initialize is None:
                x = eps_list[:, 0]
            else:
                tt = torch.tensor(start_from_step if start_from_step != -1 else self.sample_steps)
                x = self.scheduler.add_noise(noise_initialize, self.randn(noise_initialize.shape), tt)
            eps_list = eps_list[:, 1:]
        elif noise_initialize is not None:
            x = noise_initialize
        added = set()
        self.scheduler.config.prediction_type = "epsilon"
        self.scheduler.set_timesteps(self.sample_steps)
        start = -start_from_step if start_from_step != -1 else 0
        steps = self.scheduler.timesteps[start:]
        total = len(steps)
        if source_image is not None:
            source_image = source_image.clone()
        for it, t in enumerate(steps):
            if comp_step is not None:
                changed = [c for c, s in comp_step.items() if s >= (total - it)]
                if label_map is not None and len(changed) != len(added):
                    source_mask = keep_mask(label_map, changed)
                    added.update(changed)
                    sel = (source_mask > 0)[:, None].expand(-1, 3, -1, -1)
                    sel = sel.expand(source_image.shape[0], -1, -1, -1)
                    source_image[sel] = extra_image[sel]
            out = model(x, t.reshape(-1).float(), class_label).float()
            if eps_list is not None and t > 0:
                z = eps_list[:, it]
            else:
                z = self.randn(x.shape)
            x = self.scheduler.step(out, t, x, variance_noise=z, eta=self.eta,
                                    original_image=source_image, mask=source_mask).prev_sample
            if trace is not None:
                trace.append(x)
        if self.refine_iterations != 0 and self.refine_steps != 0:
            for _ in range(self.refine_iterations):
                self.scheduler.set_timesteps(1000)
                t0 = self.scheduler.timesteps[-self.refine_steps:][0]
                x = self.sample_xt(x, torch.full((x.shape[0],), int(t0), dtype=torch.long), mask=source_mask)
                for t in self.scheduler.timesteps[-self.refine_steps:]:
                    out = model(x, t.reshape(-1).float(), class_label)
                    x = self.scheduler.step(out, t, x, eta=1.0 if t > 0 else 0,
                                            variance_noise=self.randn(x.shape),
                                            original_image=source_image, mask=source_mask).prev_sample
        if source_after is not None:
            source_after.append(source_image)
        x = x.clamp(-1, 1)
        return x / 2 + 0.5 if normalize else x
