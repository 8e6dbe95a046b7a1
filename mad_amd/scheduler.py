"""DDPM / DDIM samplers of the translation path, with the reference's method names.

Mirrors ``CustomDDPMScheduler`` / ``CustomDDIMScheduler``
(``/root/reference/modeling/scheduler.py:292-585`` and ``:11-289``): ``step``, ``compute_eps``,
``ddim_inverse_step``, ``set_timesteps``, ``set_inverse_timesteps``, ``add_noise``, a mutable
``config.prediction_type`` and the same exceptions (``ValueError`` for an unknown prediction type
or a DDIM step before ``set_timesteps``, ``AssertionError`` for DDPM ``compute_eps`` at t = 0).

The per-step scalars are evaluated on the host as 0-dim fp32 torch tensors in the reference's
operation order (torch's CPU ``** 0.5`` is not always the correctly rounded square root, so the
same library evaluates them here); the tensor arithmetic is one pass of the
``mad_sampler_update`` HIP kernel.  The schedule members the
reference inherits from ``diffusers==0.24.0`` (linear betas, ``alphas_cumprod``, "leading"
timestep spacing, ``previous_timestep``, ``_get_variance``, ``clip_sample=True``) are restated
here; that package is not available to check them against.

Differences from the reference, on purpose: ``timesteps`` stays on the host (reading an element
never synchronises the GPU); ``pred_original_sample`` is only materialised when
``return_pred_original=True``; the random draw the reference makes when ``variance_noise`` is
None comes from ``torch.randn`` on the sample's device.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional, Union

import numpy as np
import torch

from . import _lib, ops
from ._lib import SamplerCoef

Tensor = torch.Tensor
_PRED = {"epsilon": _lib.PRED_EPSILON, "sample": _lib.PRED_SAMPLE, "v_prediction": _lib.PRED_V}


class SchedulerOutput(SimpleNamespace):
    """``prev_sample`` and (optionally) ``pred_original_sample``."""


class _SchedulerBase:
    _algo = None

    def __init__(self, num_train_timesteps: int = 1000, prediction_type: str = "epsilon",
                 beta_schedule: str = "linear", beta_start: float = 1e-4, beta_end: float = 0.02):
        if beta_schedule != "linear":
            raise NotImplementedError(f"{beta_schedule} is not implemented for {self.__class__}")
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, prediction_type=prediction_type,
                                      beta_schedule=beta_schedule, beta_start=beta_start, beta_end=beta_end,
                                      clip_sample=True, clip_sample_range=1.0, thresholding=False,
                                      timestep_spacing="leading", steps_offset=0)
        self.betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self._ac = self.alphas_cumprod                  # fp32 host table (0-dim reads stay fp32 tensors)
        self.one = torch.tensor(1.0)
        self.final_alpha_cumprod = torch.tensor(1.0)
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    # -- timestep tables -------------------------------------------------------------------
    def _check_count(self, n: int) -> None:
        if n > self.config.num_train_timesteps:
            raise ValueError(
                f"`num_inference_steps`: {n} cannot be larger than `self.config.train_timesteps`:"
                f" {self.config.num_train_timesteps} as the unet model trained with this scheduler can only handle"
                f" maximal {self.config.num_train_timesteps} timesteps.")

    def set_timesteps(self, num_inference_steps: int, device=None) -> None:
        self._check_count(num_inference_steps)
        self.num_inference_steps = num_inference_steps
        ratio = self.config.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
        self.timesteps = torch.from_numpy(ts + self.config.steps_offset)

    def set_inverse_timesteps(self, num_inference_steps: int, device=None) -> None:
        """Ascending table for deterministic inversion (scheduler.py:199-228, 495-524)."""
        self._check_count(num_inference_steps)
        self.num_inference_steps = num_inference_steps
        if self.config.timestep_spacing != "leading":
            raise ValueError(f"{self.config.timestep_spacing} is not supported. Please make sure to choose one of"
                             " 'leading' or 'trailing'.")
        ratio = self.config.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round().copy().astype(np.int64)
        self.timesteps = torch.from_numpy(ts + self.config.steps_offset)

    # -- schedule scalars ------------------------------------------------------------------
    def _a(self, t: int, below_zero: float = 1.0) -> Tensor:
        return self._ac[t] if t >= 0 else torch.tensor(below_zero, dtype=torch.float32)

    def sqrt_alpha_pair(self, t: int):
        """(alphas_cumprod[t] ** 0.5, (1 - alphas_cumprod[t]) ** 0.5) as Python floats."""
        a = self._ac[int(t)]
        return float(a ** 0.5), float((1 - a) ** 0.5)

    def add_noise(self, original_samples: Tensor, noise: Tensor, timesteps) -> Tensor:
        ts = torch.as_tensor(timesteps).reshape(-1).tolist()
        if len(set(ts)) != 1:
            # per-sample noise levels: one launch per sample
            outs = [self.add_noise(original_samples[i:i + 1].contiguous(), noise[i:i + 1].contiguous(), [ts[i]])
                    for i in range(len(ts))]
            return torch.cat(outs, 0)
        sa, sb = self.sqrt_alpha_pair(ts[0])
        return ops.add_noise(original_samples.contiguous(), noise.contiguous(), sa, sb)

    def _base_coef(self, op: int) -> SamplerCoef:
        pt = self.config.prediction_type
        if pt not in _PRED:
            raise ValueError(f"prediction_type given as {pt} must be one of `epsilon`, `sample`, or `v_prediction`")
        if self.config.thresholding:
            raise NotImplementedError("dynamic thresholding is not on the translation path")
        c = SamplerCoef()
        c.algo, c.pred_type, c.op = self._algo, _PRED[pt], op
        c.clip = int(bool(self.config.clip_sample))
        c.clip_range = float(self.config.clip_sample_range)
        c.eps_div = 1.0
        return c

    @staticmethod
    def _prep_mask(mask: Optional[Tensor], sample: Tensor) -> Optional[Tensor]:
        if mask is None:
            return None
        return mask.to(device=sample.device, dtype=torch.float32).contiguous()

    def _draw(self, like: Tensor, generator=None) -> Tensor:
        return torch.randn(like.shape, generator=generator, device=like.device, dtype=like.dtype)

    # -- deterministic inversion (shared) --------------------------------------------------
    def _inverse(self, model_output: Tensor, timestep, sample: Tensor, below_zero: float,
                 return_pred_original: bool) -> SchedulerOutput:
        hi = int(timestep)
        lo = min(hi - self.config.num_train_timesteps // self.num_inference_steps,
                 self.config.num_train_timesteps - 1)
        a_lo, a_hi = self._a(lo, below_zero), self._ac[hi]
        c = self._base_coef(_lib.OP_INVERSE)
        c.sqrt_a_t, c.sqrt_b_t = float(a_lo ** 0.5), float((1 - a_lo) ** 0.5)
        c.sqrt_a_prev, c.sqrt_b_prev = float(a_hi ** 0.5), float((1 - a_hi) ** 0.5)
        r = ops.sampler_update(c, model_output.contiguous(), sample.contiguous(), want_x0=return_pred_original)
        if return_pred_original:
            return SchedulerOutput(prev_sample=r[0], pred_original_sample=r[1])
        return SchedulerOutput(prev_sample=r, pred_original_sample=None)


class CustomDDPMScheduler(_SchedulerBase):
    _algo = _lib.ALGO_DDPM
    variance_type = "fixed_small"

    def previous_timestep(self, timestep: int) -> int:
        n = self.num_inference_steps if self.num_inference_steps else self.config.num_train_timesteps
        return timestep - self.config.num_train_timesteps // n

    def _get_variance(self, t: int, predicted_variance=None) -> Tensor:
        a_t, a_p = self._ac[t], self._a(self.previous_timestep(t))
        cur_beta = 1 - a_t / a_p
        var = (1 - a_p) / (1 - a_t) * cur_beta
        return torch.clamp(var, min=1e-20)

    def _coef(self, t: int, op: int) -> SamplerCoef:
        a_t, a_p = self._ac[t], self._a(self.previous_timestep(t))
        b_t, b_p = 1 - a_t, 1 - a_p
        cur_a = a_t / a_p
        cur_b = 1 - cur_a
        c = self._base_coef(op)
        c.sqrt_a_t, c.sqrt_b_t = float(a_t ** 0.5), float(b_t ** 0.5)
        c.c0 = float((a_p ** 0.5 * cur_b) / b_t)
        c.c1 = float(cur_a ** 0.5 * b_p / b_t)
        c.std = float(self._get_variance(t) ** 0.5)
        c.sqrt_a_prev, c.sqrt_b_prev = float(a_p ** 0.5), float((1.0 - a_p) ** 0.5)
        return c

    def step(self, model_output: Tensor, timestep, sample: Tensor, generator=None,
             original_image: Optional[Tensor] = None, mask: Optional[Tensor] = None,
             variance_noise: Optional[Tensor] = None, return_dict: bool = True,
             return_pred_original: bool = False, *args, **kwargs):
        t = int(timestep)
        c = self._coef(t, _lib.OP_STEP)
        noise = variance_noise if variance_noise is not None else self._draw(model_output, generator)
        c.add_variance = int(t > 0)
        c.known_noise = int(t > 0)
        use_mask = original_image is not None and mask is not None
        r = ops.sampler_update(c, model_output.contiguous(), sample.contiguous(), noise=noise.contiguous(),
                               original=original_image.contiguous() if use_mask else None,
                               mask=self._prep_mask(mask, sample) if use_mask else None,
                               want_x0=return_pred_original)
        prev, x0 = (r if return_pred_original else (r, None))
        if not return_dict:
            return (prev,)
        return SchedulerOutput(prev_sample=prev, pred_original_sample=x0)

    def compute_eps(self, model_output: Tensor, timestep, sample: Tensor, next_sample: Tensor,
                    *args, out: Optional[Tensor] = None, **kwargs) -> Tensor:
        assert timestep > 0, "computing eps for ddpm is only available for t > 0"
        c = self._coef(int(timestep), _lib.OP_COMPUTE_EPS)
        c.eps_div = c.std if c.std > 0 else 1.0
        return ops.sampler_update(c, model_output.contiguous(), sample.contiguous(),
                                  next_sample=next_sample.contiguous(), out=out)

    def ddim_inverse_step(self, model_output: Tensor, timestep, sample: Tensor, return_dict: bool = True,
                          return_pred_original: bool = False):
        # below index 0 this class uses alphas_cumprod[0] (scheduler.py:542)
        out = self._inverse(model_output, timestep, sample, float(self._ac[0]), return_pred_original)
        return out if return_dict else (out.prev_sample,)


class CustomDDIMScheduler(_SchedulerBase):
    _algo = _lib.ALGO_DDIM

    def _get_variance(self, timestep: int, prev_timestep: int) -> Tensor:
        a_t, a_p = self._ac[timestep], self._a(prev_timestep)
        return ((1 - a_p) / (1 - a_t)) * (1 - a_t / a_p)

    def _ready(self) -> None:
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after"
                             " creating the scheduler")

    def _coef(self, t: int, op: int, eta: float) -> SamplerCoef:
        prev = t - self.config.num_train_timesteps // self.num_inference_steps
        a_t, a_p = self._ac[t], self._a(prev)
        c = self._base_coef(op)
        c.sqrt_a_t, c.sqrt_b_t = float(a_t ** 0.5), float((1 - a_t) ** 0.5)
        std = eta * self._get_variance(t, prev) ** 0.5
        c.std = float(std)
        c.sqrt_a_prev, c.sqrt_b_prev = float(a_p ** 0.5), float((1.0 - a_p) ** 0.5)
        c.dir_coef = float((1 - a_p - std ** 2) ** 0.5)
        return c

    def step(self, model_output: Tensor, timestep, sample: Tensor, eta: float = 0.1,
             use_clipped_model_output: bool = False, generator=None,
             original_image: Optional[Tensor] = None, mask: Optional[Tensor] = None,
             variance_noise: Optional[Tensor] = None, return_dict: bool = True,
             return_pred_original: bool = False):
        self._ready()
        t = int(timestep)
        c = self._coef(t, _lib.OP_STEP, eta)
        c.use_clipped = int(bool(use_clipped_model_output))
        noise = variance_noise if variance_noise is not None else self._draw(model_output, generator)
        c.add_variance = int(t > 0 and eta > 0)
        c.known_noise = int(t > 0)
        use_mask = original_image is not None and mask is not None
        r = ops.sampler_update(c, model_output.contiguous(), sample.contiguous(), noise=noise.contiguous(),
                               original=original_image.contiguous() if use_mask else None,
                               mask=self._prep_mask(mask, sample) if use_mask else None,
                               want_x0=return_pred_original)
        prev, x0 = (r if return_pred_original else (r, None))
        if not return_dict:
            return (prev,)
        return SchedulerOutput(prev_sample=prev, pred_original_sample=x0)

    def compute_eps(self, model_output: Tensor, timestep, sample: Tensor, next_sample: Tensor,
                    eta: float = 0.1, out: Optional[Tensor] = None) -> Tensor:
        self._ready()
        c = self._coef(int(timestep), _lib.OP_COMPUTE_EPS, eta)
        c.eps_div = c.std if eta > 0 else 1.0
        return ops.sampler_update(c, model_output.contiguous(), sample.contiguous(),
                                  next_sample=next_sample.contiguous(), out=out)

    def ddim_inverse_step(self, model_output: Tensor, timestep, sample: Tensor, return_dict: bool = True,
                          return_pred_original: bool = False):
        # below index 0 this class uses final_alpha_cumprod = 1 (scheduler.py:246)
        out = self._inverse(model_output, timestep, sample, 1.0, return_pred_original)
        return out if return_dict else (out.prev_sample,)


SCHEDULER_FUNC = {"ddim": CustomDDIMScheduler, "ddpm": CustomDDPMScheduler}
