"""TEST INFRASTRUCTURE -- CPU oracle for the sampler arithmetic.  Never imported by the product path.

Restates, in plain PyTorch fp32 on 0-dim CPU tensors (as the reference does), the overridden
methods of ``CustomDDPMScheduler`` / ``CustomDDIMScheduler``
(``/root/reference/modeling/scheduler.py``: DDPM ``step`` :293-423, ``compute_eps`` :425-493,
``ddim_inverse_step`` :526-585; DDIM ``step`` :12-131, ``compute_eps`` :133-197,
``set_inverse_timesteps`` :199-228, ``ddim_inverse_step`` :230-289).

The base classes those subclass live in the third-party package ``diffusers==0.24.0``
(``/root/reference/requirements.txt:2``), which is absent from ``/root/reference`` and not
installed.  The members used from it -- linear ``betas``, ``alphas_cumprod``, ``one`` /
``final_alpha_cumprod`` = 1, ``set_timesteps`` with "leading" spacing, ``previous_timestep``,
``_get_variance`` ("fixed_small", clamped at 1e-20, for DDPM; eq. 16 for DDIM), ``add_noise``,
``clip_sample=True`` with range 1 -- are restated here from that release's published algorithm.
PARITY UNPINNED at that boundary: no copy of diffusers exists here to check the restatement
against; regression anchors are ``alphas_cumprod[0,179,699,999] = 0.9999, 0.7122501,
6.96611e-3, 4.03583e-5`` and DDPM ``var(1) = 5.453479e-5``, ``var(999) = 1.999996e-2``.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional

import numpy as np
import torch

Tensor = torch.Tensor


class RefScheduler:
    def __init__(self, kind: str = "ddpm", num_train_timesteps: int = 1000,
                 prediction_type: str = "epsilon", beta_schedule: str = "linear",
                 beta_start: float = 1e-4, beta_end: float = 0.02):
        assert kind in ("ddpm", "ddim")
        if beta_schedule != "linear":
            raise NotImplementedError(beta_schedule)
        self.kind = kind
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps,
                                      prediction_type=prediction_type, clip_sample=True,
                                      clip_sample_range=1.0, thresholding=False,
                                      timestep_spacing="leading", steps_offset=0)
        self.betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        self.alphas_cumprod = torch.cumprod(1.0 - self.betas, dim=0)
        self.one = torch.tensor(1.0)
        self.num_inference_steps = None
        self.timesteps = torch.arange(num_train_timesteps - 1, -1, -1)

    # -- timestep tables ---------------------------------------------------------------
    def set_timesteps(self, n: int, device=None):
        T = self.config.num_train_timesteps
        if n > T:
            raise ValueError("num_inference_steps larger than num_train_timesteps")
        self.num_inference_steps = n
        ts = (np.arange(0, n) * (T // n)).round()[::-1].copy().astype(np.int64)
        self.timesteps = torch.from_numpy(ts + self.config.steps_offset)

    def set_inverse_timesteps(self, n: int, device=None):
        """Ascending table (scheduler.py:199-228 / :495-524), 'leading' spacing."""
        T = self.config.num_train_timesteps
        if n > T:
            raise ValueError("num_inference_steps larger than num_train_timesteps")
        self.num_inference_steps = n
        ts = (np.arange(0, n) * (T // n)).round().copy().astype(np.int64)
        self.timesteps = torch.from_numpy(ts + self.config.steps_offset)

    def _prev_t(self, t: int) -> int:
        n = self.num_inference_steps
        if self.kind == "ddpm" and not n:
            n = self.config.num_train_timesteps
        return t - self.config.num_train_timesteps // n

    def _acp(self, t: int, floor_index_zero: bool = False) -> Tensor:
        if t >= 0:
            return self.alphas_cumprod[t]
        return self.alphas_cumprod[0] if floor_index_zero else self.one

    def variance(self, t: int) -> Tensor:
        a_t, a_p = self._acp(t), self._acp(self._prev_t(t))
        if self.kind == "ddpm":
            v = (1 - a_p) / (1 - a_t) * (1 - a_t / a_p)
            return torch.clamp(v, min=1e-20)
        return ((1 - a_p) / (1 - a_t)) * (1 - a_t / a_p)

    def add_noise(self, x0: Tensor, noise: Tensor, timesteps: Tensor) -> Tensor:
        a = self.alphas_cumprod[timesteps.reshape(-1).long().cpu()]
        sa = (a ** 0.5).reshape(-1, *([1] * (x0.ndim - 1)))
        sb = ((1 - a) ** 0.5).reshape(-1, *([1] * (x0.ndim - 1)))
        return sa * x0 + sb * noise

    # -- shared pieces -----------------------------------------------------------------
    def _x0_eps(self, model_output: Tensor, sample: Tensor, a_t: Tensor, clip: bool = True):
        b_t = 1 - a_t
        pt = self.config.prediction_type
        if pt == "epsilon":
            x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
            eps = model_output
        elif pt == "sample":
            x0 = model_output
            eps = (sample - a_t ** 0.5 * x0) / b_t ** 0.5
        elif pt == "v_prediction":
            x0 = (a_t ** 0.5) * sample - (b_t ** 0.5) * model_output
            eps = (a_t ** 0.5) * model_output + (b_t ** 0.5) * sample
        else:
            raise ValueError(f"prediction_type given as {pt} must be one of `epsilon`, `sample`, or `v_prediction`")
        if clip and self.config.clip_sample:
            r = self.config.clip_sample_range
            x0 = x0.clamp(-r, r)
        return x0, eps

    def _check_ready(self):
        if self.kind == "ddim" and self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")

    # -- step ---------------------------------------------------------------------------
    def step(self, model_output: Tensor, timestep, sample: Tensor, eta: float = 0.1,
             original_image: Optional[Tensor] = None, mask: Optional[Tensor] = None,
             variance_noise: Optional[Tensor] = None, generator=None,
             use_clipped_model_output: bool = False):
        self._check_ready()
        t = int(timestep)
        a_t, a_p = self._acp(t), self._acp(self._prev_t(t))
        x0, eps = self._x0_eps(model_output, sample, a_t)
        noise = variance_noise if variance_noise is not None else torch.randn(
            model_output.shape, generator=generator, dtype=model_output.dtype)
        if self.kind == "ddpm":
            b_t, b_p = 1 - a_t, 1 - a_p
            cur_a = a_t / a_p
            cur_b = 1 - cur_a
            c0 = (a_p ** 0.5 * cur_b) / b_t
            c1 = cur_a ** 0.5 * b_p / b_t
            std = self.variance(t) ** 0.5
            var_term = std * noise if t > 0 else 0
            unknown = c0 * x0 + c1 * sample + var_term
        else:
            if use_clipped_model_output:
                eps = (sample - a_t ** 0.5 * x0) / (1 - a_t) ** 0.5
            std = eta * self.variance(t) ** 0.5
            var_term = std * noise if (t > 0 and eta > 0) else 0
            direction = (1 - a_p - std ** 2) ** 0.5 * eps
            unknown = (a_p ** 0.5) * x0 + direction + var_term
        if original_image is not None and mask is not None:
            known = (a_p ** 0.5) * original_image + ((1.0 - a_p) ** 0.5) * (noise if t > 0 else 0)
            prev = mask * known + (1.0 - mask) * unknown
        else:
            prev = unknown
        return SimpleNamespace(prev_sample=prev, pred_original_sample=x0)

    # -- compute_eps --------------------------------------------------------------------
    def compute_eps(self, model_output: Tensor, timestep, sample: Tensor, next_sample: Tensor,
                    eta: float = 0.1) -> Tensor:
        self._check_ready()
        t = int(timestep)
        if self.kind == "ddpm":
            assert t > 0, "computing eps for ddpm is only available for t > 0"
        a_t, a_p = self._acp(t), self._acp(self._prev_t(t))
        x0, eps = self._x0_eps(model_output, sample, a_t)
        if self.kind == "ddpm":
            b_t, b_p = 1 - a_t, 1 - a_p
            cur_a = a_t / a_p
            cur_b = 1 - cur_a
            c0 = (a_p ** 0.5 * cur_b) / b_t
            c1 = cur_a ** 0.5 * b_p / b_t
            mean = c0 * x0 + c1 * sample
            std = self.variance(t) ** 0.5
            return (next_sample - mean) / (std if std > 0 else 1)
        std = eta * self.variance(t) ** 0.5
        direction = (1 - a_p - std ** 2) ** 0.5 * eps
        mean = a_p ** 0.5 * x0 + direction
        return (next_sample - mean) / (std if eta > 0 else 1)

    # -- deterministic inversion --------------------------------------------------------
    def ddim_inverse_step(self, model_output: Tensor, timestep, sample: Tensor):
        """x at noise level (t - T//n) -> x at noise level t.  The DDPM class falls back to
        ``alphas_cumprod[0]`` below index 0 (scheduler.py:542), the DDIM class to
        ``final_alpha_cumprod`` = 1 (scheduler.py:246)."""
        T = self.config.num_train_timesteps
        hi = int(timestep)
        lo = min(hi - T // self.num_inference_steps, T - 1)
        a_lo = self._acp(lo, floor_index_zero=(self.kind == "ddpm"))
        a_hi = self.alphas_cumprod[hi]
        x0, eps = self._x0_eps(model_output, sample, a_lo)
        prev = a_hi ** 0.5 * x0 + (1 - a_hi) ** 0.5 * eps
        return SimpleNamespace(prev_sample=prev, pred_original_sample=x0)
