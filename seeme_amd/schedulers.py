"""DDIM / DDPM schedulers with the ``diffusers`` surface the reference uses.

The reference instantiates ``diffusers.DDIMScheduler`` / ``diffusers.DDPMScheduler`` from
configs/modules/scheduler.yaml (mld/models/modeltype/mld.py:286-287) and touches only:
``init_noise_sigma``, ``set_timesteps(n)``, ``.timesteps``, ``step(eps, t, x, eta=...).prev_sample``,
``add_noise(x, noise, t)`` and ``.config.num_train_timesteps`` (mld.py:456-464,495-497,598,604-606).
``diffusers`` is not vendored, not pinned and not installed, so the published
DDIM (Song et al. 2020, eq. 12) / DDPM (Ho et al. 2020, eq. 7) updates are restated here with the
conventions of SURVEY.md App. B; parity with a particular diffusers release is UNPINNED.

Point ``target:`` at ``seeme_amd.schedulers.DDIMScheduler`` / ``DDPMScheduler``.

``coef_table()`` exports, per inference step, the 8 scalars consumed by the fused sampling kernel
(seeme_denoiser_sample):  x0 = (x - c1*eps)/c0  (or x0 = out, eps = (x - c0*x0)/c1 when c7 != 0);
x0 clipped to [-1,1] when c6 != 0;  x_prev = c2*x0 + c3*eps + c5*x + c4*noise.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional

import numpy as np
import torch


def _betas(num_train_timesteps, beta_start, beta_end, beta_schedule):
    if beta_schedule == "linear":
        return torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
    if beta_schedule == "scaled_linear":
        return torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
    raise NotImplementedError(f"{beta_schedule} is not implemented")


class _Output(SimpleNamespace):
    pass


class _SchedulerBase:
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 clip_sample=True, prediction_type="epsilon", **kwargs):
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start,
                                      beta_end=beta_end, beta_schedule=beta_schedule, clip_sample=clip_sample,
                                      prediction_type=prediction_type, **kwargs)
        self.betas = _betas(num_train_timesteps, beta_start, beta_end, beta_schedule)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.num_inference_steps = None
        self.timesteps = torch.arange(num_train_timesteps - 1, -1, -1, dtype=torch.int64)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def add_noise(self, original_samples, noise, timesteps):
        """x_t = sqrt(acp[t]) x_0 + sqrt(1-acp[t]) noise, per-sample t (mld.py:604-606)."""
        key = (str(original_samples.device), original_samples.dtype)
        if getattr(self, "_acp_dev_key", None) != key:      # device copy cached: no host-to-device copy per step
            self._acp_dev = self.alphas_cumprod.to(device=original_samples.device, dtype=original_samples.dtype)
            self._acp_dev_key = key
        acp = self._acp_dev
        t = timesteps.to(original_samples.device)
        a = acp[t] ** 0.5
        s = (1 - acp[t]) ** 0.5
        while a.dim() < original_samples.dim():
            a, s = a.unsqueeze(-1), s.unsqueeze(-1)
        return a * original_samples + s * noise

    def __len__(self):
        return self.config.num_train_timesteps

    def _acp_np(self):
        return self.alphas_cumprod.numpy()


class DDIMScheduler(_SchedulerBase):
    """configs/modules/scheduler.yaml:1-14."""

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 clip_sample=True, set_alpha_to_one=True, steps_offset=0, prediction_type="epsilon", **kwargs):
        super().__init__(num_train_timesteps, beta_start, beta_end, beta_schedule, clip_sample, prediction_type,
                         set_alpha_to_one=set_alpha_to_one, steps_offset=steps_offset, **kwargs)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]

    def set_timesteps(self, num_inference_steps: int, device=None):
        self.num_inference_steps = num_inference_steps
        ratio = self.config.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
        ts = ts + self.config.steps_offset
        self.timesteps = torch.from_numpy(ts)
        if device is not None:
            self.timesteps = self.timesteps.to(device)

    def _alphas(self, t: int):
        prev_t = t - self.config.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        return a_t, a_prev

    def step(self, model_output, timestep, sample, eta: float = 0.0, use_clipped_model_output=False,
             generator=None, variance_noise: Optional[torch.Tensor] = None, return_dict=True):
        t = int(timestep)
        a_t, a_prev = self._alphas(t)
        b_t = 1 - a_t
        if self.config.prediction_type == "epsilon":
            x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
            eps = model_output
        elif self.config.prediction_type == "sample":
            x0 = model_output
            eps = (sample - a_t ** 0.5 * x0) / b_t ** 0.5
        else:
            raise NotImplementedError(self.config.prediction_type)
        if self.config.clip_sample:
            x0 = x0.clamp(-1, 1)
        var = (1 - a_prev) / (1 - a_t) * (1 - a_t / a_prev)
        std = eta * var ** 0.5
        prev = a_prev ** 0.5 * x0 + (1 - a_prev - std ** 2) ** 0.5 * eps
        if eta > 0:
            if variance_noise is None:
                variance_noise = torch.randn(model_output.shape, generator=generator, device=model_output.device,
                                             dtype=model_output.dtype)
            prev = prev + std * variance_noise
        return _Output(prev_sample=prev, pred_original_sample=x0)

    def coef_table(self, eta: float = 0.0) -> torch.Tensor:
        rows = []
        for t in self.timesteps.tolist():
            a_t, a_prev = self._alphas(int(t))
            var = (1 - a_prev) / (1 - a_t) * (1 - a_t / a_prev)
            std = eta * var ** 0.5
            rows.append(torch.stack([a_t ** 0.5, (1 - a_t) ** 0.5, a_prev ** 0.5, (1 - a_prev - std ** 2) ** 0.5,
                                     torch.as_tensor(std, dtype=torch.float32), torch.tensor(0.0),
                                     torch.tensor(1.0 if self.config.clip_sample else 0.0),
                                     torch.tensor(0.0 if self.config.prediction_type == "epsilon" else 1.0)]))
        return torch.stack(rows).to(torch.float32).contiguous()

    def needs_noise(self, eta: float = 0.0) -> bool:
        return eta > 0


class DDPMScheduler(_SchedulerBase):
    """configs/modules/scheduler.yaml:32-42 (noise_scheduler) and modules_novae/scheduler.yaml:16-26."""

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 variance_type="fixed_small", clip_sample=True, prediction_type="epsilon", **kwargs):
        super().__init__(num_train_timesteps, beta_start, beta_end, beta_schedule, clip_sample, prediction_type,
                         variance_type=variance_type, **kwargs)
        if variance_type != "fixed_small":
            raise NotImplementedError("only variance_type 'fixed_small' is used by the reference configs")

    def set_timesteps(self, num_inference_steps: int, device=None):
        self.num_inference_steps = num_inference_steps
        ratio = self.config.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
        self.timesteps = torch.from_numpy(ts)
        if device is not None:
            self.timesteps = self.timesteps.to(device)

    def _terms(self, t: int):
        n = self.num_inference_steps or self.config.num_train_timesteps
        prev_t = t - self.config.num_train_timesteps // n
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else torch.tensor(1.0)
        b_t, b_prev = 1 - a_t, 1 - a_prev
        cur_a = a_t / a_prev
        cur_b = 1 - cur_a
        c0 = a_prev ** 0.5 * cur_b / b_t
        c1 = cur_a ** 0.5 * b_prev / b_t
        var = torch.clamp(b_prev / b_t * cur_b, min=1e-20)
        return a_t, b_t, c0, c1, var

    def step(self, model_output, timestep, sample, generator=None, variance_noise=None, return_dict=True):
        t = int(timestep)
        a_t, b_t, c0, c1, var = self._terms(t)
        if self.config.prediction_type == "epsilon":
            x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
        elif self.config.prediction_type == "sample":
            x0 = model_output
        else:
            raise NotImplementedError(self.config.prediction_type)
        if self.config.clip_sample:
            x0 = x0.clamp(-1, 1)
        prev = c0 * x0 + c1 * sample
        if t > 0:
            if variance_noise is None:
                variance_noise = torch.randn(model_output.shape, generator=generator, device=model_output.device,
                                             dtype=model_output.dtype)
            prev = prev + var ** 0.5 * variance_noise
        return _Output(prev_sample=prev, pred_original_sample=x0)

    def coef_table(self, eta: float = 0.0) -> torch.Tensor:
        rows = []
        for t in self.timesteps.tolist():
            a_t, b_t, c0, c1, var = self._terms(int(t))
            std = var ** 0.5 if t > 0 else torch.tensor(0.0)
            rows.append(torch.stack([a_t ** 0.5, b_t ** 0.5, c0, torch.tensor(0.0), std, c1,
                                     torch.tensor(1.0 if self.config.clip_sample else 0.0),
                                     torch.tensor(0.0 if self.config.prediction_type == "epsilon" else 1.0)]))
        return torch.stack(rows).to(torch.float32).contiguous()

    def needs_noise(self, eta: float = 0.0) -> bool:
        return True
