"""Tensor-timestep schedulers of the SD 2.1 path: drop-ins for dwm.schedulers.temporal_independent
(src/dwm/schedulers/temporal_independent.py) - the reference's subclasses of the diffusers schedulers whose methods take a
timestep PER (sample, frame, view) instead of one scalar:

  * DDPMScheduler.add_noise / get_velocity                 (:8-45; training pair of the UNet branch, ctsd.py:1240-1253)
  * DDIMScheduler.step / _get_variance                     (:47-170; the default test scheduler of the UNet, ctsd.py:969-974)
  * FlowMatchEulerDiscreteScheduler.step_by_indices        (:173-197; diffusion forcing - served by ops.cfg_euler_step with
    per-frame sigma steps, pipeline.CTSDDenoiser)

Same method names, argument meaning and return forms.  The per-element arithmetic runs in HIP kernels
(dwm_frame_affine, dwm_cfg_ddim_step); the per-timestep coefficient tables (a few hundred scalars) are gathered with torch
indexing on the device, in fp64 like nothing else on this path needs to be.  The noise tables restate diffusers 0.31.0
(`scaled_linear` / `linear` betas, `leading` / `trailing` / `linspace` spacing, steps_offset, set_alpha_to_one)."""
from __future__ import annotations

import types
from typing import Optional

import numpy as np
import torch

from . import ops

PREDICTION_TYPES = {"epsilon": 0, "sample": 1, "v_prediction": 2}


def make_betas(num_train_timesteps: int, beta_start: float, beta_end: float, beta_schedule: str) -> torch.Tensor:
    if beta_schedule == "linear":
        return torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
    if beta_schedule == "scaled_linear":
        return torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
    raise NotImplementedError(f"beta_schedule {beta_schedule}")


class _SchedulerBase:
    """config + alphas_cumprod, the part of diffusers' DDPM / DDIM scheduler constructors the reference methods read.
    Defaults = the scheduler_config.json of stabilityai/stable-diffusion-2-1 (v_prediction, scaled_linear betas)."""

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012,
                 beta_schedule: str = "scaled_linear", prediction_type: str = "v_prediction", clip_sample: bool = False,
                 clip_sample_range: float = 1.0, set_alpha_to_one: bool = False, steps_offset: int = 1,
                 timestep_spacing: str = "leading", thresholding: bool = False, **unused):
        if prediction_type not in PREDICTION_TYPES:
            raise ValueError(f"prediction_type given as {prediction_type} must be one of `epsilon`, `sample`, or `v_prediction`")
        self.config = types.SimpleNamespace(
            num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end, beta_schedule=beta_schedule,
            prediction_type=prediction_type, clip_sample=clip_sample, clip_sample_range=clip_sample_range,
            set_alpha_to_one=set_alpha_to_one, steps_offset=steps_offset, timestep_spacing=timestep_spacing, thresholding=thresholding)
        self.betas = make_betas(num_train_timesteps, beta_start, beta_end, beta_schedule)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    @classmethod
    def from_config(cls, config: dict):
        return cls(**{k: v for k, v in config.items() if not k.startswith("_")})

    def _table(self, device) -> torch.Tensor:
        if self.alphas_cumprod.device != device:
            self.alphas_cumprod = self.alphas_cumprod.to(device)           # kept there, as the reference does (:17-19)
        return self.alphas_cumprod


def _group_elems(t_shape, x_shape) -> int:
    """elements of x that share one entry of a timestep tensor whose shape is a prefix of x's"""
    if tuple(x_shape[:len(t_shape)]) != tuple(t_shape):
        raise ValueError(f"timesteps {tuple(t_shape)} must be a leading-shape of the samples {tuple(x_shape)}")
    n = 1
    for d in x_shape[len(t_shape):]:
        n *= d
    return n


class DDPMScheduler(_SchedulerBase):
    """add_noise / get_velocity with a timestep tensor of any leading shape of the samples
    (the reference unsqueezes it against the sample, temporal_independent.py:12-14, 33-35)"""

    def _pair(self, a: torch.Tensor, b: torch.Tensor, timesteps: torch.Tensor, sign: float) -> torch.Tensor:
        dev = a.device
        acp = self._table(dev)[timesteps.to(dev).long()].double()
        coef = torch.stack([acp.sqrt(), sign * (1.0 - acp).sqrt()], -1).float().contiguous()
        x = a.float().contiguous()
        y = b.to(device=dev, dtype=torch.float32).contiguous()
        out = ops.frame_affine(x, y, coef.view(-1, 2), _group_elems(timesteps.shape, a.shape))
        return out.to(a.dtype)

    def add_noise(self, original_samples: torch.Tensor, noise: torch.Tensor, timesteps: torch.Tensor) -> torch.Tensor:
        """sqrt(acp[t]) * original_samples + sqrt(1 - acp[t]) * noise                                  (:8-27)"""
        return self._pair(original_samples, noise, timesteps, 1.0)

    def get_velocity(self, sample: torch.Tensor, noise: torch.Tensor, timesteps: torch.Tensor) -> torch.Tensor:
        """sqrt(acp[t]) * noise - sqrt(1 - acp[t]) * sample                                            (:29-45)"""
        return self._pair(noise.to(sample.device), sample, timesteps, -1.0).to(sample.dtype)


class DDIMScheduler(_SchedulerBase):
    def set_timesteps(self, num_inference_steps: int, device=None):
        """diffusers DDIMScheduler.set_timesteps (0.31.0)"""
        n_train = self.config.num_train_timesteps
        if num_inference_steps > n_train:
            raise ValueError("num_inference_steps cannot exceed num_train_timesteps")
        self.num_inference_steps = num_inference_steps
        sp = self.config.timestep_spacing
        if sp == "linspace":
            ts = np.linspace(0, n_train - 1, num_inference_steps).round()[::-1].copy().astype(np.int64)
        elif sp == "leading":
            ratio = n_train // num_inference_steps
            ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64) + self.config.steps_offset
        elif sp == "trailing":
            ratio = n_train / num_inference_steps
            ts = np.round(np.arange(n_train, 0, -ratio)).astype(np.int64) - 1
        else:
            raise ValueError(f"{sp} is not supported")
        self.timesteps = torch.from_numpy(ts).to(device)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def _alpha_prev(self, prev_timestep: torch.Tensor, table: torch.Tensor) -> torch.Tensor:
        return torch.where(prev_timestep >= 0, table[prev_timestep.clamp_min(0)], self.final_alpha_cumprod.to(table))

    def _get_variance(self, timestep: torch.Tensor, prev_timestep: torch.Tensor) -> torch.Tensor:
        """(1 - a_prev) / (1 - a_t) * (1 - a_t / a_prev)                                                (:49-65)"""
        table = self._table(timestep.device)
        a_t, a_prev = table[timestep], self._alpha_prev(prev_timestep, table)
        return ((1 - a_prev) / (1 - a_t)) * (1 - a_t / a_prev)

    def coefficients(self, timestep: torch.Tensor, eta: float = 0.0) -> torch.Tensor:
        """[*timestep.shape, 6] fp32 rows of dwm_cfg_ddim_step for integer timesteps (device tensor)"""
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        t = timestep.long()
        prev = t - self.config.num_train_timesteps // self.num_inference_steps
        table = self._table(t.device).double()
        a_t, a_prev = table[t], self._alpha_prev(prev, table)
        var = ((1 - a_prev) / (1 - a_t)) * (1 - a_t / a_prev)
        std = eta * var.sqrt()
        return torch.stack([a_t.sqrt(), (1 - a_t).sqrt(), a_prev.sqrt(), (1 - a_prev - std * std).sqrt(), std, torch.zeros_like(std)],
                           -1).float().contiguous()

    def step(self, model_output: torch.Tensor, timestep: torch.Tensor, sample: torch.Tensor, eta: float = 0.0,
             use_clipped_model_output: bool = False, generator=None, variance_noise=None, return_dict: bool = True,
             guidance_scale: Optional[float] = None, model_in: Optional[torch.Tensor] = None):
        """the reference's step (:67-170) with `timestep` an integer tensor of any leading shape of `sample`.
        Extension used by the fused denoise loop: `guidance_scale` (model_output = [uncond; cond], the CFG combine of
        ctsd.py:1548-1552 happens in the same kernel) and `model_in` (bf16 buffer receiving the next model input)."""
        if self.config.thresholding:
            raise NotImplementedError("DDIMScheduler: thresholding (no shipped CTSD scheduler config enables it)")
        if eta > 0 and variance_noise is not None and generator is not None:
            raise ValueError("Cannot pass both generator and variance_noise. Please make sure that either `generator` or "
                             "`variance_noise` stays `None`.")
        dev = sample.device
        timestep = timestep.to(dev)
        coef = self.coefficients(timestep, eta)
        prev = sample.float().contiguous()
        if prev.data_ptr() == sample.data_ptr():
            prev = prev.clone()                                   # the kernel updates in place; `sample` stays the caller's
        noise = None
        if eta > 0:
            if variance_noise is None:
                variance_noise = torch.randn(sample.shape, generator=generator, device=generator.device if generator is not None else dev,
                                             dtype=torch.float32)
            noise = variance_noise.to(device=dev, dtype=torch.float32).contiguous()
        x0 = torch.empty_like(prev)
        mo = model_output if model_output.dtype in (torch.bfloat16, torch.float32) else model_output.float()
        ops.cfg_ddim_step(mo.contiguous(), prev, coef.view(-1, 6), _group_elems(timestep.shape, sample.shape),
                          PREDICTION_TYPES[self.config.prediction_type], guidance=guidance_scale,
                          clip_range=self.config.clip_sample_range if self.config.clip_sample else 0.0,
                          use_clipped_model_output=use_clipped_model_output, noise=noise, x0_out=x0, model_in=model_in)
        if not return_dict:
            return (prev, x0)
        return types.SimpleNamespace(prev_sample=prev, pred_original_sample=x0)
