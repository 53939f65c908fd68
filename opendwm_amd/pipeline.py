"""Denoise-loop harness: the hot loop of CrossviewTemporalSD.inference_pipeline
(src/dwm/pipelines/ctsd.py:1496-1575) for the full-sequence / classifier-free-guidance /
FlowMatch-Euler case (examples/ctsd_35_6views_video_generation.json:34-35), driving
opendwm_amd.dit.DiTCrossviewTemporalConditionModel.  It exists so the path can be timed
and parity-checked without diffusers / the dataset stack; with diffusers installed the
unchanged ctsd.py drives the same model class.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import ops

bf16 = torch.bfloat16


class FlowMatchEulerSchedule:
    """Sigma table of diffusers FlowMatchEulerDiscreteScheduler (0.31.0) without dynamic
    shifting: set_timesteps(n) -> sigmas[n+1] (trailing 0), timesteps = sigmas[:-1]*1000;
    step: x += (sigma[i+1] - sigma[i]) * v   (fp32)."""

    def __init__(self, num_train_timesteps: int = 1000, shift: float = 3.0):
        self.num_train_timesteps, self.shift = num_train_timesteps, shift
        ts = torch.linspace(1, num_train_timesteps, num_train_timesteps).flip(0) / num_train_timesteps
        ts = shift * ts / (1 + (shift - 1) * ts)
        self._sigma_max, self._sigma_min = ts[0].item(), ts[-1].item()
        self.sigmas = None
        self.timesteps = None

    def set_timesteps(self, num_inference_steps: int):
        n = self.num_train_timesteps
        t = torch.linspace(self._sigma_max * n, self._sigma_min * n, num_inference_steps)
        sig = t / n
        sig = self.shift * sig / (1 + (self.shift - 1) * sig)
        self.sigmas = torch.cat([sig, torch.zeros(1)]).float()
        self.timesteps = self.sigmas[:-1] * n
        return self


class CTSDDenoiser:
    """latents [B,T,V,C,H,W] fp32 on device; conditions = the CFG-doubled model kwargs
    ([2B,...], unconditional half first, as get_conditions builds them, ctsd.py:416-453)."""

    def __init__(self, model, guidance_scale: float = 4.0, inference_steps: int = 40, shift: float = 3.0):
        self.model = model
        self.guidance_scale = guidance_scale
        self.schedule = FlowMatchEulerSchedule(shift=shift).set_timesteps(inference_steps)
        self.inference_steps = inference_steps
        self._ts_dev = None

    def prepare(self, latents: torch.Tensor, conditions: Dict[str, torch.Tensor]):
        dev = latents.device
        self.latents = latents.to(torch.float32).contiguous().clone()
        B = latents.shape[0]
        self.model_in = torch.empty((2 * B, *latents.shape[1:]), dtype=bf16, device=dev)
        lat16 = ops.cast_bf16(self.latents)
        self.model_in[:B].copy_(lat16)
        self.model_in[B:].copy_(lat16)
        self.conditions = {k: (v.to(bf16) if torch.is_tensor(v) and v.is_floating_point() and k != "added_time_ids" else v)
                           for k, v in conditions.items()}
        self._ts_dev = self.schedule.timesteps.to(dev)
        return self

    def step(self, i: int):
        """One denoise step = model forward at the CFG batch + guidance combine + Euler update."""
        B, T, V = self.latents.shape[:3]
        ts = self._ts_dev[i].expand(2 * B, T, V)
        out, _, _ = self.model(self.model_in, ts, **self.conditions)
        pred = out[0]
        dsigma = float(self.schedule.sigmas[i + 1] - self.schedule.sigmas[i])
        ops.cfg_euler_step(pred, self.latents, self.guidance_scale, dsigma, model_in=self.model_in)

    def run(self, latents: torch.Tensor, conditions: Dict[str, torch.Tensor], stop: Optional[int] = None):
        self.prepare(latents, conditions)
        for i in range(self.inference_steps if stop is None else stop):
            self.step(i)
        return self.latents
