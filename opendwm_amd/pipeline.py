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
    ([2B,...], unconditional half first, as get_conditions builds them, ctsd.py:416-453).

    Modes of inference_pipeline (ctsd.py:1439-1575):
      * full-sequence denoising (default);
      * reference frames: `image_latents` + `reference_frame_count` — the first frames are fed as clean
        latents at timestep 0 every step and restored at the end (:1514-1526, :1623-1627);
      * diffusion forcing: per-frame timestep indices min(i - take_time*spi, max(0, i - j*spi)), per-frame
        scheduler step, frames outside the schedule range left untouched (:1498-1507, :1554-1572)."""

    def __init__(self, model, guidance_scale: float = 4.0, inference_steps: int = 40, shift: float = 3.0):
        self.model = model
        self.guidance_scale = guidance_scale
        self.schedule = FlowMatchEulerSchedule(shift=shift).set_timesteps(inference_steps)
        self.inference_steps = inference_steps
        self._ts_dev = None

    def prepare(self, latents: torch.Tensor, conditions: Dict[str, torch.Tensor],
                image_latents: Optional[torch.Tensor] = None, reference_frame_count: int = 0,
                diffusion_forcing: bool = False, take_time: int = 0, clear_reference_frame_count: int = 0):
        dev = latents.device
        self.diffusion_forcing = diffusion_forcing
        self.take_time = take_time
        if diffusion_forcing and image_latents is not None:
            latents = image_latents                                   # ctsd.py:1470-1471
            image_latents = None
        self.latents = latents.to(torch.float32).contiguous().clone()
        B, T = latents.shape[:2]
        self.ref = reference_frame_count if image_latents is not None else 0
        self.image_latents = None if image_latents is None else image_latents.to(torch.float32).to(dev)
        if diffusion_forcing:
            if self.inference_steps % (T - clear_reference_frame_count) != 0:
                raise ValueError("inference_steps must be a multiple of the frame count in diffusion-forcing mode")
            self.spi = self.inference_steps // (T - clear_reference_frame_count)
        self.model_in = torch.empty((2 * B, *latents.shape[1:]), dtype=bf16, device=dev)
        self._refresh_model_in()
        self.conditions = {k: (v.to(bf16) if torch.is_tensor(v) and v.is_floating_point() and k != "added_time_ids" else v)
                           for k, v in conditions.items()}
        self._ts_dev = self.schedule.timesteps.to(dev)
        self._sig_dev = self.schedule.sigmas.to(dev)
        return self

    def _refresh_model_in(self):
        B = self.latents.shape[0]
        lat16 = ops.cast_bf16(self.latents)
        self.model_in[:B].copy_(lat16)
        self.model_in[B:].copy_(lat16)
        self._inject_reference()

    def _inject_reference(self):
        if self.ref > 0:                                              # clean reference frames, every step
            B = self.latents.shape[0]
            r16 = self.image_latents[:, :self.ref].to(bf16)
            self.model_in[:B, :self.ref].copy_(r16)
            self.model_in[B:, :self.ref].copy_(r16)

    def step(self, i: int):
        """One denoise step = model forward at the CFG batch + guidance combine + scheduler update."""
        B, T, V = self.latents.shape[:3]
        if self.diffusion_forcing:
            j = torch.arange(T)
            idx = torch.minimum(torch.full((T,), i - self.take_time * self.spi), torch.clamp(i - j * self.spi, min=0))
            idx = idx.to(self._ts_dev.device)
            ts = self._ts_dev[idx].view(1, T, 1).expand(2 * B, T, V)
            in_range = (i - j * self.spi >= 0).to(self._sig_dev.device)
            dsig = torch.where(in_range, self._sig_dev[idx + 1] - self._sig_dev[idx], torch.zeros((), device=idx.device))
            dsig = dsig.view(1, T, 1).expand(B, T, V).contiguous().float()
        else:
            ts = self._ts_dev[i].expand(2 * B, T, V)
            if self.ref > 0:
                ts = ts.clone()
                ts[:, :self.ref] = 0
            dsig = float(self.schedule.sigmas[i + 1] - self.schedule.sigmas[i])
        out, _, _ = self.model(self.model_in, ts, **self.conditions)
        pred = out[0]
        if torch.is_tensor(dsig):
            ops.cfg_euler_step(pred, self.latents, self.guidance_scale, dsig, model_in=self.model_in,
                               group_elems=self.latents[0, 0, 0].numel())
        else:
            ops.cfg_euler_step(pred, self.latents, self.guidance_scale, dsig, model_in=self.model_in)
        self._inject_reference()

    def result(self) -> torch.Tensor:
        if self.ref > 0:                                              # ctsd.py:1623-1627
            return torch.cat([self.image_latents[:, :self.ref], self.latents[:, self.ref:]], 1)
        return self.latents

    def run(self, latents: torch.Tensor, conditions: Dict[str, torch.Tensor], stop: Optional[int] = None,
            start: int = 0, **prepare_kw):
        self.prepare(latents, conditions, **prepare_kw)
        for i in range(start, self.inference_steps if stop is None else stop):
            self.step(i)
        return self.result()
