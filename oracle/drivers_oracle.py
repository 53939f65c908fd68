"""TEST INFRASTRUCTURE (oracle) - CPU restatement of the window drivers that sit around the denoise loop:

  * CrossviewTemporalSD.autoregressive_inference_pipeline      src/dwm/pipelines/ctsd.py:1656-1833
  * StreamingCrossviewTemporalSD (reset_streaming / send_frame_condition / receive_frame /
    fifo_inference_pipeline)                                   src/dwm/pipelines/ctsd.py:2009-2248

Only tests/ may import this file.  **Pinned** against vectors produced by executing the reference methods
themselves (tests/golden/make_reference_driver_fixtures.py -> reference_drivers.pt; tests/test_reference_fixtures_cpu.py):
emitted frames, final queue latents and the per-window (start, stop, take_time) agree.  The reference ships no tests
of its own; the restatement follows the reference control flow statement by statement, with the model +
scheduler loop (`inference_pipeline`, :1439-1654 / :2032-2103) abstracted into a `window` callable so the
same driver runs over the fp32 oracle model (ctsd_oracle.denoise) or over a recording stub.

window(latent_shape, conditions, image_latents, reference_frame_count, start, stop, take_time, noise)
    -> {"latents": [B,T,V,C,H,W], "images": tensor whose dim 0 is (b t v)}
`noise` is the fresh full-window draw the reference makes at :1473-1475 (None when it does not draw).
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional

import torch

Tensor = torch.Tensor


def take_sequence_clip(item, start: int, stop: int):
    """src/dwm/functional.py:172-181"""
    if isinstance(item, (int, float, bool, str)):
        return item
    if isinstance(item, torch.Tensor):
        return item if item.dim() <= 1 else item[:, start:stop]
    if isinstance(item, list):
        assert len(item) > 0 and all(isinstance(i, list) for i in item)
        return [i[start:stop] for i in item]
    raise Exception("Unsupported type to take sequence clip.")


def latent_sequence_length(n: int, vae_pre: int = 0, vae_stride: int = 1) -> int:
    """get_latent_sequence_length, ctsd.py:1113-1118"""
    assert n % vae_stride == vae_pre or n == 0
    return (n - vae_pre) // vae_stride + (1 if vae_pre > 0 else 0)


def autoregressive(window: Callable, latent_shape, batch: Dict, total_frame_count: int, inference_config: dict,
                   diffusion_forcing: bool, generator: torch.Generator, image_latents: Optional[Tensor] = None,
                   init_noise_sigma: float = 1.0, is_temporal_vae: bool = False) -> Dict:
    """ctsd.py:1656-1833.  `batch` holds per-frame tensors [B, total_frames, ...] (conditions); `image_latents`
    = encoded reference frames when generate_frames_for_reference is False (:1677-1703), else None."""
    cfg = inference_config
    steps = cfg["inference_steps"]
    T = latent_shape[1]
    if diffusion_forcing:
        assert total_frame_count > cfg["sequence_length_per_iteration"]
        clear_ref = cfg.get("clear_reference_frame_count", 0)
        spi = steps // (T - clear_ref)
        queue_head = -1
    reference_frame_count = cfg.get("reference_frame_count", 1)
    seq = cfg["sequence_length_per_iteration"]
    exceptions = cfg.get("autoregression_data_exception_for_take_sequence", [])
    images: List[Tensor] = []

    def clip(a, b):
        return {k: (v if k in exceptions else take_sequence_clip(v, a, b)) for k, v in batch.items()}

    def draw():
        return torch.randn(tuple(latent_shape), generator=generator) * init_noise_sigma

    def call(cond, il, ref, start, stop, take_time):
        # inference_pipeline draws the window noise unless (DF and image_latents given), :1470-1475
        noise = None if (diffusion_forcing and il is not None) else draw()
        return window(latent_shape, cond, il, ref, start, stop, take_time, noise)

    if diffusion_forcing:                                                    # :1710-1727  warm-up of the queue
        out = call(clip(0, seq), image_latents, 0, 0, steps - spi, 0)
        image_latents = out["latents"]

    iteration_batch = None
    for i in range(0, total_frame_count - seq + 1, seq - reference_frame_count):
        iteration_batch = clip(i, i + seq)
        this_ref = 0 if image_latents is None else reference_frame_count
        if diffusion_forcing:
            if queue_head < clear_ref:
                this_ref = T
                queue_head += 1
            out = call(iteration_batch, image_latents, this_ref, steps + (queue_head - 1) * spi,
                       steps + queue_head * spi, queue_head)
            if is_temporal_vae and i == 0:
                images.append(out["images"].chunk(4)[-1])
            else:
                images.append(out["images"])
            fin = torch.tensor([j <= queue_head for j in range(T)]).view(1, -1, 1, 1, 1, 1)
            image_latents = torch.where(fin, image_latents, out["latents"])
            if queue_head == clear_ref and i + seq - reference_frame_count < total_frame_count - seq + 1:
                image_latents = torch.cat([
                    image_latents[:, 1:],
                    torch.randn((latent_shape[0], 1) + tuple(latent_shape[2:]), generator=generator) * init_noise_sigma], 1)
        else:
            ref_lat = latent_sequence_length(this_ref, cfg.get("vae_pre", 0), cfg.get("vae_stride", 1))
            out = call(iteration_batch, image_latents, ref_lat, 0, None, 0)
            images.append(out["images"][latent_shape[0] * this_ref * latent_shape[2]:])
            if i + seq - reference_frame_count < total_frame_count - seq + 1:
                n = latent_sequence_length(reference_frame_count, cfg.get("vae_pre", 0), cfg.get("vae_stride", 1))
                image_latents = out["latents"][:, -n:]

    if diffusion_forcing:                                                    # :1802-1827  flush
        for i in range(queue_head + 1, T):
            out = call(iteration_batch, image_latents, T, steps + (i - 1) * spi, steps + i * spi, i)
            images.append(out["images"])
            fin = torch.tensor([j <= i for j in range(T)]).view(1, -1, 1, 1, 1, 1)
            image_latents = torch.where(fin, image_latents, out["latents"])
    return {"images": torch.cat(images), "latents": image_latents}


class Streaming:
    """StreamingCrossviewTemporalSD, ctsd.py:2009-2248.  `window(latent_shape, conditions, latents, start, stop,
    take_time)` -> (latents, frame or None): the per-call loop :2032-2103 (a frame is produced when
    stop >= inference_steps, from latents[:, take_time])."""

    def __init__(self, window: Callable, inference_config: dict, generator: torch.Generator, init_noise_sigma: float = 1.0):
        self.window, self.cfg, self.generator, self.init_noise_sigma = window, inference_config, generator, init_noise_sigma

    def reset_streaming(self, latent_shape):
        self.conditions, self.condition_count, self.latents = {}, 0, None
        self.frames, self.latent_shape = [], tuple(latent_shape)

    def _run(self, start, stop, take_time=0):
        lat, frame = self.window(self.latent_shape, self.conditions, self.latents, start, stop, take_time)
        if frame is not None:
            self.frames.append(frame)
        return lat

    def send_frame_condition(self, frame_conditions: Optional[Dict]):
        cfg, T = self.cfg, self.latent_shape[1]
        steps = cfg["inference_steps"]
        spi = steps // T
        exc = cfg.get("autoregression_condition_exception_for_take_sequence", [])
        if frame_conditions is None:                                         # flushing, :2112-2133
            assert self.condition_count == cfg["sequence_length_per_iteration"]
            for i in range(1, T):
                lat = self._run(steps + (i - 1) * spi, steps + i * spi, i)
                fin = torch.tensor([j <= i for j in range(T)]).view(1, -1, 1, 1, 1, 1)
                self.latents = torch.where(fin, self.latents, lat)
            return
        if self.condition_count < cfg["sequence_length_per_iteration"]:      # gathering, :2163-2188
            for k, v in frame_conditions.items():
                if k not in self.conditions or k in exc:
                    self.conditions[k] = v
                else:
                    self.conditions[k] = torch.cat([self.conditions[k], v], 1)
            self.condition_count += 1
            if self.condition_count == cfg["sequence_length_per_iteration"]:
                self.latents = torch.randn(self.latent_shape, generator=self.generator) * self.init_noise_sigma
                self.latents = self._run(0, steps)
        else:                                                                # streaming, :2190-2215
            for k, v in frame_conditions.items():
                if k not in self.conditions or k in exc:
                    self.conditions[k] = v
                else:
                    self.conditions[k] = torch.cat([self.conditions[k][:, 1:], v], 1)
            self.latents = torch.cat([
                self.latents[:, 1:],
                torch.randn((self.latent_shape[0], 1) + self.latent_shape[2:], generator=self.generator) * self.init_noise_sigma], 1)
            self.latents = self._run(steps - spi, steps)

    def receive_frame(self):
        return self.frames.pop(0) if self.frames else None

    def fifo(self, latent_shape, batch: Dict, total_frame_count: int) -> Tensor:
        """fifo_inference_pipeline, :2230-2275"""
        assert total_frame_count > self.cfg["sequence_length_per_iteration"]
        exc = self.cfg.get("autoregression_data_exception_for_take_sequence", [])
        out = []
        self.reset_streaming(latent_shape)
        for i in range(total_frame_count):
            self.send_frame_condition({k: (v if k in exc else take_sequence_clip(v, i, i + 1)) for k, v in batch.items()})
            f = self.receive_frame()
            if f is not None:
                out.append(f)
        self.send_frame_condition(None)
        while True:
            f = self.receive_frame()
            if f is None:
                break
            out.append(f)
        return torch.cat(out)
