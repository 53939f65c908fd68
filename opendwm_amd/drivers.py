"""Window drivers around the denoise loop: long sequences are generated window by window with the model's
[B, T, V] latent window held resident on the GPU.

  AutoregressiveDriver   CrossviewTemporalSD.autoregressive_inference_pipeline   src/dwm/pipelines/ctsd.py:1656-1833
  StreamingDriver        StreamingCrossviewTemporalSD.reset_streaming / send_frame_condition / receive_frame /
                         fifo_inference_pipeline                                  src/dwm/pipelines/ctsd.py:2009-2275

Both operate on *embedded* conditions (the CFG-doubled model kwargs, time on axis 1: what `get_conditions`,
ctsd.py:416-453, returns) - text encoders and the dataset stack are outside the hot path.
The per-window model + scheduler loop is `pipeline.CTSDDenoiser` (HIP kernels through the C ABI); the drivers
only plan the windows, carry latents between them (device tensors, no host round trip) and draw the noise
from the host generator in the reference's order, so a seeded run consumes the same random stream.

A driver is planned first (`plan()` returns the list of window calls: conditions clip, start/stop step, frame
to emit, what is carried over) and then executed; the plan is plain data and is what the CPU tests compare
with the restated reference control flow.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, List, Optional

import torch


def take_sequence_clip(item, start: int, stop: int):
    """Per-frame slice of one batch / condition entry (src/dwm/functional.py:172-181): scalars and <=1-D tensors
    pass through, tensors are cut on axis 1, nested lists per sample."""
    if isinstance(item, (int, float, bool, str)):
        return item
    if torch.is_tensor(item):
        return item if item.dim() <= 1 else item[:, start:stop]
    if isinstance(item, list):
        if not item or not all(isinstance(i, list) for i in item):
            raise TypeError("list entries must be non-empty lists of per-sample lists")
        return [i[start:stop] for i in item]
    raise TypeError("Unsupported type to take sequence clip.")


def latent_sequence_length(frames: int, vae_pre: int = 0, vae_stride: int = 1) -> int:
    """frames -> latent frames of a (temporal) VAE, ctsd.py:1113-1118"""
    if frames != 0 and frames % vae_stride != vae_pre:
        raise ValueError(f"{frames} vs {vae_pre} vs {vae_stride}")
    return (frames - vae_pre) // vae_stride + (1 if vae_pre > 0 else 0)


class LatentDecoder:
    """latents [B, t, V, C, h, w] -> images [(b t v), 3, H, W], the decode tail of inference_pipeline (ctsd.py:1606-1647):
    `vae.decode(latents / scaling_factor + shift_factor)` through memory_efficient_split_call chunks, then
    VaeImageProcessor.postprocess(output_type="pt") = (x / 2 + 0.5).clamp(0, 1).

    2-D VAE (opendwm_amd.vae.AutoencoderKL): frames are independent images "(b t v) c h w".  Temporal VAE
    (opendwm_amd.vae_cogvideox.AutoencoderKLCogVideoX): clips "(b v) c t h w"; in diffusion-forcing mode a single
    latent frame is decoded as [frame, zeros] and the first half of the 8 output frames is kept (:1611-1622)."""

    def __init__(self, vae, memory_efficient_batch: int = -1, postprocess: bool = True, group=None):
        """group: a torch.distributed process group - the images (2-D VAE) or per-view clips (temporal VAE) of one
        sample are decoded 1/R per rank and all-gathered, so every rank returns the whole batch (the decode half of
        the intra-sample sharding; the denoise half is CTSDDenoiser(frame_group=...))."""
        from .vae_cogvideox import AutoencoderKLCogVideoX
        self.vae, self.batch, self.postprocess, self.group = vae, memory_efficient_batch, postprocess, group
        self.is_temporal_vae = isinstance(vae, AutoencoderKLCogVideoX)

    def _sharded(self, fn, x: torch.Tensor) -> torch.Tensor:
        """fn over this rank's contiguous share of dim 0, all-gathered; shares are padded to equal size with item 0"""
        if self.group is None:
            return fn(x)
        import torch.distributed as dist
        R, r, n = dist.get_world_size(self.group), dist.get_rank(self.group), x.shape[0]
        per = -(-n // R)
        mine = x[r * per:(r + 1) * per]
        if mine.shape[0] < per:
            mine = torch.cat([mine, x[:1].expand(per - mine.shape[0], *x.shape[1:])])
        y = fn(mine.contiguous()).contiguous()
        parts = [torch.empty_like(y) for _ in range(R)]
        dist.all_gather(parts, y, group=self.group)
        return torch.cat(parts)[:n]

    def _split_call(self, x: torch.Tensor) -> torch.Tensor:
        """dwm.functional.memory_efficient_split_call, src/dwm/functional.py:184-193"""
        if self.batch == -1:
            return self.vae.decode(x, return_dict=False)[0]
        return torch.cat([self.vae.decode(c, return_dict=False)[0] for c in x.split(self.batch)])

    def __call__(self, latents: torch.Tensor, diffusion_forcing: bool = False) -> torch.Tensor:
        B, t, V = latents.shape[:3]
        cfgv = self.vae.config
        shift = cfgv.shift_factor if cfgv.shift_factor is not None else 0
        x = latents.to(self.vae.dtype) / cfgv.scaling_factor + shift
        if self.is_temporal_vae:
            clips = x.permute(0, 2, 3, 1, 4, 5).flatten(0, 1)                   # b t v c h w -> (b v) c t h w
            if diffusion_forcing:
                if t != 1:
                    raise ValueError("diffusion forcing decodes one queue slot at a time")
                img = self._sharded(lambda c: self.vae.decode(torch.cat([c, c * 0], 2), return_dict=False)[0].chunk(2, dim=2)[0], clips)
            else:
                img = self._sharded(self._split_call, clips)
            img = img.unflatten(0, (B, V)).permute(0, 3, 1, 2, 4, 5).flatten(0, 2)   # (b v) c t h w -> (b t v) c h w
        else:
            img = self._sharded(self._split_call if not diffusion_forcing else (lambda c: self.vae.decode(c, return_dict=False)[0]),
                                x.flatten(0, 2))
        return (img.float() / 2 + 0.5).clamp(0, 1) if self.postprocess else img


class LatentEncoder:
    """images [B, t, V, 3, H, W], already in the VAE's input range (VaeImageProcessor.preprocess: 2 x - 1) -> latents
    [B, t', V, C, h, w] = `(vae.encode(x).latent_dist.sample() - shift_factor) * scaling_factor` through
    memory_efficient_split_call chunks: the encode of the training step (ctsd.py:1196-1225, `.sample()`) and of the
    reference frames of the autoregressive pipeline (:1677-1703, `.mode()`).  2-D VAE: "(b t v) c h w" images; temporal
    VAE: "(b v) c t h w" clips."""

    def __init__(self, vae, memory_efficient_batch: int = -1, is_temporal_vae: Optional[bool] = None):
        if is_temporal_vae is None:
            from .vae_cogvideox import AutoencoderKLCogVideoX
            is_temporal_vae = isinstance(vae, AutoencoderKLCogVideoX)
        self.vae, self.batch, self.is_temporal_vae = vae, memory_efficient_batch, is_temporal_vae

    def __call__(self, images: torch.Tensor, sample: bool = True) -> torch.Tensor:
        B, t, V = images.shape[:3]
        cfgv = self.vae.config
        shift = cfgv.shift_factor if cfgv.shift_factor is not None else 0
        x = images.permute(0, 2, 3, 1, 4, 5).flatten(0, 1) if self.is_temporal_vae else images.flatten(0, 2)

        def enc(c):
            dist = self.vae.encode(c).latent_dist
            return ((dist.sample() if sample else dist.mode()) - shift) * cfgv.scaling_factor
        lat = enc(x) if self.batch == -1 else torch.cat([enc(c) for c in x.split(self.batch)])
        if self.is_temporal_vae:
            return lat.unflatten(0, (B, V)).permute(0, 3, 1, 2, 4, 5)          # (b v) c t h w -> b t v c h w
        return lat.unflatten(0, (B, t, V))


@dataclass
class Window:
    """One call of the per-window denoise loop."""
    clip: tuple                 # (first frame, last frame + 1) of the conditions
    start: int                  # first scheduler step index
    stop: int                   # one past the last step index
    take_time: int = 0          # diffusion forcing: queue slot whose frame is emitted
    reference: int = 0          # latent frames injected clean (full-sequence mode) / T in the DF steady state
    draws_noise: bool = True    # the window starts from fresh noise (host generator, full latent shape)
    carry: str = "none"         # what the next window receives: "all" | "tail" | "merge" | "none"
    push_noise: bool = False    # DF: after the merge, drop frame 0 and append one frame of fresh noise
    emit_from: int = 0          # full-sequence: first (pixel) frame of this window that is part of the result
    quarter: bool = False       # temporal VAE, first DF window: only the last quarter of the decoded images


class AutoregressiveDriver:
    """denoiser: pipeline.CTSDDenoiser-like object with run(latents, conditions, stop=, start=, image_latents=,
    reference_frame_count=, diffusion_forcing=, take_time=, clear_reference_frame_count=) -> latents.
    decode(latents [B, t, V, C, H, W], diffusion_forcing=bool) -> images with dim 0 = (b t v), e.g. a LatentDecoder;
    defaults to returning the latents flattened the same way, so the driver is usable (and testable) without a VAE."""

    def __init__(self, denoiser, inference_config: dict, diffusion_forcing: bool = False,
                 decode: Optional[Callable] = None, generator: Optional[torch.Generator] = None,
                 init_noise_sigma: float = 1.0, is_temporal_vae: bool = False):
        self.denoiser, self.cfg, self.df = denoiser, dict(inference_config), diffusion_forcing
        self.decode = decode if decode is not None else (lambda lat, diffusion_forcing=False: lat.flatten(0, 2))
        self.generator, self.init_noise_sigma = generator, init_noise_sigma
        self.is_temporal_vae = is_temporal_vae or bool(getattr(decode, "is_temporal_vae", False))

    # ---------------------------------------------------------------- planning (pure host logic)
    def plan(self, latent_frames: int, total_frame_count: int, have_reference: bool) -> List[Window]:
        cfg = self.cfg
        steps, seq = cfg["inference_steps"], cfg["sequence_length_per_iteration"]
        ref = cfg.get("reference_frame_count", 1)
        stride = seq - ref
        if stride <= 0:
            raise ValueError("sequence_length_per_iteration must exceed reference_frame_count")
        starts = list(range(0, total_frame_count - seq + 1, stride))
        lat_len = lambda n: latent_sequence_length(n, cfg.get("vae_pre", 0), cfg.get("vae_stride", 1))
        wins: List[Window] = []
        if not self.df:
            have = have_reference
            for n, i in enumerate(starts):
                r = ref if have else 0
                last = n == len(starts) - 1
                wins.append(Window(clip=(i, i + seq), start=0, stop=steps, reference=lat_len(r), emit_from=r,
                                   carry="none" if last else "tail"))
                have = have or not last
            return wins
        T = latent_frames
        if total_frame_count <= seq:
            raise ValueError("diffusion forcing needs more frames than one window")
        clear = cfg.get("clear_reference_frame_count", 0)
        if steps % (T - clear) != 0:
            raise ValueError("inference_steps must be a multiple of the queue length")
        spi = steps // (T - clear)
        # queue warm-up: every slot j denoised up to (steps - spi) - j*spi
        wins.append(Window(clip=(0, seq), start=0, stop=steps - spi, draws_noise=not have_reference, carry="all"))
        head = -1
        for n, i in enumerate(starts):
            r = ref
            if head < clear:
                r, head = T, head + 1
            more = n < len(starts) - 1
            wins.append(Window(clip=(i, i + seq), start=steps + (head - 1) * spi, stop=steps + head * spi, take_time=head,
                               reference=r, draws_noise=False, carry="merge", push_noise=head == clear and more,
                               quarter=self.is_temporal_vae and i == 0))
        for k in range(head + 1, T):                                             # flush the queue with the last clip
            wins.append(Window(clip=wins[-1].clip, start=steps + (k - 1) * spi, stop=steps + k * spi, take_time=k,
                               reference=T, draws_noise=False, carry="merge"))
        return wins

    # ---------------------------------------------------------------- execution
    def _randn(self, shape, device):
        return (torch.randn(tuple(shape), generator=self.generator) * self.init_noise_sigma).to(device)

    def run(self, latent_shape, conditions: Dict, total_frame_count: int, device,
            image_latents: Optional[torch.Tensor] = None) -> Dict:
        exc = self.cfg.get("autoregression_data_exception_for_take_sequence", [])
        B, T, V = latent_shape[:3]
        clear = self.cfg.get("clear_reference_frame_count", 0)
        images = []
        carried = image_latents
        for w in self.plan(T, total_frame_count, image_latents is not None):
            # `conditions`: the embedded model kwargs of the whole sequence (sliced per window), or a callable
            # (first frame, last frame + 1) -> model kwargs of that window.  The reference builds the conditions from each
            # window's clip of the batch (ctsd.py:1478-1488), so anything derived from neighbouring frames - the action ids
            # are pose differences - restarts at the window's first frame; `conditions_from_batch` reproduces that.
            cond = conditions(*w.clip) if callable(conditions) else \
                {k: (v if k in exc else take_sequence_clip(v, *w.clip)) for k, v in conditions.items()}
            if self.df:
                noise = self._randn(latent_shape, device) if (w.draws_noise or carried is None) else carried
                lat = self.denoiser.run(noise, cond, stop=w.stop, start=w.start, image_latents=carried,
                                        diffusion_forcing=True, take_time=w.take_time, clear_reference_frame_count=clear)
                if w.carry == "all":
                    carried = lat
                    continue
                img = self.decode(lat[:, w.take_time:w.take_time + 1], diffusion_forcing=True)
                images.append(img.chunk(4)[-1] if w.quarter else img)
                keep = (torch.arange(T, device=lat.device) <= w.take_time).view(1, T, 1, 1, 1, 1)
                carried = torch.where(keep, carried, lat)
                if w.push_noise:
                    carried = torch.cat([carried[:, 1:], self._randn((B, 1) + tuple(latent_shape[2:]), device)], 1)
            else:
                noise = self._randn(latent_shape, device)
                lat = self.denoiser.run(noise, cond, image_latents=carried, reference_frame_count=w.reference)
                images.append(self.decode(lat)[B * w.emit_from * V:])
                if w.carry == "tail":
                    n = latent_sequence_length(self.cfg.get("reference_frame_count", 1), self.cfg.get("vae_pre", 0),
                                               self.cfg.get("vae_stride", 1))
                    carried = lat[:, -n:]
        return {"images": torch.cat(images), "latents": carried}


def conditions_from_batch(batch: Dict, common_config: dict, inference_config: dict, latent_shape, device, dtype,
                          embed_text: Optional[Callable] = None) -> Callable:
    """-> callable(first frame, last frame + 1) for AutoregressiveDriver.run: the model kwargs of one window built from that
    window's clip of a dataset batch, as inference_pipeline does (ctsd.py:1478-1488 with the clip of :1735-1745):
    take_sequence_clip over the batch (minus `autoregression_data_exception_for_take_sequence`), prompts flattened and
    embedded (`embed_text(flat_prompts, parsed_shape, frames, view_count) -> (encoder_hidden_states, pooled_projections)`),
    conditions.build_conditions with CFG iff the inference config has a guidance_scale."""
    from .conditions import build_conditions, flatten_clip_text
    exc = inference_config.get("autoregression_data_exception_for_take_sequence", [])
    cfg = "guidance_scale" in inference_config

    def window_conditions(start: int, stop: int) -> Dict:
        clip = {k: (v if k in exc else take_sequence_clip(v, start, stop)) for k, v in batch.items()}
        ehs = pooled = None
        if embed_text is not None:
            flat, shape = flatten_clip_text(clip["clip_text"], do_classifier_free_guidance=cfg)
            ehs, pooled = embed_text(flat, shape, clip["pts"].shape[1], latent_shape[2])
        return build_conditions(common_config, latent_shape, clip, device, dtype, encoder_hidden_states=ehs,
                                pooled_projections=pooled, do_classifier_free_guidance=cfg, latents_shape=tuple(latent_shape))
    return window_conditions


class StreamingDriver:
    """Frame-in / frame-out FIFO generation with diffusion forcing (ctsd.py:2009-2275): conditions arrive one frame
    at a time; once `sequence_length_per_iteration` frames are gathered the queue is warmed up (one full run), after
    which every new frame costs inference_steps / T denoise steps and emits one frame.

    `frame_conditions`: embedded model kwargs of ONE frame ([2B, 1, V, ...]); entries named in
    `autoregression_condition_exception_for_take_sequence` replace instead of queueing."""

    def __init__(self, denoiser, inference_config: dict, decode: Optional[Callable] = None,
                 generator: Optional[torch.Generator] = None, init_noise_sigma: float = 1.0):
        self.denoiser, self.cfg = denoiser, dict(inference_config)
        self.decode = decode if decode is not None else (lambda lat, diffusion_forcing=False: lat.flatten(0, 2))
        self.generator, self.init_noise_sigma = generator, init_noise_sigma
        self.latent_shape = None

    def reset_streaming(self, latent_shape, device):
        T = latent_shape[1]
        if self.cfg["inference_steps"] % T != 0:
            raise ValueError("inference_steps must be a multiple of the queue length")
        self.latent_shape, self.device = tuple(latent_shape), device
        self.spi = self.cfg["inference_steps"] // T
        self.conditions: Dict = {}
        self.condition_count = 0
        self.latents = None
        self.frames: List[torch.Tensor] = []
        self.text_prompt_counter = 0
        self.prev_ego_transforms = None          # the last frame's ego pose: the action ids are pose differences

    def _randn(self, shape):
        return (torch.randn(tuple(shape), generator=self.generator) * self.init_noise_sigma).to(self.device)

    def _window(self, start: int, stop: int, take_time: int = 0):
        lat = self.denoiser.run(self.latents, self.conditions, stop=stop, start=start, image_latents=self.latents,
                                diffusion_forcing=True, take_time=take_time)
        if stop >= self.cfg["inference_steps"]:                                   # :2092-2101
            self.frames.append(self.decode(lat[:, take_time:take_time + 1], diffusion_forcing=True))
        return lat

    def _queue(self, frame_conditions: Dict, slide: bool):
        exc = self.cfg.get("autoregression_condition_exception_for_take_sequence", [])
        for k, v in frame_conditions.items():
            if k not in self.conditions or k in exc or not torch.is_tensor(v) or v.dim() <= 1:
                self.conditions[k] = v
            else:
                old = self.conditions[k][:, 1:] if slide else self.conditions[k]
                self.conditions[k] = torch.cat([old, v], 1)

    def send_frame_condition(self, frame_conditions: Optional[Dict]):
        if self.latent_shape is None:
            raise RuntimeError("reset_streaming() first")
        steps, T, seq = self.cfg["inference_steps"], self.latent_shape[1], self.cfg["sequence_length_per_iteration"]
        if frame_conditions is None:                                              # flush the queue
            if self.condition_count != seq:
                raise RuntimeError("flush before the queue was filled")
            for i in range(1, T):
                lat = self._window(steps + (i - 1) * self.spi, steps + i * self.spi, i)
                keep = (torch.arange(T, device=lat.device) <= i).view(1, T, 1, 1, 1, 1)
                self.latents = torch.where(keep, self.latents, lat)
        elif self.condition_count < seq:                                          # gathering
            self._queue(frame_conditions, slide=False)
            self.condition_count += 1
            if self.condition_count == seq:
                self.latents = self._randn(self.latent_shape)
                self.latents = self._window(0, steps)
        else:                                                                     # steady state: one frame in, one out
            self._queue(frame_conditions, slide=True)
            B = self.latent_shape[0]
            self.latents = torch.cat([self.latents[:, 1:], self._randn((B, 1) + self.latent_shape[2:])], 1)
            self.latents = self._window(steps - self.spi, steps)

    def send_frame(self, frame_batch: Optional[Dict], common_config: dict, embed_text: Optional[Callable] = None,
                   dtype=torch.bfloat16):
        """One frame of a dataset batch in (None = flush), the ingestion half of send_frame_condition (ctsd.py:2134-2160):
        model kwargs through conditions.build_conditions in streaming mode (action ids against the previous frame's ego
        pose), CFG iff the inference config has a guidance_scale; the prompts are embedded only every
        `text_prompt_interval`-th frame - `embed_text(flat_prompts, parsed_shape, view_count) -> (encoder_hidden_states
        [B', 1, V, L, D], pooled_projections [B', 1, V, P])`, e.g. the callers' encoders + conditions.assemble_sd3_text -
        and the newest queued embeddings are reused in between."""
        if frame_batch is None:
            return self.send_frame_condition(None)
        from .conditions import build_conditions, flatten_clip_text
        cfg = "guidance_scale" in self.cfg
        ehs = pooled = None
        if self.text_prompt_counter == 0 and embed_text is not None:
            flat, shape = flatten_clip_text(frame_batch["clip_text"], do_classifier_free_guidance=cfg)
            ehs, pooled = embed_text(flat, shape, self.latent_shape[2])
        cond = build_conditions(common_config, (self.latent_shape[0], 1) + tuple(self.latent_shape[2:]), frame_batch, self.device, dtype,
                                encoder_hidden_states=ehs, pooled_projections=pooled, streaming_mode=True,
                                prev_ego_transforms=self.prev_ego_transforms, do_classifier_free_guidance=cfg)
        if "ego_transforms" in frame_batch:
            self.prev_ego_transforms = frame_batch["ego_transforms"]
        if self.text_prompt_counter > 0:
            cond["encoder_hidden_states"] = self.conditions["encoder_hidden_states"][:, -1:]
            cond["pooled_projections"] = self.conditions["pooled_projections"][:, -1:]
        self.text_prompt_counter = (self.text_prompt_counter + 1) % self.cfg.get("text_prompt_interval", 1)
        self.send_frame_condition(cond)

    def receive_frame(self):
        return self.frames.pop(0) if self.frames else None

    def fifo(self, latent_shape, conditions: Dict, total_frame_count: int, device) -> torch.Tensor:
        if total_frame_count <= self.cfg["sequence_length_per_iteration"]:
            raise ValueError("fifo generation needs more frames than one window")
        exc = self.cfg.get("autoregression_data_exception_for_take_sequence", [])
        out = []
        self.reset_streaming(latent_shape, device)
        for i in range(total_frame_count):
            self.send_frame_condition({k: (v if k in exc else take_sequence_clip(v, i, i + 1)) for k, v in conditions.items()})
            f = self.receive_frame()
            if f is not None:
                out.append(f)
        self.send_frame_condition(None)
        while (f := self.receive_frame()) is not None:
            out.append(f)
        return torch.cat(out)
