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

    def __init__(self, model, guidance_scale: float = 4.0, inference_steps: int = 40, shift: float = 3.0,
                 cfg_group=None, frame_group=None):
        """frame_group: a torch.distributed process group whose R ranks hold T/R frames each of ONE sample
        (opendwm_amd.sharding: one all-to-all before and after every temporal block; everything else is local).  Every
        rank passes the same full latents / conditions to prepare() and gets the full result() back.

        cfg_group: a torch.distributed process group of size 2 -> classifier-free-guidance split: the
        two halves of the CFG batch are independent inside the model, so rank 0 of the group runs the unconditional
        half and rank 1 the conditional half of ONE sample, the halves of the prediction are exchanged with one
        all-gather per step (2.2 MB at config 3; RCCL over xGMI) and both ranks apply the same guidance + scheduler
        update, keeping bit-identical latents.  Halves the per-sample latency; throughput scaling stays with replicas."""
        self.model = model
        self.cfg_group = cfg_group
        self.cfg_rank = 0
        self.frame_shard = None
        if frame_group is not None:
            from .sharding import FrameShard
            self.frame_shard = FrameShard(frame_group)
        if cfg_group is not None:
            import torch.distributed as dist
            if dist.get_world_size(cfg_group) != 2:
                raise ValueError("the CFG split needs a process group of exactly two ranks")
            self.cfg_rank = dist.get_rank(cfg_group)
        self.guidance_scale = guidance_scale
        self.schedule = FlowMatchEulerSchedule(shift=shift).set_timesteps(inference_steps)
        self.inference_steps = inference_steps
        self._ts_dev = None
        self.use_graph = False
        self._graph = None
        self._side = None

    def prepare(self, latents: torch.Tensor, conditions: Dict[str, torch.Tensor],
                image_latents: Optional[torch.Tensor] = None, reference_frame_count: int = 0,
                diffusion_forcing: bool = False, take_time: int = 0, clear_reference_frame_count: int = 0):
        dev = latents.device
        self._graph = None                                             # buffers below are re-created: capture again
        self.diffusion_forcing = diffusion_forcing
        self.take_time = take_time
        if diffusion_forcing and image_latents is not None:
            latents = image_latents                                   # ctsd.py:1470-1471
            image_latents = None
        B, T = latents.shape[:2]
        self.total_frames, self.t0 = T, 0
        self.ref = reference_frame_count if image_latents is not None else 0
        if diffusion_forcing:
            if self.inference_steps % (T - clear_reference_frame_count) != 0:
                raise ValueError("inference_steps must be a multiple of the frame count in diffusion-forcing mode")
            self.spi = self.inference_steps // (T - clear_reference_frame_count)
        if hasattr(self.model, "frame_shard"):
            self.model.frame_shard = self.frame_shard
        elif self.frame_shard is not None:
            raise ValueError("this model has no frame-sharded forward")
        if self.frame_shard is not None:                               # keep this rank's frames of everything per-frame
            if self.use_graph:
                raise NotImplementedError("frame sharding is not captured into a HIP graph")
            t0, t1 = self.frame_shard.frame_range(T)
            self.t0 = t0
            latents = latents[:, t0:t1]
            if image_latents is not None:
                image_latents = image_latents[:, t0:t1]
            self.ref = min(max(self.ref - t0, 0), t1 - t0)
            conditions = {k: (v[:, t0:t1] if torch.is_tensor(v) and k not in self.NO_FRAME_AXIS and v.dim() >= 2 and v.shape[1] == T
                              else v) for k, v in conditions.items()}
        self.latents = latents.to(torch.float32).contiguous().clone()
        self.image_latents = None if image_latents is None else image_latents.to(torch.float32).to(dev)
        # model input / conditions / prediction in the model's compute dtype: bf16, or fp32 for the accuracy path
        # (`model.compute_dtype = torch.float32`)
        self.cd = cd = getattr(self.model, "compute_dtype", bf16)
        self.model_in = torch.empty((2 * B, *latents.shape[1:]), dtype=cd, device=dev)
        self._refresh_model_in()
        self.conditions = {k: (v.to(cd) if torch.is_tensor(v) and v.is_floating_point() and k != "added_time_ids" else v)
                           for k, v in conditions.items()}
        if self.cfg_group is not None:                                 # this rank's half of every CFG-doubled condition
            r = self.cfg_rank
            self.conditions = {k: (v[r * B:(r + 1) * B].contiguous() if torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == 2 * B else v)
                               for k, v in self.conditions.items()}
            self._pred_full = torch.empty((2 * B, *latents.shape[1:]), dtype=cd, device=dev)
        self._ts_dev = self.schedule.timesteps.to(dev)
        self._sig_dev = self.schedule.sigmas.to(dev)
        return self

    # conditions without a frame axis at dim 1 (crossview_temporal_dit.py:372-391)
    NO_FRAME_AXIS = ("disable_crossview", "disable_temporal", "crossview_attention_mask", "crossview_attention_index")

    def _refresh_model_in(self):
        B = self.latents.shape[0]
        lat16 = self.latents if self.cd == torch.float32 else ops.cast_bf16(self.latents)
        self.model_in[:B].copy_(lat16)
        self.model_in[B:].copy_(lat16)
        self._inject_reference()

    def _inject_reference(self):
        if self.ref > 0:                                              # clean reference frames, every step
            B = self.latents.shape[0]
            r16 = self.image_latents[:, :self.ref].to(self.cd)
            self.model_in[:B, :self.ref].copy_(r16)
            self.model_in[B:, :self.ref].copy_(r16)

    def _step_inputs(self, i: int):
        """(timesteps [2B,T,V] device fp32, sigma step: host float or device fp32 [B,T,V]) of step i"""
        B, T, V = self.latents.shape[:3]
        if self.diffusion_forcing:
            j = self.t0 + torch.arange(T)                              # this rank's frames of the sample
            idx = torch.minimum(torch.full((T,), i - self.take_time * self.spi), torch.clamp(i - j * self.spi, min=0))
            idx = idx.to(self._ts_dev.device)
            ts = self._ts_dev[idx].view(1, T, 1).expand(2 * B, T, V)
            in_range = (i - j * self.spi >= 0).to(self._sig_dev.device)
            dsig = torch.where(in_range, self._sig_dev[idx + 1] - self._sig_dev[idx], torch.zeros((), device=idx.device))
            return ts, dsig.view(1, T, 1).expand(B, T, V).contiguous().float()
        ts = self._ts_dev[i].expand(2 * B, T, V)
        if self.ref > 0:
            ts = ts.clone()
            ts[:, :self.ref] = 0
        return ts, float(self.schedule.sigmas[i + 1] - self.schedule.sigmas[i])

    def _step_body(self, ts: torch.Tensor, dsig):
        if self.cfg_group is None:
            out, _, _ = self.model(self.model_in, ts, **self.conditions)
            pred = out[0]
        else:
            import torch.distributed as dist
            B, r = self.latents.shape[0], self.cfg_rank
            out, _, _ = self.model(self.model_in[r * B:(r + 1) * B], ts[r * B:(r + 1) * B], **self.conditions)
            dist.all_gather(list(self._pred_full.split(B)), out[0].contiguous(), group=self.cfg_group)
            pred = self._pred_full
        if torch.is_tensor(dsig):
            ops.cfg_euler_step(pred, self.latents, self.guidance_scale, dsig, model_in=self.model_in,
                               group_elems=self.latents[0, 0, 0].numel())
        else:
            ops.cfg_euler_step(pred, self.latents, self.guidance_scale, dsig, model_in=self.model_in)
        self._inject_reference()

    def step(self, i: int):
        """One denoise step = model forward at the CFG batch + guidance combine + scheduler update."""
        ts, dsig = self._step_inputs(i)
        if self.use_graph:
            self._graph_step(ts, dsig)
        else:
            self._step_body(ts, dsig)

    # ---- whole-step HIP graph: the ~600 launches of one step (model forward, CFG combine, scheduler
    # update, reference re-injection) are captured once per prepare() and replayed; per-step inputs (timesteps, sigma
    # steps) live in static device buffers that are refreshed before each replay.
    def enable_graph(self, on: bool = True):
        self.use_graph = on
        self._graph = None
        return self

    def _graph_step(self, ts: torch.Tensor, dsig):
        B, T, V = self.latents.shape[:3]
        if not torch.is_tensor(dsig):
            dsig = torch.full((B, T, V), dsig, dtype=torch.float32, device=self.latents.device)
        if self._graph is None:
            self._g_ts = ts.to(torch.float32).contiguous().clone()
            self._g_dsig = dsig.clone()
            # one eager pass on a side stream fills every cache the forward keeps (packed weights, adapter residuals,
            # scratch buffers) and warms the allocator; the state it touched is restored before the capture
            keep_lat, keep_in = self.latents.clone(), self.model_in.clone()
            if self._side is None:
                self._side = torch.cuda.Stream(device=self.latents.device)
            side = self._side
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self._step_body(self._g_ts, self._g_dsig)
            torch.cuda.current_stream().wait_stream(side)
            self.latents.copy_(keep_lat)
            self.model_in.copy_(keep_in)
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph):
                self._step_body(self._g_ts, self._g_dsig)
        self._g_ts.copy_(ts)
        self._g_dsig.copy_(dsig)
        self._graph.replay()

    def result(self) -> torch.Tensor:
        out = self.latents
        if self.ref > 0:                                              # ctsd.py:1623-1627
            out = torch.cat([self.image_latents[:, :self.ref], self.latents[:, self.ref:]], 1)
        if self.frame_shard is not None:
            out = self.frame_shard.gather_frames(out, 1)               # every rank returns the whole sample
        return out

    def run(self, latents: torch.Tensor, conditions: Dict[str, torch.Tensor], stop: Optional[int] = None,
            start: int = 0, **prepare_kw):
        self.prepare(latents, conditions, **prepare_kw)
        for i in range(start, self.inference_steps if stop is None else stop):
            self.step(i)
        return self.result()


# ------------------------------------------------------------------------------------------ training
def flow_match_train_sigmas(num_train_timesteps: int = 1000, shift: float = 3.0) -> torch.Tensor:
    """sigmas[i] of the *training* FlowMatchEulerDiscreteScheduler (timesteps[i] = 1000 * sigmas[i], descending);
    ctsd.py:1255-1270 indexes it with floor(u * num_train_timesteps)."""
    s = torch.linspace(1, num_train_timesteps, num_train_timesteps).flip(0) / num_train_timesteps
    return (shift * s / (1 + (shift - 1) * s)).float()


def sample_timestep_indices(shape, generator: Optional[torch.Generator] = None, weighting_scheme: str = "logit_normal",
                            num_train_timesteps: int = 1000, logit_mean: float = 0.0, logit_std: float = 1.0) -> torch.Tensor:
    """sd3_compute_density_for_timestep_sampling + `(u * num_train_timesteps).long()` (ctsd.py:1256-1262)."""
    if weighting_scheme == "logit_normal":
        u = torch.sigmoid(torch.randn(shape, generator=generator) * logit_std + logit_mean)
    elif weighting_scheme in ("none", "uniform"):
        u = torch.rand(shape, generator=generator)
    else:
        raise NotImplementedError(weighting_scheme)
    return (u * num_train_timesteps).long().clamp_(max=num_train_timesteps - 1)


def make_input_for_prediction(noisy_input: torch.Tensor, latents: torch.Tensor, timesteps: torch.Tensor, training_config: dict,
                              common_config: dict, generator: Optional[torch.Generator] = None, reference_latent_count=None):
    """CrossviewTemporalSD.try_make_input_for_prediction (ctsd.py:619-741): the training task mixer.  Host random draws in the
    reference's order (CPU generator), tensors stay on the latents' device.

    frame_prediction_style None: nothing changes.  "diffusion_forcing": per-sample image task (temporal blocks disabled, with
    probability image_generation_ratio), otherwise the reference-frame scale / offset augmentation of the noisy input.
    "ctsd": generation vs prediction tasks - for prediction the first `reference_latent_count` frames (an int, or a
    {count: probability} dict) are shown clean (timestep 0), all of them or a random subset.
    Returns (model input, timesteps, extra conditions or None, reference_frame_indicator [B, T, V])."""
    import itertools
    dev = latents.device
    B, T, V = noisy_input.shape[:3]
    rf_scale, rf_offset = 1, 0
    if "reference_frame_scale_std" in training_config:
        rf_scale = (torch.randn(latents.shape[:2], generator=generator) * training_config["reference_frame_scale_std"] + 1) \
            .view(B, T, 1, 1, 1, 1).to(dev)
    if "reference_frame_offset_std" in training_config:
        rf_offset = (torch.randn(latents.shape[:2], generator=generator) * training_config["reference_frame_offset_std"]) \
            .view(B, T, 1, 1, 1, 1).to(dev)
    style = common_config.get("frame_prediction_style", None)
    indicator = torch.zeros(B, T, V, dtype=torch.bool, device=dev)
    if style is None:
        return noisy_input, timesteps, None, indicator
    if style == "diffusion_forcing":
        disable_temporal = torch.rand((B, 1, 1), generator=generator) < training_config.get("image_generation_ratio", 0.0)
        made = torch.where(disable_temporal.view(B, 1, 1, 1, 1, 1).to(dev), noisy_input, noisy_input * rf_scale + rf_offset)
        return made, timesteps, {"disable_temporal": disable_temporal.to(dev)}, indicator
    if style != "ctsd":
        raise ValueError("Unknown frame prediction type")
    generation = torch.rand((B, 1, 1), generator=generator) < training_config.get("generation_task_ratio", 0.0)
    disable_temporal = torch.logical_and(
        torch.rand((B, 1, 1), generator=generator) < training_config.get("image_generation_ratio", 0.0), generation)
    all_visible = torch.rand((B, 1, 1), generator=generator) < training_config.get("all_reference_visible_ratio", 0.0)
    partial = torch.rand((B, T, V), generator=generator) < training_config.get("reference_visible_rate", 1.0)
    if isinstance(reference_latent_count, int):
        count = reference_latent_count * torch.ones((B, 1, 1), dtype=torch.int32)
    elif isinstance(reference_latent_count, dict):
        counts = torch.tensor([int(i) for i in reference_latent_count.keys()], dtype=torch.int32)
        cum = torch.tensor(list(itertools.accumulate(reference_latent_count.values())))
        count = counts[torch.searchsorted(cum, torch.rand((B, 1, 1), generator=generator))]
    else:
        raise NotImplementedError("Un implemented dynamic reference frame count")
    in_count = torch.arange(T, dtype=torch.int32).view(1, T, 1).repeat(B, 1, V) < count
    indicator = torch.logical_and(torch.logical_and(torch.logical_not(generation), torch.logical_or(all_visible, partial)), in_count)
    made = torch.where(indicator.view(B, T, V, 1, 1, 1).to(dev), latents * rf_scale + rf_offset, noisy_input)
    made_timesteps = torch.where(indicator.to(timesteps.device), torch.zeros_like(timesteps), timesteps)
    return made, made_timesteps, {"disable_temporal": disable_temporal.to(dev)}, indicator


def freeze_modules(model, pattern: str):
    """training_config["freezing_pattern"] (ctsd.py:1014-1022): requires_grad_(False) on every module whose qualified name
    the regex matches (re.match: anchored at the start); returns the names"""
    import re
    pat, names = re.compile(pattern), []
    for name, module in model.named_modules():
        if pat.match(name) is not None:
            module.requires_grad_(False)
            names.append(name)
    return names


class CTSDTrainer:
    """The SD 3 branch of CrossviewTemporalSD.train_step (ctsd.py:1195-1437) on latents that are already
    VAE-encoded and conditions that are already embedded:

        idx ~ logit-normal, sigma = sigmas[idx], t = 1000 sigma           :1255-1266
        x_t = sigma * noise + (1 - sigma) * x0,  target = x0               :1267-1272
        pred = model(x_t, t, conditions);  x0_hat = pred * (-sigma) + x_t  :1355-1360
        loss = mse(x0_hat.float(), x0.float()) * coef                      :1368-1370
        loss.backward(); [clip_grad_norm_]; optimizer.step(); zero_grad    :1401-1432

    With the SD 2.1 UNet as the model the trainer takes the reference's other branch (ctsd.py:1240-1253):

        t ~ randint(0, num_train_timesteps) per sample (per frame with diffusion forcing), from the pipeline generator
        x_t = train_scheduler.add_noise(x0, noise, t)          (schedulers.DDPMScheduler, tensor timesteps)
        target = noise ("epsilon") | train_scheduler.get_velocity(x0, noise, t) ("v_prediction")
        loss = mse(model(x_t, t, conditions).float(), target.float()) * coef

    `ddp=True` wraps the model in torch DistributedDataParallel (ctsd.py:1051-1054): the block Functions of
    opendwm_amd.train hand their parameter gradients to autograd block by block, so the bucketed RCCL
    all-reduce overlaps the rest of the backward.  The buckets are fp32 by default, as the reference's DDP all-reduces them
    (ctsd.py:1051-1054), 200 MB each; `ddp_comm_dtype=torch.bfloat16` is the opt-in throughput mode (torch's
    bf16_compress_hook: 7.6 GB instead of 15.1 GB per step at 3.78 B parameters - the ring all-reduce is bound by the xGMI
    links, SURVEY.md s5 - at the price of a bf16 rounding of every averaged gradient, accumulated micro-steps included).

    training_config keys honoured as the reference does: "freezing_pattern" (regex over module names, ctsd.py:1014-1022),
    "gradient_accumulation_steps" (optimizer step every k-th call, :1401-1432; the micro-steps in between run under DDP's
    no_sync, so one all-reduce per optimizer step carries the accumulated gradient - the same sum the reference gets
    with an all-reduce per micro-step), "max_norm_for_grad_clip", "enable_grad_scaler" (torch.amp.GradScaler around our fused
    AdamW: :1040-1048, :1401-1432).  `lr_scheduler` (a callable optimizer -> scheduler, or
    a scheduler) is stepped once per train_step (:1434-1435)."""

    def __init__(self, model, lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2,
                 shift: float = 3.0, num_train_timesteps: int = 1000, loss_coef: float = 1.0,
                 max_grad_norm: Optional[float] = None, weighting_scheme: str = "logit_normal", ddp: bool = False,
                 ddp_kwargs: Optional[dict] = None, common_config: Optional[dict] = None, training_config: Optional[dict] = None,
                 reference_latent_count=0, lr_scheduler=None, ddp_comm_dtype: Optional[torch.dtype] = None,
                 train_scheduler=None):
        """common_config["frame_prediction_style"] (None | "diffusion_forcing" | "ctsd") and training_config select the
        training task mix of `make_input_for_prediction`; with "diffusion_forcing" every frame draws its own timestep
        (ctsd.py:1232-1237)."""
        from . import train as _train
        self.model = model.train()
        self.wrapper = model
        self.common_config, self.training_config = dict(common_config or {}), dict(training_config or {})
        self.frozen_modules = freeze_modules(model, self.training_config["freezing_pattern"]) \
            if "freezing_pattern" in self.training_config else []
        self.ddp = bool(ddp)
        self.ddp_comm_dtype = ddp_comm_dtype
        if ddp:
            dev = next(model.parameters()).device
            kw = dict(device_ids=[dev.index] if dev.type == "cuda" else None, gradient_as_bucket_view=True, bucket_cap_mb=200)
            kw.update(ddp_kwargs or {})
            self.wrapper = torch.nn.parallel.DistributedDataParallel(model, **kw)
            if ddp_comm_dtype == bf16:
                from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
                self.wrapper.register_comm_hook(None, default_hooks.bf16_compress_hook)
            elif ddp_comm_dtype not in (None, torch.float32):
                raise ValueError("ddp_comm_dtype: torch.bfloat16, torch.float32 or None")
        # every parameter, frozen ones included, as the reference builds it (ctsd.py:1089-1092): state-dict indices match
        self.optimizer = _train.AdamW(model.parameters(), lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self.lr_scheduler = lr_scheduler(self.optimizer) if callable(lr_scheduler) else lr_scheduler
        self.global_step = 0
        self.sigmas = flow_match_train_sigmas(num_train_timesteps, shift)
        self.num_train_timesteps, self.loss_coef = num_train_timesteps, loss_coef
        self.max_grad_norm = self.training_config.get("max_norm_for_grad_clip", max_grad_norm)
        # "enable_grad_scaler" (every shipped training config sets it: ctsd.py:1040-1048, :1401-1432): dynamic loss scaling with
        # torch.amp.GradScaler's defaults and state.  The compute type here is bf16, whose exponent range is fp32's, so
        # the scale changes no rounding (a power of two) - what the mode keeps is the reference's control flow: gradients are
        # unscaled before the clip, and a step whose gradients hold an inf / nan is SKIPPED and halves the scale.
        self.grad_scaler = None
        if self.training_config.get("enable_grad_scaler", False):
            # ctsd.py:1040-1048: a plain GradScaler unless torch.distributed is initialised AND the framework is fsdp (then the
            # reference takes ShardedGradScaler, which belongs to the FSDP wrap this package does not build) - a single-process run of
            # a shipped fsdp config gets the plain scaler, as in the reference
            import torch.distributed as _dist
            if (self.common_config.get("distribution_framework", "ddp") != "ddp" and _dist.is_available() and _dist.is_initialized()):
                raise NotImplementedError("enable_grad_scaler with distribution_framework != 'ddp' in a distributed run "
                                          "(ShardedGradScaler / FSDP) - SURVEY.md s2")
            dev_type = next(model.parameters()).device.type
            self.grad_scaler = torch.amp.GradScaler(dev_type)
        self.weighting_scheme = weighting_scheme
        self.reference_latent_count = reference_latent_count
        # SD 2.1 branch (ctsd.py:1240-1253): the model is the UNet -> DDPM noising, epsilon / v_prediction target
        from .unet import UNetCrossviewTemporalConditionModel
        self.is_unet = isinstance(model, UNetCrossviewTemporalConditionModel)
        if self.is_unet and train_scheduler is None:
            from .schedulers import DDPMScheduler
            train_scheduler = DDPMScheduler(num_train_timesteps=num_train_timesteps)
        self.train_scheduler = train_scheduler

    def draw_condition_masks(self, batch_size: int, generator: Optional[torch.Generator] = None) -> Dict[str, torch.Tensor]:
        """the per-sample condition dropout masks of the training step in the reference's draw order (ctsd.py:1278-1301):
        keyword arguments of conditions.build_conditions (text_condition_mask is a list, for the caller's text encoders)"""
        tc, draw = self.training_config, lambda: torch.rand((batch_size,), generator=generator)
        out = {"text_condition_mask": (draw() < tc.get("text_prompt_condition_ratio", 1.0)).tolist(),
               "_3dbox_condition_mask": draw() < tc.get("3dbox_condition_ratio", 1.0),
               "hdmap_condition_mask": draw() < tc.get("hdmap_condition_ratio", 1.0),
               "action_condition_mask": draw() < tc.get("action_condition_ratio", 1.0)}
        if self.common_config.get("explicit_view_modeling", False):
            out["explicit_view_modeling_mask"] = draw() < tc.get("explicit_view_modeling_ratio", 1.0)
        return out

    def draw_training_inputs(self, latents_shape, generator: Optional[torch.Generator] = None):
        """(noise, timestep_indices, condition masks) of one training step, drawn in the reference's order (ctsd.py:1229-1301:
        noise from the pipeline generator, timestep density from the global generator, then the dropout masks); pass the
        first two to loss() / train_step() - whose task mixer continues on the same generator - and the masks to
        conditions.build_conditions."""
        B, T = latents_shape[:2]
        noise = torch.randn(tuple(latents_shape), generator=generator)
        per_frame = self.common_config.get("frame_prediction_style") == "diffusion_forcing"
        if getattr(self, "is_unet", False): # :1241-1244: integer timesteps from the SAME generator, right after the noise
            idx = torch.randint(0, self.num_train_timesteps, (B, T) if per_frame else (B,), generator=generator)
        else:
            idx = sample_timestep_indices((B, T) if per_frame else (B,), None, self.weighting_scheme, self.num_train_timesteps)
        return noise, idx, self.draw_condition_masks(B, generator)

    def make_training_pair(self, latents: torch.Tensor, generator: Optional[torch.Generator] = None,
                           timestep_indices: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None):
        """returns (noisy_latents, timesteps [B,T,V], sigmas [B,1|T,1,1,1,1], noise); host RNG like the reference (CPU
        generator).  One timestep per sample, or per (sample, frame) in the diffusion-forcing style (ctsd.py:1232-1272)."""
        B, T, V = latents.shape[:3]
        if noise is None:
            noise = torch.randn(latents.shape, generator=generator)
        if timestep_indices is None:
            per_frame = getattr(self, "common_config", {}).get("frame_prediction_style") == "diffusion_forcing"
            timestep_indices = sample_timestep_indices((B, T) if per_frame else (B,), generator, self.weighting_scheme, self.num_train_timesteps)
        sig = self.sigmas[timestep_indices.cpu()]
        F = sig.shape[1] if sig.dim() == 2 else 1
        timesteps = (sig * self.num_train_timesteps).view(B, F, 1).expand(B, T, V).contiguous()
        dev = latents.device
        sig_b = sig.view(B, F, 1, 1, 1, 1).to(dev)
        noise = noise.to(dev)
        noisy = sig_b * noise + (1.0 - sig_b) * latents.float()
        return noisy, timesteps.to(dev), sig_b, noise

    def make_unet_training_pair(self, latents: torch.Tensor, generator: Optional[torch.Generator] = None,
                                timesteps: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None):
        """SD 2.1 branch (ctsd.py:1229-1253, 1273-1276): returns (noisy_latents, timesteps [B,T,V] int64, target, noise)"""
        B, T, V = latents.shape[:3]
        if noise is None:
            noise = torch.randn(latents.shape, generator=generator)
        if timesteps is None:
            per_frame = self.common_config.get("frame_prediction_style") == "diffusion_forcing"
            timesteps = torch.randint(0, self.num_train_timesteps, (B, T) if per_frame else (B,), generator=generator)
        dev = latents.device
        noise, timesteps = noise.to(dev), timesteps.to(dev)
        sch = self.train_scheduler
        noisy = sch.add_noise(latents.float(), noise, timesteps)
        pt = sch.config.prediction_type
        if pt == "epsilon":
            target = noise
        elif pt == "v_prediction":
            target = sch.get_velocity(latents.float(), noise, timesteps)
        else:
            raise Exception("Unknown training target of the UNet.")
        while timesteps.dim() < 3:                                                     # :1273-1276
            timesteps = timesteps.unsqueeze(-1).repeat_interleave(latents.shape[timesteps.dim()], -1)
        return noisy, timesteps, target.float(), noise

    def _unet_loss(self, latents, conditions, generator, timesteps, noise) -> torch.Tensor:
        noisy, timesteps, target, _ = self.make_unet_training_pair(latents, generator, timesteps, noise)
        noisy, timesteps, extra, reference = make_input_for_prediction(
            noisy, latents.float(), timesteps, self.training_config, self.common_config, generator, self.reference_latent_count)
        cond = {k: (v.to(bf16) if torch.is_tensor(v) and v.is_floating_point() and k != "added_time_ids" else v)
                for k, v in conditions.items()}
        if extra is not None:
            cond.update(extra)
        pred = self.wrapper(noisy.to(bf16), timesteps, **cond)[0][0].float()           # :1358-1360: sd_pred[0] as it is
        if self.training_config.get("disable_reference_frame_loss", False):          # :1363-1367
            keep = ~reference.view(*pred.shape[:3], 1, 1, 1).to(pred.device)
            pred, target = pred * keep, target * keep
        return torch.nn.functional.mse_loss(pred, target, reduction="mean") * self.loss_coef

    def loss(self, latents: torch.Tensor, conditions: Dict[str, torch.Tensor], generator=None,
             timestep_indices=None, noise=None) -> torch.Tensor:
        if getattr(self, "is_unet", False):
            return self._unet_loss(latents, conditions, generator, timestep_indices, noise)
        noisy, timesteps, sig, _ = self.make_training_pair(latents, generator, timestep_indices, noise)
        noisy, timesteps, extra, reference = make_input_for_prediction(
            noisy, latents.float(), timesteps, self.training_config, self.common_config, generator, self.reference_latent_count)
        cond = {k: (v.to(bf16) if torch.is_tensor(v) and v.is_floating_point() and k != "added_time_ids" else v)
                for k, v in conditions.items()}
        if extra is not None:
            cond.update(extra)
        pred = self.wrapper(noisy.to(bf16), timesteps, **cond)[0][0]
        x0_hat, target = pred.float() * (-sig) + noisy, latents.float()
        if self.training_config.get("disable_reference_frame_loss", False):          # ctsd.py:1363-1367
            keep = ~reference.view(*x0_hat.shape[:3], 1, 1, 1).to(x0_hat.device)
            x0_hat, target = x0_hat * keep, target * keep
        return torch.nn.functional.mse_loss(x0_hat, target, reduction="mean") * self.loss_coef

    # ---- checkpoint / resume in the reference's on-disk layout (ctsd.py:1134-1155 save_checkpoint, :988-992 and
    # :1093-1096 resume; src/dwm/distributed.py): <output>/checkpoints/<step>.pth = the model's state dict (reference
    # keys), <output>/optimizer/<step>.pth = the optimizer state in torch.optim.AdamW's format.  Rank 0 writes.
    def save_checkpoint(self, output_path: str, steps: int) -> None:
        import os
        import torch.distributed as dist
        if dist.is_initialized() and dist.get_rank() != 0:
            return
        os.makedirs(os.path.join(output_path, "checkpoints"), exist_ok=True)
        os.makedirs(os.path.join(output_path, "optimizer"), exist_ok=True)
        torch.save({k: v.detach().cpu() for k, v in self.model.state_dict().items()},
                   os.path.join(output_path, "checkpoints", f"{steps}.pth"))
        osd = self.optimizer.state_dict()
        osd["state"] = {i: {k: (v.detach().cpu() if torch.is_tensor(v) else v) for k, v in st.items()} for i, st in osd["state"].items()}
        torch.save(osd, os.path.join(output_path, "optimizer", f"{steps}.pth"))

    def load_checkpoint(self, output_path: str, resume_from: int) -> None:
        """model + optimizer state of step `resume_from`; the call counter continues there, so the gradient-accumulation
        phase `(global_step + 1) % k` lines up with the reference loop, which passes the resumed global_step
        (ctsd.py:1401-1404).  The lr_scheduler is NOT restored: the reference re-creates it on every start too."""
        import os
        self.global_step = int(resume_from)
        self.model.load_state_dict(torch.load(os.path.join(output_path, "checkpoints", f"{resume_from}.pth"),
                                              map_location="cpu", weights_only=True))
        self.optimizer.load_state_dict(torch.load(os.path.join(output_path, "optimizer", f"{resume_from}.pth"),
                                                  map_location="cpu", weights_only=True))

    def train_step(self, latents: torch.Tensor, conditions: Dict[str, torch.Tensor], generator=None,
                   timestep_indices=None, noise=None, global_step: Optional[int] = None) -> torch.Tensor:
        """one call of the reference's train_step (ctsd.py:1195-1437); `global_step` defaults to the number of calls so far"""
        import contextlib
        gs = self.global_step if global_step is None else global_step
        k = self.training_config.get("gradient_accumulation_steps")
        should_optimize = k is None or (gs + 1) % k == 0                                 # :1401-1404
        sync = contextlib.nullcontext() if (should_optimize or not self.ddp) else self.wrapper.no_sync()
        scaler = getattr(self, "grad_scaler", None)
        with sync:
            loss = self.loss(latents, conditions, generator, timestep_indices, noise)
            (loss if scaler is None else scaler.scale(loss)).backward()                   # :1401-1404
        if should_optimize:
            if self.max_grad_norm is not None:
                if scaler is not None:
                    scaler.unscale_(self.optimizer)                                      # :1411-1413
                torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.max_grad_norm)
            if scaler is not None:
                scaler.step(self.optimizer)                                              # :1426-1428 (skips on inf / nan)
                scaler.update()
            else:
                self.optimizer.step()
            self.optimizer.zero_grad()
        if self.lr_scheduler is not None:                                                # :1434-1435, every call
            self.lr_scheduler.step()
        self.global_step = gs + 1
        return loss.detach()


# ------------------------------------------------------------------------------------------ SD 2.1 (UNet) denoise loop
def dpm_solver_tables(num_inference_steps: int, num_train_timesteps: int = 1000, beta_start: float = 0.00085,
                      beta_end: float = 0.012):
    """(timesteps [n], sigmas [n+1]) of diffusers DPMSolverMultistepScheduler.set_timesteps with the SD 2.1 scheduler
    config (scaled_linear betas, 'linspace' spacing, final sigma 0)."""
    import numpy as np
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
    acp = torch.cumprod(1.0 - betas, 0).numpy().astype(np.float64)
    ts = np.linspace(0, num_train_timesteps - 1, num_inference_steps + 1).round()[::-1][:-1].copy().astype(np.int64)
    sig = np.interp(ts, np.arange(num_train_timesteps), ((1 - acp) / acp) ** 0.5)
    return torch.from_numpy(ts), torch.from_numpy(np.concatenate([sig, [0.0]]))


def dpm_solver_coefficients(sigmas: torch.Tensor, i: int, prediction_type: str):
    """(kx, ko, A, B, C) of step i for dwm_cfg_multistep: dpmsolver++ with solver_order 2, midpoint, first order on the
    first step, x' = x0 on the last (final sigma 0)."""
    import math
    n = sigmas.numel() - 1

    def a_s(s):
        a = 1.0 / math.sqrt(s * s + 1.0)
        return a, s * a
    a0, s0 = a_s(float(sigmas[i]))
    kx, ko = (1.0 / a0, -s0 / a0) if prediction_type == "epsilon" else (a0, -s0)
    if i == n - 1:
        return kx, ko, 0.0, 1.0, 0.0
    at, st = a_s(float(sigmas[i + 1]))
    lam = lambda a, s: math.log(a) - math.log(s)
    h = lam(at, st) - lam(a0, s0)
    e = math.exp(-h) - 1.0
    A = st / s0
    if i == 0:
        return kx, ko, A, -at * e, 0.0
    a1, s1 = a_s(float(sigmas[i - 1]))
    r0 = (lam(a0, s0) - lam(a1, s1)) / h
    return kx, ko, A, -at * e * (1.0 + 0.5 / r0), 0.5 * at * e / r0


class UNetDenoiser:
    """Hot loop of inference_pipeline (ctsd.py:1496-1575) for the SD 2.1 configs
    (examples/ctsd_21_6views_*_generation.json: DPMSolverMultistepScheduler, guidance 3, 50 steps): UNet forward at the
    CFG batch + guidance combine + DPM-Solver++(2M) update in one kernel.  latents fp32 [B,T,V,4,H,W]; conditions =
    CFG-doubled model kwargs (unconditional half first)."""

    def __init__(self, model, guidance_scale: float = 3.0, inference_steps: int = 50, prediction_type: str = "v_prediction",
                 scheduler=None):
        """scheduler None: DPM-Solver++(2M) (the example JSONs); a schedulers.DDIMScheduler: the reference's default test
        scheduler of the UNet (ctsd.py:969-974, `inference_config` without "scheduler"), its tensor-timestep step
        (temporal_independent.py:67-170) fused with the guidance combine."""
        self.model, self.guidance_scale, self.inference_steps = model, guidance_scale, inference_steps
        self.prediction_type = prediction_type
        self.scheduler = scheduler
        if scheduler is not None:
            if scheduler.config.prediction_type != prediction_type:
                raise ValueError(f"UNetDenoiser: prediction_type {prediction_type!r} disagrees with the scheduler's "
                                 f"{scheduler.config.prediction_type!r}")
            scheduler.set_timesteps(inference_steps)
            self.timesteps, self.sigmas = scheduler.timesteps.cpu(), None
        else:
            self.timesteps, self.sigmas = dpm_solver_tables(inference_steps)

    def prepare(self, latents: torch.Tensor, conditions: Dict[str, torch.Tensor]):
        dev = latents.device
        self.latents = latents.to(torch.float32).contiguous().clone()        # init_noise_sigma = 1
        self.x0_prev = torch.zeros_like(self.latents)
        B = latents.shape[0]
        # model input / conditions / prediction in the model's compute dtype: bf16, or fp32 for the accuracy path
        # (`model.compute_dtype = torch.float32`: BASELINE.json configs[0], the reference's fp32 denoise)
        self.cd = cd = getattr(self.model, "compute_dtype", bf16)
        self.model_in = torch.empty((2 * B, *latents.shape[1:]), dtype=cd, device=dev)
        lat16 = self.latents if cd == torch.float32 else ops.cast_bf16(self.latents)
        self.model_in[:B].copy_(lat16)
        self.model_in[B:].copy_(lat16)
        self.conditions = {k: (v.to(cd) if torch.is_tensor(v) and v.is_floating_point() and k != "added_time_ids" else v)
                           for k, v in conditions.items()}
        self._ts = self.timesteps.to(dev).float()
        # DDIM: the [steps, 6] coefficient rows of every step, on the device, once per prepare() (no host-to-device copy and
        # no fp64 table gather inside the loop: the step is a plain index, as in the DPM-Solver path)
        self._ddim_coef = None
        if self.scheduler is not None:
            self._ddim_coef = self.scheduler.coefficients(self.timesteps.to(dev))
        return self

    def step(self, i: int):
        B, T, V = self.latents.shape[:3]
        ts = self._ts[i].expand(2 * B, T, V)
        out = self.model(self.model_in, ts, **self.conditions)
        pred = out["noise_pred"] if isinstance(out, dict) else out[0][0]
        if self.scheduler is not None:
            from .schedulers import PREDICTION_TYPES
            sc = self.scheduler
            coef = self._ddim_coef[i].expand(B * T * V, 6).contiguous()
            f32_path = self.cd == torch.float32       # (the kernel writes a bf16 model input; the fp32 path copies the latents)
            ops.cfg_ddim_step(pred, self.latents, coef, self.latents[0, 0, 0].numel(), PREDICTION_TYPES[sc.config.prediction_type],
                              guidance=self.guidance_scale, clip_range=sc.config.clip_sample_range if sc.config.clip_sample else 0.0,
                              model_in=None if f32_path else self.model_in)
            if f32_path:
                self.model_in[:B].copy_(self.latents)
                self.model_in[B:].copy_(self.latents)
            return
        kx, ko, A, Bc, Cc = dpm_solver_coefficients(self.sigmas, i, self.prediction_type)
        ops.cfg_multistep(pred, self.latents, self.x0_prev, self.guidance_scale, kx, ko, A, Bc, Cc, model_in=self.model_in)

    def run(self, latents, conditions, stop: Optional[int] = None):
        self.prepare(latents, conditions)
        for i in range(self.inference_steps if stop is None else stop):
            self.step(i)
        return self.latents
