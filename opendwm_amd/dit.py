"""MI355X-native drop-in for the reference's CTSD MMDiT denoiser

    dwm.models.crossview_temporal_dit.DiTCrossviewTemporalConditionModel
    (src/dwm/models/crossview_temporal_dit.py:105-630)

Same constructor kwargs (reference kwargs :107-130 + the SD3Transformer2DModel kwargs the
JSON configs pass, e.g. examples/ctsd_35_6views_video_generation.json:45-107), same
forward signature and 3-tuple return (:372-391, :623-630), same state-dict keys
(SURVEY.md §8b), so `"_class_name": "opendwm_amd.dit.DiTCrossviewTemporalConditionModel"`
in a pipeline JSON is the whole integration (src/dwm/common.py:133-179).

Inference (`model.eval()` or no autograd): bf16 operands / fp32 accumulation through libdwm_hip.so, the hidden and
context streams kept in fp32 across the residual adds (`residual_dtype`), or everything in fp32
(`compute_dtype = torch.float32`, the accuracy path).  Training (`model.train()` with autograd enabled) routes the same
`forward` through opendwm_amd.train.forward_train (hand-written HIP backward).  There is no eager-PyTorch fallback: on a
machine without the built library or without a GPU the forward raises.
"""
from __future__ import annotations

import types
from typing import Optional

import torch
from torch import nn

from . import ops
from .blocks import (AlphaBlender, CombinedTimestepTextProjEmbeddings, JointTransformerBlock,
                     TimestepEmbedding, VTSelfAttentionBlock, _AdaNorm, _bf, stream32)
from .ops import EPI_RESID

bf16 = torch.bfloat16

try:   # when diffusers is importable, satisfy ctsd.py's isinstance(model, diffusers.SD3Transformer2DModel)
    import diffusers as _diffusers   # noqa: F401
    _Base = _diffusers.SD3Transformer2DModel
except Exception:   # diffusers absent (this container): plain nn.Module
    _Base = nn.Module


def _sincos_1d(embed_dim: int, pos: torch.Tensor) -> torch.Tensor:
    omega = torch.arange(embed_dim // 2, dtype=torch.float64) / (embed_dim / 2.0)
    omega = 1.0 / 10000 ** omega
    out = pos.reshape(-1).double()[:, None] * omega[None, :]
    return torch.cat([torch.sin(out), torch.cos(out)], dim=1)


def sincos_pos_embed_2d(embed_dim: int, grid_size: int, base_size: int) -> torch.Tensor:
    """diffusers get_2d_sincos_pos_embed as SD3's PatchEmbed builds its buffer."""
    g = torch.arange(grid_size, dtype=torch.float32) / (grid_size / base_size)
    gw, gh = torch.meshgrid(g, g, indexing="xy")
    return torch.cat([_sincos_1d(embed_dim // 2, gw), _sincos_1d(embed_dim // 2, gh)], dim=1).float()[None]


class PatchEmbed(nn.Module):
    """SD3 PatchEmbed: keys pos_embed.proj.{weight,bias}, persistent buffer pos_embed.pos_embed."""

    def __init__(self, sample_size: int, patch_size: int, in_channels: int, embed_dim: int,
                 pos_embed_max_size: int):
        super().__init__()
        self.patch_size, self.pos_embed_max_size = patch_size, pos_embed_max_size
        self.proj = nn.Conv2d(in_channels, embed_dim, kernel_size=patch_size, stride=patch_size, bias=True)
        self.register_buffer("pos_embed", sincos_pos_embed_2d(embed_dim, pos_embed_max_size,
                                                              sample_size // patch_size), persistent=True)
        self._cache = {}
        self._wcache = {}

    def cropped(self, h: int, w: int) -> torch.Tensor:
        from .blocks import STORE
        cd = STORE.precision
        key = (h, w, self.pos_embed.device, self.pos_embed.data_ptr(), cd)
        if key not in self._cache:
            m = self.pos_embed_max_size
            if h > m or w > m:
                raise ValueError(f"Height/width ({h},{w}) exceed pos_embed_max_size {m}")
            top, left = (m - h) // 2, (m - w) // 2
            t = self.pos_embed.reshape(1, m, m, -1)[:, top:top + h, left:left + w, :]
            self._cache = {kk: v for kk, v in self._cache.items() if kk[:4] == key[:4]}          # the other precision of this crop stays
            self._cache[key] = t.reshape(h * w, -1).to(cd).contiguous()
        return self._cache[key]

    def packed_weight(self):
        from .blocks import STORE
        key = (self.proj.weight.data_ptr(), self.proj.weight._version, STORE.step, STORE.precision)
        if key not in self._wcache:
            w = self.proj.weight.detach().reshape(self.proj.weight.shape[0], -1)
            k = w.shape[1]
            kp = (k + 63) // 64 * 64
            wp = torch.zeros((w.shape[0], kp), dtype=STORE.precision, device=w.device)
            wp[:, :k] = w
            self._wcache = {kk: v for kk, v in self._wcache.items() if kk[:3] == key[:3]}      # drop stale steps, keep the other precision
            self._wcache[key] = wp
        return self._wcache[key]

    def run(self, x: torch.Tensor) -> torch.Tensor:
        """x [I, C, H, W] -> bf16 [I*h*w, D] = conv(x) + cropped pos embed."""
        p = self.patch_size
        h, w = x.shape[-2] // p, x.shape[-1] // p
        wp = self.packed_weight()
        cols = ops.patchify(x, p, wp.shape[1], dtype=wp.dtype)
        return ops.gemm(cols, wp, _bf(self.proj.bias), epilogue=EPI_RESID, res=self.cropped(h, w), res_mod=h * w)


class RayEncoder(nn.Module):
    """Explicit perspective modelling (crossview_temporal_dit.py:39-64): positional encodings of the camera origin (8 octaves)
    and of the per-token view ray (4 octaves) -> Linear(72, D, bias=False).  State-dict key: `proj.weight`."""

    def __init__(self, pos_octaves=8, pos_start_octave=0, ray_octaves=4, ray_start_octave=0, cond_proj_dim=72, in_channels=1536):
        super().__init__()
        if (pos_octaves, pos_start_octave, ray_octaves, ray_start_octave, cond_proj_dim) != (8, 0, 4, 0, 72):
            raise NotImplementedError("RayEncoder: the reference's fixed octave layout (8 + 4 octaves from 0, 72 inputs)")
        self.proj = nn.Linear(cond_proj_dim, in_channels, bias=False)

    def packed(self) -> torch.Tensor:
        """proj.weight with K zero-padded 72 -> 128 (GEMM K granularity)"""
        from .blocks import STORE
        w = self.proj.weight

        def make():                                   # (per compute precision: STORE.derived keys on it)
            wp = torch.zeros((w.shape[0], 128), dtype=STORE.precision, device=w.device)
            wp[:, :w.shape[1]] = _bf(w)
            return wp
        return STORE.derived(w, "k128", make)

    @staticmethod
    def camera_rows(camera_intrinsics_norm: torch.Tensor, camera2referego: torch.Tensor, height: int, width: int) -> torch.Tensor:
        """[I, 21] fp32 rows of dwm_ray_features: inverse token-resolution intrinsics (:441-449, get_rays :66-102),
        camera -> reference-ego rotation, camera origin.  3x3 matrices per image: host-sized arithmetic."""
        K = camera_intrinsics_norm.flatten(0, -3).to(torch.float32).clone()
        K[:, 0, 0] *= width
        K[:, 1, 1] *= height
        K[:, 0, 2] *= width
        K[:, 1, 2] *= height
        M = camera2referego.flatten(0, -3).to(torch.float32)
        return torch.cat([torch.inverse(K).reshape(-1, 9), M[:, :3, :3].reshape(-1, 9), M[:, :3, 3]], 1).contiguous()

    def features(self, camera_intrinsics_norm, camera2referego, height: int, width: int) -> torch.Tensor:
        from .blocks import STORE
        return ops.ray_features(self.camera_rows(camera_intrinsics_norm, camera2referego, height, width), height, width, 128,
                                dtype=STORE.precision)


class DiTCrossviewTemporalConditionModel(_Base):
    def __init__(
        self,
        patch_size: int = 2,
        num_layers: int = 18,
        attention_head_dim: int = 64,
        num_attention_heads: int = 18,
        projection_class_embeddings_input_dim: int = None,
        condition_image_adapter_config: Optional[dict] = None,
        enable_crossview: bool = False,
        enable_temporal: bool = False,
        crossview_attention_type: str = None,
        temporal_attention_type: str = None,
        merge_factor: float = 2, merge_strategy="learned_with_images",
        crossview_block_layers: Optional[list] = None,
        temporal_block_layers: Optional[list] = None,
        crossview_gradient_checkpointing: bool = False,
        temporal_gradient_checkpointing: bool = False,
        mixer_type: str = "AlphaBlender",
        perspective_modeling_type: str = "",
        disable_view_emb_on_temporal_module: bool = False,
        qk_norm_on_additional_modules=None,
        mask_module=None,
        # --- diffusers.SD3Transformer2DModel kwargs (0.31.0 defaults)
        sample_size: int = 128,
        in_channels: int = 16,
        joint_attention_dim: int = 4096,
        caption_projection_dim: int = 1152,
        pooled_projection_dim: int = 2048,
        out_channels: int = 16,
        pos_embed_max_size: int = 96,
        dual_attention_layers=(),
        qk_norm: Optional[str] = None,
    ):
        nn.Module.__init__(self)
        if mask_module is not None:
            raise NotImplementedError("mask_module (MaskGWM) is training-only and out of scope (SURVEY.md §2)")
        if mixer_type != "AlphaBlender":
            raise NotImplementedError("only mixer_type='AlphaBlender' (every shipped config) is supported")
        if perspective_modeling_type not in ("", "implicit", "explicit"):
            raise NotImplementedError(f"perspective_modeling_type={perspective_modeling_type!r}")
        if attention_head_dim != 64:
            raise NotImplementedError("the attention kernel is built for head_dim 64 (SD 3 / 3.5)")

        inner_dim = attention_head_dim * num_attention_heads
        self.inner_dim = inner_dim
        self.out_channels = out_channels if out_channels is not None else in_channels
        self._cfg = types.SimpleNamespace(
            patch_size=patch_size, num_layers=num_layers, attention_head_dim=attention_head_dim,
            num_attention_heads=num_attention_heads, sample_size=sample_size, in_channels=in_channels,
            joint_attention_dim=joint_attention_dim, caption_projection_dim=caption_projection_dim,
            pooled_projection_dim=pooled_projection_dim, out_channels=self.out_channels,
            pos_embed_max_size=pos_embed_max_size, dual_attention_layers=tuple(dual_attention_layers),
            qk_norm=qk_norm)
        self.gradient_checkpointing = False
        self.crossview_gradient_checkpointing = crossview_gradient_checkpointing
        self.temporal_gradient_checkpointing = temporal_gradient_checkpointing
        self.disable_view_emb_on_temporal_module = disable_view_emb_on_temporal_module
        self.num_attention_heads = num_attention_heads

        # ---- SD3Transformer2DModel members
        self.pos_embed = PatchEmbed(sample_size, patch_size, in_channels, inner_dim, pos_embed_max_size)
        self.time_text_embed = CombinedTimestepTextProjEmbeddings(inner_dim, pooled_projection_dim)
        self.context_embedder = nn.Linear(joint_attention_dim, caption_projection_dim)
        self.transformer_blocks = nn.ModuleList([
            JointTransformerBlock(inner_dim, num_attention_heads, attention_head_dim,
                                  context_pre_only=i == num_layers - 1, qk_norm=qk_norm,
                                  use_dual_attention=i in dual_attention_layers)
            for i in range(num_layers)])
        self.norm_out = _AdaNorm(inner_dim, 2)
        self.proj_out = nn.Linear(inner_dim, patch_size * patch_size * self.out_channels)

        # ---- reference members (crossview_temporal_dit.py:142-221)
        if condition_image_adapter_config is not None:
            from .adapters import ImageAdapter
            self.condition_image_adapter = ImageAdapter(**condition_image_adapter_config)
        else:
            self.condition_image_adapter = None
        self._adapter_cache = (None, None)
        # True: the layout residuals (step-invariant: they depend on the condition images only) are computed once, kept in
        # fp32 and re-used while the condition tensor is the same object; False: recomputed by every forward, as the
        # reference's forward does (crossview_temporal_dit.py:459-462) - then each zero convolution adds straight into the
        # hidden state from its GEMM epilogue (adapters.ImageAdapter.residual_adders)
        self.cache_adapter_residuals = True
        self.adapter_cache_dtype = torch.float32      # arithmetic of the CACHED residuals: fp32 path (default) or torch.bfloat16
        self.frame_shard = None             # set by CTSDDenoiser(frame_group=...): opendwm_amd.sharding.FrameShard
        self.compute_dtype = bf16           # torch.float32 selects the fp32 accuracy path of the inference forward
        # dtype of the hidden / context streams of the bf16 inference forward.  fp32 (default): the ~130 residual adds of a
        # forward accumulate in fp32 (GEMM RESID epilogues with dwm_gemm_args.C32, LayerNorms reading fp32) - each add into a
        # bf16 stream is a rounding of the whole stream (1.1e-3 rms each; they add up to the 1.3e-2 a bf16-stream forward
        # shows against the fp32 oracle).  bf16: the round-1..3 behaviour (half the stream traffic).
        self.gemm_4wave = True               # 4-wave GEMM kernels for the launches they cover (ops.gemm_4wave_scope): inference and training
        # True (default): the AdaLN modulation rows of ALL joint blocks and of norm_out (they depend on the timestep embedding only) come
        # from ONE stacked GEMM per forward instead of two M = I launches per block (35-223 TFLOP/s each): validated and measured in
        # round 5 (393.8 -> 392.6 ms per step, profiles/r5a_*).  Costs a packed copy of those weights (1.7 GB at full size).
        self.stack_modulation = True
        self.residual_dtype = torch.float32
        self._index_sinusoids = {}
        self.perspective_modeling_type = perspective_modeling_type
        if perspective_modeling_type == "implicit":
            self.view_embedding = TimestepEmbedding(projection_class_embeddings_input_dim, inner_dim)
        elif perspective_modeling_type == "explicit":
            self.rayencoder = RayEncoder(cond_proj_dim=72, in_channels=inner_dim)                  # :156-159

        self.enable_crossview = enable_crossview
        self.crossview_attention_type = crossview_attention_type
        self.crossview_block_layers = crossview_block_layers
        if enable_crossview:
            n = len(crossview_block_layers)
            self.view_pos_embeds = nn.ModuleList([
                TimestepEmbedding(inner_dim, inner_dim * 4, out_dim=inner_dim) for _ in range(n)])
            self.crossview_transformer_blocks = nn.ModuleList([
                VTSelfAttentionBlock(inner_dim, inner_dim, num_attention_heads, attention_head_dim,
                                     qk_norm=qk_norm_on_additional_modules) for _ in range(n)])
            self.view_mixers = nn.ModuleList([
                AlphaBlender(merge_factor, merge_strategy=merge_strategy) for _ in range(n)])

        self.enable_temporal = enable_temporal
        self.temporal_attention_type = temporal_attention_type
        self.temporal_block_layers = temporal_block_layers
        if enable_temporal:
            n = len(temporal_block_layers)
            self.time_pos_embeds = nn.ModuleList([
                TimestepEmbedding(inner_dim, inner_dim * 4, out_dim=inner_dim) for _ in range(n)])
            self.temporal_transformer_blocks = nn.ModuleList([
                VTSelfAttentionBlock(inner_dim, inner_dim, num_attention_heads, attention_head_dim,
                                     qk_norm=qk_norm_on_additional_modules) for _ in range(n)])
            self.time_mixers = nn.ModuleList([
                AlphaBlender(merge_factor, merge_strategy=merge_strategy) for _ in range(n)])

        self.depth_net = None
        self.mask_module = None

    # ---- nn.Module / ModelMixin protocol used by ctsd.py
    @property
    def config(self):
        return self._cfg

    def enable_gradient_checkpointing(self):
        self.gradient_checkpointing = True

    def _invalidate_packed(self):
        for m in self.modules():
            if hasattr(m, "_pk"):
                m._pk = None
            if hasattr(m, "_cache"):
                m._cache = {}
            if hasattr(m, "_wcache"):
                m._wcache = {}
            if hasattr(m, "_alpha_cache"):
                m._alpha_cache = None            # AlphaBlender.get_alpha (also keyed on STORE.step, bumped below)
        self._adapter_cache = (None, None)
        from .blocks import STORE
        STORE.bump()

    def _apply(self, fn, *a, **kw):
        out = super()._apply(fn, *a, **kw)
        self._invalidate_packed()
        return out

    def load_state_dict(self, state_dict, *a, **kw):
        out = super().load_state_dict(state_dict, *a, **kw)
        self._invalidate_packed()
        return out

    # ---- forward (crossview_temporal_dit.py:372-630)
    def forward(self, sample, timestep=None, *args, **kwargs):
        """Reference signature.  In train() mode with autograd enabled the prediction carries a grad_fn
        (opendwm_amd.train: checkpointed block Functions with HIP backward kernels); otherwise the fused
        inference path runs under no_grad."""
        if self.training and torch.is_grad_enabled():
            from . import train as _train
            names = ["frustum_bev_residuals", "encoder_hidden_states", "pooled_projections", "condition_image_tensor",
                     "disable_crossview", "disable_temporal", "crossview_attention_mask", "crossview_attention_index",
                     "camera_intrinsics", "camera_transforms", "camera_intrinsics_norm", "camera2referego",
                     "added_time_ids", "noise", "return_dict"]
            kw = dict(zip(names, args))
            kw.update(kwargs)
            squeeze = sample.dim() < 6
            if squeeze:
                sample, timestep = sample.unsqueeze(2), timestep.unsqueeze(2)
                for k in ("encoder_hidden_states", "pooled_projections", "disable_temporal"):
                    if kw.get(k) is not None:
                        kw[k] = kw[k].unsqueeze(2)
            with ops.gemm_4wave_scope(self.gemm_4wave):     # (the block Functions carry the scope into their backward)
                out = _train.forward_train(self, sample, timestep, kw.get("encoder_hidden_states"), kw.get("pooled_projections"),
                                           disable_crossview=kw.get("disable_crossview"), disable_temporal=kw.get("disable_temporal"),
                                           crossview_attention_mask=kw.get("crossview_attention_mask"),
                                           added_time_ids=kw.get("added_time_ids"),
                                           condition_image_tensor=kw.get("condition_image_tensor"),
                                           camera_intrinsics_norm=kw.get("camera_intrinsics_norm"),
                                           camera2referego=kw.get("camera2referego"))
            if kw.get("return_dict"):                       # crossview_temporal_dit.py:620-630: only the dict form is squeezed
                return {"noise_pred": out.squeeze(2) if squeeze else out}
            return [out], None, None
        from .blocks import STORE
        try:
            with ops.gemm_4wave_scope(self.gemm_4wave):      # this thread's launches may run on the 4-wave GEMM kernels
                return self._forward_infer(sample, timestep, *args, **kwargs)
        finally:
            STORE.set_precision(bf16)       # the fp32 accuracy path is scoped to this forward (compute_dtype = torch.float32)

    def _stacked_modulation(self, silu_temb: torch.Tensor):
        """[norm1, norm1_context] modulation rows of every joint block + norm_out's, as column slices of ONE GEMM over the
        stacked `linear` weights (the packed copy lives with the other packed weights: rebuilt after an optimizer step /
        state-dict load)"""
        from .blocks import STORE
        lins = [lin for blk in self.transformer_blocks for lin in (blk.norm1.linear, blk.norm1_context.linear)] + [self.norm_out.linear]

        def make():
            return {"w": torch.cat([_bf(l.weight) for l in lins]).contiguous(), "b": torch.cat([_bf(l.bias) for l in lins]).contiguous()}
        pk = STORE.cached(self, make)
        allmod = ops.gemm(silu_temb, pk["w"], pk["b"])
        out, off = [], 0
        for l in lins:
            n = l.weight.shape[0]
            out.append(allmod[:, off:off + n])
            off += n
        return out

    def _index_sinusoid(self, kind: str, B: int, T: int, V: int, D: int, device, dtype) -> torch.Tensor:
        """sinusoid features [B*T*V, D] of the frame ("t") or view ("v") index of every image (crossview_temporal_dit.py:
        528-531, 553-556): input-independent, kept on the model"""
        key = (kind, B, T, V, D, str(device), dtype)
        hit = self._index_sinusoids.get(key)
        if hit is None:
            if kind == "t":
                idx = torch.arange(T, device=device).view(1, T, 1).expand(B, T, V)
            else:
                idx = torch.arange(V, device=device).view(1, 1, V).expand(B, T, V)
            hit = ops.timestep_sinusoid(idx, D, dtype=dtype)
            # Kept for the life of the model, never evicted: a HIP graph captured by CTSDDenoiser.enable_graph holds the POINTERS of
            # these tensors, and freeing one a captured graph still replays would let it read recycled memory.  They are a few KiB per
            # (batch, frames, views) shape; a model that has seen more than 64 shapes says so once instead of growing silently.
            if len(self._index_sinusoids) == 64:
                import warnings
                warnings.warn("DiTCrossviewTemporalConditionModel: more than 64 distinct (batch, frames, views) shapes seen; the "
                              "index-embedding cache keeps growing (entries are never freed: captured graphs may reference them)")
            self._index_sinusoids[key] = hit
        return hit

    @staticmethod
    def _mixer_alphas(mixers, image_only_indicator, batch: int):
        """alpha[batch] of every AlphaBlender of a list (crossview_temporal.py:33-51) from ONE stacked evaluation when they
        share a learned strategy; rows of the result = mixers"""
        strategies = {m.merge_strategy for m in mixers}
        if len(strategies) != 1 or "fixed" in strategies:
            return [m.get_alpha(image_only_indicator, batch) for m in mixers]
        sig = torch.sigmoid(torch.cat([m.mix_factor.detach().float().reshape(1) for m in mixers]))[:, None]      # [mixers, 1]
        if strategies == {"learned"}:
            return list(sig.expand(len(mixers), batch).contiguous())
        if image_only_indicator is None:
            raise ValueError("Please provide image_only_indicator to use learned_with_images merge strategy")
        flag = image_only_indicator.reshape(1, batch).to(device=sig.device, dtype=torch.bool)
        return list(torch.where(flag, torch.ones((), device=sig.device), sig).contiguous())

    @torch.no_grad()
    def _forward_infer(
        self,
        sample: torch.FloatTensor,
        timestep: torch.LongTensor = None,
        frustum_bev_residuals: torch.Tensor = None,
        encoder_hidden_states: torch.FloatTensor = None,
        pooled_projections: torch.FloatTensor = None,
        condition_image_tensor: torch.Tensor = None,
        disable_crossview: torch.BoolTensor = None,
        disable_temporal: torch.BoolTensor = None,
        crossview_attention_mask: torch.Tensor = None,
        crossview_attention_index: torch.Tensor = None,
        camera_intrinsics: torch.Tensor = None,
        camera_transforms: torch.Tensor = None,
        camera_intrinsics_norm: torch.Tensor = None,
        camera2referego: torch.Tensor = None,
        added_time_ids: torch.Tensor = None,
        noise: torch.Tensor = None,
        return_dict: bool = False,
    ):
        if not sample.is_cuda:
            raise RuntimeError("opendwm_amd runs on an MI355X (HIP) device only; got a CPU tensor")
        should_add_dim = sample.dim() < 6
        if should_add_dim:
            sample = sample.unsqueeze(2)
            timestep = timestep.unsqueeze(2)
            if encoder_hidden_states is not None:
                encoder_hidden_states = encoder_hidden_states.unsqueeze(2)
            if disable_temporal is not None:
                disable_temporal = disable_temporal.unsqueeze(2)
            if pooled_projections is not None:
                pooled_projections = pooled_projections.unsqueeze(2)

        B, T, V, _, H, W = sample.shape
        p = self._cfg.patch_size
        height, width = H // p, W // p
        N, I, D = height * width, B * T * V, self.inner_dim
        self.view_count, self.width = V, width
        # frame_shard (opendwm_amd.sharding.FrameShard): `sample` and the per-frame conditions hold only this rank's
        # frames of the sample; T is the local frame count, Tg the sample's
        fs, cam_all = self.frame_shard, None
        Tg = T if fs is None else T * fs.size
        if fs is not None and self.enable_temporal:
            fs.check(height, self.temporal_attention_type)

        # compute dtype: bf16 (storage bf16, fp32 accumulation / statistics), or - `model.compute_dtype = torch.float32` -
        # the fp32 accuracy path (north_star's 1e-3 tolerance; the reference runs this graph in fp32 when no autocast /
        # fp16 cast is configured, ctsd.py:1189-1193)
        from .blocks import STORE
        cd = self.compute_dtype
        STORE.set_precision(cd)           # reset to bf16 by `forward` on the way out

        def as_bf16(t):
            if cd == torch.float32:
                return t if t.dtype == torch.float32 else t.float()
            return t if t.dtype == bf16 else (ops.cast_bf16(t.contiguous()) if t.dtype == torch.float32 else t.to(bf16))

        x = sample.flatten(0, 2).contiguous()
        if cd == torch.float32:
            x = x.float()
        elif x.dtype not in (torch.float32, bf16):
            x = x.to(bf16)
        h = self.pos_embed.run(x)                                                      # [I*N, D]
        ehs = as_bf16(encoder_hidden_states.flatten(0, 2))
        Lc = ehs.shape[1]
        c = ops.gemm(ehs.reshape(I * Lc, -1), _bf(self.context_embedder.weight), _bf(self.context_embedder.bias))
        pooled = as_bf16(pooled_projections.flatten(0, 2)).contiguous()
        temb = self.time_text_embed.run(timestep.flatten(), pooled)                    # [I, D]
        silu_temb = ops.silu(temb)
        if cd == bf16 and self.residual_dtype == torch.float32:                        # fp32 residual streams (blocks.stream32)
            h, c = ops.cast_f32(h), ops.cast_f32(c)
        elif self.residual_dtype not in (bf16, torch.float32):
            raise ValueError("residual_dtype must be torch.float32 or torch.bfloat16")

        view_cam_emb = None
        ray_feat = None
        if self.perspective_modeling_type == "implicit":
            ve = ops.timestep_sinusoid(added_time_ids.flatten(), 256, dtype=cd).view(I, -1)
            view_cam_emb = self.view_embedding.run(ve)                                 # [I, D]
        elif self.perspective_modeling_type == "explicit":
            # per-TOKEN embedding raymap[I*N, D] (:440-458).  Kept as its 72 (padded 128) input features: every VT block
            # builds `index embedding + raymap` in ONE small GEMM (K = 128) whose residual is the per-image index
            # embedding, instead of holding a 264 MB raymap and adding it in a separate pass
            if camera_intrinsics_norm is None or camera2referego is None:
                raise RuntimeError("perspective_modeling_type='explicit' needs camera_intrinsics_norm and camera2referego")
            ray_feat = self.rayencoder.features(camera_intrinsics_norm, camera2referego, height, width)
            if fs is not None and self.enable_temporal and self.enable_crossview and not self.disable_view_emb_on_temporal_module:
                # the temporal blocks run on "all frames, my token rows" (sharding.py): the features of those rows for every
                # frame of the sample, from the gathered per-image camera matrices (a few hundred bytes per image) - the
                # features themselves never cross the links
                hl = height // fs.size
                rf_all = self.rayencoder.features(fs.gather_frames(camera_intrinsics_norm, 1), fs.gather_frames(camera2referego, 1),
                                                  height, width)
                ray_rows = rf_all.view(B, Tg, V, height, width, rf_all.shape[-1])[:, :, :, fs.rank * hl:(fs.rank + 1) * hl] \
                    .contiguous().view(-1, rf_all.shape[-1])

        if self.enable_crossview and disable_crossview is None:
            disable_crossview = torch.zeros(B, dtype=torch.bool, device=sample.device)
        if self.enable_temporal and disable_temporal is None:
            disable_temporal = torch.zeros(B, dtype=torch.bool, device=sample.device)

        # layout residuals (crossview_temporal_dit.py:459-462).  They depend only on the condition
        # images, which do not change across denoise steps: cached on the tensor's identity.
        condition_residuals = residual_adders = None
        if self.condition_image_adapter is not None and condition_image_tensor is not None and \
                (not self.cache_adapter_residuals or cd == torch.float32):      # (the fp32 accuracy path always recomputes)
            residual_adders = self.condition_image_adapter.residual_adders(condition_image_tensor)
        elif self.condition_image_adapter is not None and condition_image_tensor is not None:
            # the key holds the optimizer step (residuals of old adapter weights must not survive a training step) and
            # the cache keeps the tensor alive (a freed tensor's address can be handed to the next same-shape batch)
            from .blocks import STORE
            key = (condition_image_tensor.data_ptr(), condition_image_tensor._version, tuple(condition_image_tensor.shape), STORE.step)
            if self._adapter_cache[0] != key:
                # Computed once per condition tensor, so it can afford the fp32 accuracy path (dwm_gemm_f32, 3 x the MFMA work of
                # the bf16 adapter, ~1 % of a 40-step loop): the adapter's error is the SAME at every denoise step and differs
                # between the conditional and the unconditional half (different layout images), so classifier-free guidance
                # multiplies it (4 c - 3 u) and the steps add it up coherently - it is what holds the text+layout model at
                # 1.3e-2 after 40 steps when the text-only model is at 6e-3 (profiles/r4b_gpu_parity.log).
                if self.adapter_cache_dtype == torch.float32:
                    STORE.set_precision(torch.float32)
                    try:
                        feats = self.condition_image_adapter.run(condition_image_tensor)
                    finally:
                        STORE.set_precision(cd)
                else:
                    feats = self.condition_image_adapter.run(condition_image_tensor, precise=True)
                self._adapter_cache = (key, feats, condition_image_tensor)
            condition_residuals = list(self._adapter_cache[1])
            for f in condition_residuals:
                if f.shape != h.shape:
                    raise RuntimeError(f"condition residual {tuple(f.shape)} does not match hidden states {tuple(h.shape)}")

        # per-block constants that do not depend on the hidden state, prepared once: the frame / view index sinusoids
        # (input-independent: cached on the model) and the mixers' alpha vectors (one stacked sigmoid instead of one per block)
        seq_sin = view_sin = None
        if self.enable_temporal and self.temporal_block_layers:
            seq_sin = self._index_sinusoid("t", B, Tg, V, D, h.device, cd)
            t_alpha = self._mixer_alphas(self.time_mixers, disable_temporal, B)
        if self.enable_crossview and self.crossview_block_layers:
            view_sin = self._index_sinusoid("v", B, T, V, D, h.device, cd)
            v_alpha = self._mixer_alphas(self.view_mixers, disable_crossview, B)

        mods = self._stacked_modulation(silu_temb) if (self.stack_modulation and cd == bf16) else None
        for i, block in enumerate(self.transformer_blocks):
            if condition_residuals:
                ops.add_(h, condition_residuals.pop(0))                                 # :491-494 (fp32 residual, one rounding)
            elif residual_adders is not None:
                add = next(residual_adders, None)
                if add is None:
                    residual_adders = None
                else:
                    add(h)                                                              # :491-494, fused into the zero-conv GEMM
            if mods is None:
                c, h = block.run(h, c, silu_temb, I)
            else:
                c, h = block.run(h, c, silu_temb, I, mod=mods[2 * i], cmod=mods[2 * i + 1])

            if self.enable_temporal and i in self.temporal_block_layers:
                k = self.temporal_block_layers.index(i)
                seq = seq_sin
                use_cam = self.enable_crossview and not self.disable_view_emb_on_temporal_module \
                    and view_cam_emb is not None
                if use_cam and fs is not None and cam_all is None:
                    cam_all = fs.gather_frames(view_cam_emb.view(B, T, V, D), 1).view(-1, D)
                seq_emb = self.time_pos_embeds[k].run(seq, res=(view_cam_emb if fs is None else cam_all) if use_cam else None)
                rpe = N
                if ray_feat is not None and self.enable_crossview and not self.disable_view_emb_on_temporal_module:
                    seq_emb = ops.gemm(ray_feat if fs is None else ray_rows, self.rayencoder.packed(), None, epilogue=ops.EPI_RESID,
                                       res=seq_emb, res_mod=-(N if fs is None else (height // fs.size) * width))
                    rpe = 1
                tt = self.temporal_attention_type
                mk = ops.rowmap_temporal_full if tt == "full" else \
                    ops.rowmap_temporal_rowwise if tt == "rowwise" else ops.rowmap_temporal_pointwise
                alpha = t_alpha[k]
                if fs is None:
                    self.temporal_transformer_blocks[k].run(
                        h, mk(B, T, V, height, width), emb=seq_emb, rows_per_emb=rpe,
                        blend_alpha=alpha, rows_per_alpha=T * V * N, blend_into=h)
                else:
                    # frames of this sample live on other ranks: all frames of MY token rows, block + mixer, and back
                    hl = height // fs.size
                    hx = fs.frames_to_rows(h, B, T, V, height, width)
                    self.temporal_transformer_blocks[k].run(
                        hx, mk(B, Tg, V, hl, width), emb=seq_emb, rows_per_emb=1 if rpe == 1 else hl * width,
                        blend_alpha=alpha, rows_per_alpha=Tg * V * hl * width, blend_into=hx)
                    fs.rows_to_frames(hx, B, T, V, height, width, out=h)

            if self.enable_crossview and i in self.crossview_block_layers:
                k = self.crossview_block_layers.index(i)
                ve = view_sin
                view_emb = self.view_pos_embeds[k].run(ve, res=view_cam_emb)
                rpe = N
                if ray_feat is not None:
                    view_emb = ops.gemm(ray_feat, self.rayencoder.packed(), None, epilogue=ops.EPI_RESID, res=view_emb, res_mod=-N)
                    rpe = 1
                ct = self.crossview_attention_type
                gmask = dmask = None
                if ct == "rowwise":
                    rm = ops.rowmap_crossview_rowwise(B, T, V, height, width)
                    gmask = crossview_attention_mask
                elif ct == "full":
                    rm = ops.rowmap_crossview_full(B, T, V, height, width)
                    dmask = crossview_attention_mask       # reference passes it through un-expanded
                else:
                    raise NotImplementedError(f"Not support {ct}")
                alpha = v_alpha[k]
                self.crossview_transformer_blocks[k].run(
                    h, rm, emb=view_emb, rows_per_emb=rpe, group_mask=gmask, dense_mask=dmask,
                    blend_alpha=alpha, rows_per_alpha=T * V * N, blend_into=h)

        # norm_out (AdaLayerNormContinuous: scale first) + proj_out + unpatchify
        mod = mods[-1] if mods is not None else ops.gemm(silu_temb, _bf(self.norm_out.linear.weight), _bf(self.norm_out.linear.bias))
        nh = ops.layernorm(h, eps=1e-6, scale=mod[:, :D], shift=mod[:, D:], rows_per_mod=N, x32=stream32(h))
        y = ops.gemm(nh, _bf(self.proj_out.weight), _bf(self.proj_out.bias))
        out = ops.unpatchify(y, I, self.out_channels, height, width, p)
        output = out.view(B, T, V, self.out_channels, height * p, width * p)

        result = [output]
        if should_add_dim:
            output = output.squeeze(2)
        if return_dict:
            return {"noise_pred": output}
        return result, None, None


def model_flops(cfg: dict, B: int, T: int, V: int, H: int, W: int, text_len: int = 154) -> dict:
    """Algorithmic FLOPs of one forward: 2·MAC per Linear + 4·L²·d per
    attention problem.  `cfg` = constructor kwargs."""
    d = cfg["num_attention_heads"] * cfg["attention_head_dim"]
    p = cfg.get("patch_size", 2)
    h, w = H // p, W // p
    N, I = h * w, B * T * V
    tok, ctx = I * N, I * text_len
    nl = cfg["num_layers"]
    nd = len(cfg.get("dual_attention_layers", ()))
    ncv = len(cfg.get("crossview_block_layers") or []) if cfg.get("enable_crossview") else 0
    ntm = len(cfg.get("temporal_block_layers") or []) if cfg.get("enable_temporal") else 0
    f = {}
    f["joint_linear"] = tok * 2 * 12 * d * d * nl + ctx * 2 * 12 * d * d * (nl - 1) + ctx * 2 * 3 * d * d \
        + tok * 2 * 4 * d * d * nd + I * 2 * d * (9 * d * nd + 6 * d * (nl - nd) + 6 * d * (nl - 1) + 2 * d)
    f["joint_attn"] = I * 4 * (N + text_len) ** 2 * d * nl + I * 4 * N * N * d * nd
    f["vt_linear"] = tok * 2 * 28 * d * d * (ncv + ntm)
    f["cv_attn"] = ((B * T * h) * 4 * (V * w) ** 2 * d if cfg.get("crossview_attention_type") == "rowwise"
                    else (B * T) * 4 * (V * N) ** 2 * d) * ncv
    tt = cfg.get("temporal_attention_type")
    f["t_attn"] = ((B * V * h) * 4 * (T * w) ** 2 * d if tt == "rowwise" else
                   (B * V) * 4 * (T * N) ** 2 * d if tt == "full" else (B * V * N) * 4 * T * T * d) * ntm
    cj, pd = cfg.get("joint_attention_dim", 4096), cfg.get("pooled_projection_dim", 2048)
    ac = cfg.get("condition_image_adapter_config")
    f["adapter"] = 0
    if ac is not None:      # convs on the token grid: 3x3 (K = 9C) + 1x1 per resnet, in_conv, zero convs
        chs = ac["channels"]
        cin = ac.get("in_channels", 3) * ac.get("downscale_factor", 8) ** 2
        for i, ch in enumerate(chs):
            prev = cin if i == 0 else chs[i - 1]
            f["adapter"] += tok * 2 * ((prev * ch if prev != ch else 0) + ac.get("num_res_blocks", 2) * 10 * ch * ch
                                       + (ch * ch if ac.get("use_zero_convs") else 0))
    f["embeds"] = tok * 2 * 64 * d + ctx * 2 * cj * d + I * 2 * (256 * d + d * d + pd * d + d * d) \
        + I * 2 * (11 * 256 * d + d * d) + (ncv + ntm) * I * 2 * 8 * d * d + I * 4 * d * d + tok * 2 * 64 * d
    f["attention"] = f["joint_attn"] + f["cv_attn"] + f["t_attn"]
    f["total"] = f["joint_linear"] + f["vt_linear"] + f["attention"] + f["embeds"]      # adapter reported separately
    return f
