"""CPU oracle for the CTSD SD-3.5 MMDiT denoising hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``opendwm_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and there only as the checker / reported baseline.

PARITY UNPINNED for the diffusers arithmetic, pinned for the reference's own
logic: the reference (SenseTime-FVG/OpenDWM @ 2025-07-04) ships no tests, golden
tensors or fixtures for this path and its model classes cannot
be built in this container (``diffusers==0.31.0`` is absent).  What can be
executed is: AlphaBlender, the cross-view / temporal rearrange + mask + mix
methods, the inference / autoregressive / streaming control flow and
``step_by_indices`` (imported behind import-only stubs;
tests/golden/make_reference_fixtures.py, make_reference_driver_fixtures.py) -
``alpha_blender``, ``crossview_block_and_mix``, ``temporal_block_and_mix`` and
``denoise`` below are checked against those vectors
(tests/test_reference_fixtures_cpu.py); so are ``dit_forward`` - against the real
``DiTCrossviewTemporalConditionModel.forward`` / ``VTSelfAttentionBlock.forward`` /
``ImageAdapter.forward`` composed over this file's leaf functions
(make_reference_forward_fixture.py) - and ``train_loss`` - against the real
``train_step`` (make_reference_train_fixture.py).  This file is
a plain-PyTorch fp32 *restatement* of

* the reference-owned arithmetic
    - ``src/dwm/models/crossview_temporal_dit.py:372-630``  (forward)
    - ``src/dwm/models/crossview_temporal_dit.py:223-370``  (cross-view /
      temporal block + mix)
    - ``src/dwm/models/crossview_temporal.py:9-72``   (AlphaBlender)
    - ``src/dwm/models/crossview_temporal.py:536-582`` (VTSelfAttentionBlock)
    - ``src/dwm/pipelines/ctsd.py:1496-1575`` (denoise loop: CFG + scheduler)
* the third-party arithmetic those files call (``diffusers==0.31.0``,
  ``requirements.txt:6``): SD3Transformer2DModel pieces — PatchEmbed,
  CombinedTimestepTextProjEmbeddings, Timesteps/TimestepEmbedding,
  AdaLayerNormZero / SD35AdaLayerNormZeroX / AdaLayerNormContinuous, RMSNorm,
  Attention + JointAttnProcessor2_0 / AttnProcessor2_0, FeedForward (GEGLU,
  gelu-approximate), FlowMatchEulerDiscreteScheduler — restated from the
  published behaviour of that release.

Everything is a pure function of (config, state_dict, inputs).  The
state-dict key names equal the reference module tree
(``crossview_temporal_dit.py:131-221`` + diffusers naming) so one set of
weights loads into this oracle, the HIP model and — when a machine with
diffusers is available — the reference class itself.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# --------------------------------------------------------------------------
# diffusers==0.31.0 embeddings
# --------------------------------------------------------------------------

def timesteps_sinusoid(t: Tensor, num_channels: int, flip_sin_to_cos: bool = True,
                       downscale_freq_shift: float = 0.0,
                       max_period: float = 10000.0) -> Tensor:
    """diffusers ``Timesteps`` / ``get_timestep_embedding``; fp32 output.

    Used at crossview_temporal_dit.py:153-154 (index_proj), :163-164
    (view_cam_proj) and inside time_text_embed."""
    half = num_channels // 2
    exponent = -math.log(max_period) * torch.arange(
        half, dtype=torch.float32, device=t.device)
    exponent = exponent / (half - downscale_freq_shift)
    emb = t.reshape(-1)[:, None].float() * torch.exp(exponent)[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    if num_channels % 2 == 1:
        emb = F.pad(emb, (0, 1))
    return emb


def linear(sd: SD, prefix: str, x: Tensor) -> Tensor:
    return F.linear(x, sd[prefix + ".weight"], sd.get(prefix + ".bias"))


def timestep_embedding_mlp(sd: SD, prefix: str, x: Tensor) -> Tensor:
    """diffusers ``TimestepEmbedding``: linear_2(silu(linear_1(x)))."""
    return linear(sd, prefix + ".linear_2", F.silu(linear(sd, prefix + ".linear_1", x)))


def get_1d_sincos(embed_dim: int, pos: Tensor) -> Tensor:
    omega = torch.arange(embed_dim // 2, dtype=torch.float64) / (embed_dim / 2.0)
    omega = 1.0 / 10000 ** omega
    out = pos.reshape(-1).double()[:, None] * omega[None, :]
    return torch.cat([torch.sin(out), torch.cos(out)], dim=1)


def make_pos_embed_table(embed_dim: int, pos_embed_max_size: int, base_size: int,
                         interpolation_scale: float = 1.0) -> Tensor:
    """diffusers ``get_2d_sincos_pos_embed`` as SD3 ``PatchEmbed`` calls it
    (grid_size = pos_embed_max_size, base_size = sample_size // patch_size).
    Returns the persistent buffer ``pos_embed.pos_embed`` [1, max*max, D] fp32.
    With real checkpoints the buffer comes from the state dict."""
    g = pos_embed_max_size
    grid_h = torch.arange(g, dtype=torch.float32) / (g / base_size) / interpolation_scale
    grid_w = torch.arange(g, dtype=torch.float32) / (g / base_size) / interpolation_scale
    # np.meshgrid(grid_w, grid_h): w varies fastest; grid[0] = w coords, grid[1] = h coords
    gw, gh = torch.meshgrid(grid_w, grid_h, indexing="xy")
    emb_h = get_1d_sincos(embed_dim // 2, gw)   # diffusers feeds grid[0] to "emb_h"
    emb_w = get_1d_sincos(embed_dim // 2, gh)
    return torch.cat([emb_h, emb_w], dim=1).float()[None]


def cropped_pos_embed(table: Tensor, h: int, w: int, pos_embed_max_size: int) -> Tensor:
    top = (pos_embed_max_size - h) // 2
    left = (pos_embed_max_size - w) // 2
    t = table.reshape(1, pos_embed_max_size, pos_embed_max_size, -1)
    return t[:, top:top + h, left:left + w, :].reshape(1, h * w, -1)


def patch_embed(sd: SD, cfg: dict, x: Tensor) -> Tensor:
    """SD3 ``PatchEmbed.forward`` (crossview_temporal_dit.py:421)."""
    p = cfg["patch_size"]
    h, w = x.shape[-2] // p, x.shape[-1] // p
    y = F.conv2d(x, sd["pos_embed.proj.weight"], sd["pos_embed.proj.bias"], stride=p)
    y = y.flatten(2).transpose(1, 2)
    pos = cropped_pos_embed(sd["pos_embed.pos_embed"], h, w, cfg["pos_embed_max_size"])
    return (y + pos.to(y.dtype)).to(y.dtype)


# --------------------------------------------------------------------------
# norms (Appendix A.2)
# --------------------------------------------------------------------------

def layer_norm_noaffine(x: Tensor, eps: float = 1e-6) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), None, None, eps)


def rms_norm(x: Tensor, weight: Tensor, eps: float) -> Tensor:
    var = x.float().pow(2).mean(-1, keepdim=True)
    return x * torch.rsqrt(var + eps) * weight


# --------------------------------------------------------------------------
# attention (Appendix A.3 / A.4)
# --------------------------------------------------------------------------

def sdpa(q: Tensor, k: Tensor, v: Tensor, mask: Optional[Tensor] = None) -> Tensor:
    """Naive softmax(QK^T / sqrt(d)) V in fp32; q,k,v [B,H,L,d]; mask bool
    [B,1|H,Lq,Lk] (True = attend), as F.scaled_dot_product_attention."""
    s = torch.matmul(q, k.transpose(-1, -2)) * (q.shape[-1] ** -0.5)
    if mask is not None:
        s = s.masked_fill(~mask, float("-inf"))
    return torch.matmul(torch.softmax(s, dim=-1), v)


def _heads(x: Tensor, heads: int) -> Tensor:
    b, l, c = x.shape
    return x.view(b, l, heads, c // heads).transpose(1, 2)


def _unheads(x: Tensor) -> Tensor:
    b, h, l, d = x.shape
    return x.transpose(1, 2).reshape(b, l, h * d)


def joint_attention(sd: SD, p: str, heads: int, h: Tensor, c: Optional[Tensor],
                    context_pre_only: bool, eps: float = 1e-6):
    """diffusers ``JointAttnProcessor2_0`` with qk_norm='rms_norm'."""
    q = _heads(linear(sd, p + ".to_q", h), heads)
    k = _heads(linear(sd, p + ".to_k", h), heads)
    v = _heads(linear(sd, p + ".to_v", h), heads)
    q = rms_norm(q, sd[p + ".norm_q.weight"], eps)
    k = rms_norm(k, sd[p + ".norm_k.weight"], eps)
    n = h.shape[1]
    if c is not None:
        cq = _heads(linear(sd, p + ".add_q_proj", c), heads)
        ck = _heads(linear(sd, p + ".add_k_proj", c), heads)
        cv = _heads(linear(sd, p + ".add_v_proj", c), heads)
        cq = rms_norm(cq, sd[p + ".norm_added_q.weight"], eps)
        ck = rms_norm(ck, sd[p + ".norm_added_k.weight"], eps)
        q = torch.cat([q, cq], dim=2)   # sample first, context second
        k = torch.cat([k, ck], dim=2)
        v = torch.cat([v, cv], dim=2)
    o = _unheads(sdpa(q, k, v))
    ho = linear(sd, p + ".to_out.0", o[:, :n])
    co = None
    if c is not None and not context_pre_only:
        co = linear(sd, p + ".to_add_out", o[:, n:])
    return ho, co


def feed_forward(sd: SD, p: str, x: Tensor, activation: str) -> Tensor:
    """diffusers ``FeedForward`` (mult 4, dropout 0)."""
    y = linear(sd, p + ".net.0.proj", x)
    if activation == "geglu":
        hcat, gate = y.chunk(2, dim=-1)
        y = hcat * F.gelu(gate)                      # exact erf GELU
    elif activation == "gelu-approximate":
        y = F.gelu(y, approximate="tanh")
    else:
        raise ValueError(activation)
    return linear(sd, p + ".net.2", y)


def joint_transformer_block(sd: SD, p: str, cfg: dict, i: int, h: Tensor,
                            c: Tensor, temb: Tensor):
    """diffusers ``JointTransformerBlock.forward`` (called at
    crossview_temporal_dit.py:517-521)."""
    heads = cfg["num_attention_heads"]
    dual = i in cfg.get("dual_attention_layers", ())
    pre_only = i == cfg["num_layers"] - 1
    st = F.silu(temb)

    emb = linear(sd, p + ".norm1.linear", st)
    if dual:
        (sh_msa, sc_msa, g_msa, sh_mlp, sc_mlp, g_mlp,
         sh_msa2, sc_msa2, g_msa2) = emb.chunk(9, dim=1)
    else:
        sh_msa, sc_msa, g_msa, sh_mlp, sc_mlp, g_mlp = emb.chunk(6, dim=1)
    nh0 = layer_norm_noaffine(h)
    nh = nh0 * (1 + sc_msa[:, None]) + sh_msa[:, None]
    if dual:
        nh2 = nh0 * (1 + sc_msa2[:, None]) + sh_msa2[:, None]

    cemb = linear(sd, p + ".norm1_context.linear", st)
    if pre_only:
        c_sc, c_sh = cemb.chunk(2, dim=1)            # AdaLayerNormContinuous: scale first
        nc = layer_norm_noaffine(c) * (1 + c_sc)[:, None] + c_sh[:, None]
    else:
        c_sh_msa, c_sc_msa, c_g_msa, c_sh_mlp, c_sc_mlp, c_g_mlp = cemb.chunk(6, dim=1)
        nc = layer_norm_noaffine(c) * (1 + c_sc_msa[:, None]) + c_sh_msa[:, None]

    a, ca = joint_attention(sd, p + ".attn", heads, nh, nc, pre_only)
    h = h + g_msa[:, None] * a
    if dual:
        a2, _ = joint_attention(sd, p + ".attn2", heads, nh2, None, False)
        h = h + g_msa2[:, None] * a2
    nh = layer_norm_noaffine(h) * (1 + sc_mlp[:, None]) + sh_mlp[:, None]
    h = h + g_mlp[:, None] * feed_forward(sd, p + ".ff", nh, "gelu-approximate")

    if pre_only:
        c = None
    else:
        c = c + c_g_msa[:, None] * ca
        nc = layer_norm_noaffine(c) * (1 + c_sc_mlp[:, None]) + c_sh_mlp[:, None]
        c = c + c_g_mlp[:, None] * feed_forward(sd, p + ".ff_context", nc, "gelu-approximate")
    return c, h


# --------------------------------------------------------------------------
# reference-owned blocks
# --------------------------------------------------------------------------

def vt_attention(sd: SD, p: str, heads: int, y: Tensor, mask: Optional[Tensor] = None) -> Tensor:
    """``VTSelfAttentionBlock.attn1``: diffusers Attention(bias=False, out_bias=True, qk_norm rms eps 1e-5) with
    AttnProcessor2_0 as self-attention; mask bool [Bp, Lq, Lk] (True = attend).  The leaf the reference block calls
    at crossview_temporal.py:572-574."""
    q = _heads(linear(sd, p + ".to_q", y), heads)
    k = _heads(linear(sd, p + ".to_k", y), heads)
    v = _heads(linear(sd, p + ".to_v", y), heads)
    if (p + ".norm_q.weight") in sd:
        q = rms_norm(q, sd[p + ".norm_q.weight"], 1e-5)
        k = rms_norm(k, sd[p + ".norm_k.weight"], 1e-5)
    m = None if mask is None else mask[:, None]
    return linear(sd, p + ".to_out.0", _unheads(sdpa(q, k, v, m)))


def vt_self_attention_block(sd: SD, p: str, heads: int, x: Tensor,
                            mask: Optional[Tensor] = None) -> Tensor:
    """``VTSelfAttentionBlock.forward`` (crossview_temporal.py:562-582);
    attn1 = diffusers Attention(bias=False, out_bias=True, qk_norm rms eps 1e-5)
    with AttnProcessor2_0; mask bool [Bp, Lq, Lk] (True = attend)."""
    d = x.shape[-1]
    res = x
    y = F.layer_norm(x, (d,), sd[p + ".norm_in.weight"], sd[p + ".norm_in.bias"], 1e-5)
    x = feed_forward(sd, p + ".ff_in", y, "geglu") + res

    y = F.layer_norm(x, (d,), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], 1e-5)
    a = vt_attention(sd, p + ".attn1", heads, y, mask)
    x = a + x

    y = F.layer_norm(x, (d,), sd[p + ".norm3.weight"], sd[p + ".norm3.bias"], 1e-5)
    return feed_forward(sd, p + ".ff", y, "geglu") + x


def alpha_blender(sd: SD, p: str, a: Tensor, b: Tensor, image_only: Tensor) -> Tensor:
    """``AlphaBlender.forward`` with merge_strategy 'learned_with_images'
    (crossview_temporal.py:33-72): alpha = where(flag, 1, sigmoid(mix_factor))."""
    alpha = torch.where(image_only, torch.ones((1,), device=a.device),
                        torch.sigmoid(sd[p + ".mix_factor"])).to(a.dtype)
    alpha = alpha.view(*alpha.shape, *([1] * (a.dim() - alpha.dim())))
    return alpha * a + (1.0 - alpha) * b


def rearr(x: Tensor, pattern: str, **kw) -> Tensor:
    import einops
    return einops.rearrange(x, pattern, **kw)


def crossview_block_and_mix(sd: SD, cfg: dict, k: int, h: Tensor, view_emb: Tensor,
                            B: int, T: int, V: int, width: int, height: int,
                            disable_crossview: Tensor, mask: Optional[Tensor]) -> Tensor:
    """``forward_crossview_block_and_mix_result`` (crossview_temporal_dit.py:223-327),
    'rowwise' and 'full' types (the two ctsd.py can drive; 'fuse'/'adj_fuse'
    need crossview_attention_index which ctsd.py never passes)."""
    heads = cfg["num_attention_heads"]
    p = f"crossview_transformer_blocks.{k}"
    x = h + view_emb
    typ = cfg["crossview_attention_type"]
    if typ == "rowwise":
        m = None
        if mask is not None:
            m = mask.repeat_interleave(width, 2).repeat_interleave(width, 1) \
                    .repeat_interleave(T * height, 0)
        x = rearr(x, "(bt v) (h w) c -> (bt h) (v w) c", w=width, v=V)
        x = vt_self_attention_block(sd, p, heads, x, m)
        x = rearr(x, "(bt h) (v w) c -> (bt v) (h w) c", bt=B * T, v=V)
    elif typ == "full":
        x = rearr(x, "(bt v) (h w) c -> bt (h v w) c", v=V, w=width)
        x = vt_self_attention_block(sd, p, heads, x, mask)
        x = rearr(x, "bt (h v w) c -> (bt v) (h w) c", v=V, w=width)
    else:
        raise ValueError(f"Not support {typ}")
    return alpha_blender(sd, f"view_mixers.{k}",
                         h.view(B, T * V, *h.shape[1:]), x.view(B, T * V, *x.shape[1:]),
                         disable_crossview).flatten(0, 1)


def temporal_block_and_mix(sd: SD, cfg: dict, k: int, h: Tensor, seq_emb: Tensor,
                           B: int, T: int, V: int, width: int,
                           disable_temporal: Tensor) -> Tensor:
    """``forward_temporal_block_and_mix_result`` (crossview_temporal_dit.py:329-370)."""
    heads = cfg["num_attention_heads"]
    p = f"temporal_transformer_blocks.{k}"
    x = h + seq_emb
    typ = cfg["temporal_attention_type"]
    if typ == "full":
        x = rearr(x, "(b t v) hw c -> (b v) (t hw) c", b=B, t=T)
        x = vt_self_attention_block(sd, p, heads, x)
        x = rearr(x, "(b v) (t hw) c -> (b t v) hw c", b=B, t=T)
    elif typ == "rowwise":
        x = rearr(x, "(b t v) (h w) c -> (b v h) (t w) c", b=B, v=V, w=width)
        x = vt_self_attention_block(sd, p, heads, x)
        x = rearr(x, "(b v h) (t w) c -> (b t v) (h w) c", b=B, v=V, w=width)
    else:   # "pointwise" falls to the reference's else branch
        x = rearr(x, "(b t v) hw c -> (b v hw) t c", b=B, t=T)
        x = vt_self_attention_block(sd, p, heads, x)
        x = rearr(x, "(b v hw) t c -> (b t v) hw c", b=B, v=V, t=T)
    return alpha_blender(sd, f"time_mixers.{k}",
                         h.view(B, T * V, *h.shape[1:]), x.view(B, T * V, *x.shape[1:]),
                         disable_temporal).flatten(0, 1)


def adapter_block(sd: SD, cfg: dict, i: int, x: Tensor, prefix: str = "condition_image_adapter") -> Tensor:
    """diffusers T2I ``AdapterBlock`` i (the leaf ``ImageAdapter.forward`` loops over, adapters.py:44-50):
    [AvgPool2d(2, ceil_mode) if down] -> [Conv1x1 if in != out] -> num_res_blocks x AdapterResnetBlock
    (x + Conv1x1(ReLU(Conv3x3(x))))."""
    ac = cfg["condition_image_adapter_config"]
    b = f"{prefix}.body.{i}"
    if ac["is_downblocks"][i]:
        x = F.avg_pool2d(x, kernel_size=2, stride=2, ceil_mode=True)
    if (b + ".in_conv.weight") in sd:
        x = F.conv2d(x, sd[b + ".in_conv.weight"], sd[b + ".in_conv.bias"])
    for j in range(ac.get("num_res_blocks", 2)):
        r = f"{b}.resnets.{j}"
        hh = F.relu(F.conv2d(x, sd[r + ".block1.weight"], sd[r + ".block1.bias"], padding=1))
        x = x + F.conv2d(hh, sd[r + ".block2.weight"], sd[r + ".block2.bias"])
    return x


def image_adapter(sd: SD, cfg: dict, x: Tensor, prefix: str = "condition_image_adapter") -> List[Tensor]:
    """``ImageAdapter.forward`` (src/dwm/models/adapters.py:40-60): PixelUnshuffle, the AdapterBlocks, optional zero
    1x1 convs.  x [..., C, H, W]; returns features [*base_shape, C_i, h_i, w_i]."""
    ac = cfg["condition_image_adapter_config"]
    base_shape = x.shape[:-3]
    x = F.pixel_unshuffle(x.flatten(0, -4), ac.get("downscale_factor", 8))
    feats = []
    for i in range(len(ac["is_downblocks"])):
        x = adapter_block(sd, cfg, i, x, prefix)
        x_out = x
        if ac.get("use_zero_convs", False):
            x_out = F.conv2d(x, sd[f"{prefix}.zero_convs.{i}.weight"], sd[f"{prefix}.zero_convs.{i}.bias"])
        feats.append(x_out.view(*base_shape, *x_out.shape[1:]))
    return feats


# --------------------------------------------------------------------------
# the model forward
# --------------------------------------------------------------------------

def positional_encoding(coords: Tensor, num_octaves: int, start_octave: int = 0) -> Tensor:
    """``PositionalEncoding.forward`` (crossview_temporal_dit.py:11-36): coords [B, P, dim] ->
    [B, P, 2 * dim * num_octaves] = cat(sin, cos) of coords * 2^k * pi, laid out dim-major."""
    Bn, P, dim = coords.shape
    mult = 2.0 ** torch.arange(start_octave, start_octave + num_octaves).float().to(coords) * math.pi
    sc = coords.unsqueeze(-1) * mult.view(1, 1, 1, -1)
    return torch.cat((torch.sin(sc).reshape(Bn, P, dim * num_octaves), torch.cos(sc).reshape(Bn, P, dim * num_octaves)), -1)


def ray_encoder(sd: SD, pos: Tensor, rays: Tensor, prefix: str = "rayencoder") -> Tensor:
    """``RayEncoder.forward`` (crossview_temporal_dit.py:39-64; pos 8 octaves, rays 4 octaves, proj 72 -> D, no bias):
    pos [I, 3], rays [I, h, w, 3] -> [I, h, w, D]."""
    I, hh, ww, _ = rays.shape
    pe = positional_encoding(pos.unsqueeze(1), 8).view(I, 1, 1, -1).repeat(1, hh, ww, 1)
    re = positional_encoding(rays.flatten(1, 2), 4).view(I, hh, ww, -1)
    return F.linear(torch.cat((pe, re), -1), sd[prefix + ".proj.weight"])


def get_rays(camera_intrinsics: Tensor, camera_transforms: Tensor, target_size) -> tuple:
    """``get_rays`` (crossview_temporal_dit.py:66-102): intrinsics [I,3,3], cam2world [I,4,4] ->
    (rays_o [I,3], unit rays_d [I,H,W,3]) through the pixel centres of an H x W grid."""
    dtype = camera_transforms.dtype
    ct, ci = camera_transforms.float(), camera_intrinsics.float()
    Hh, Ww = (target_size, target_size) if isinstance(target_size, int) else target_size
    dev = ct.device
    xs = torch.arange(Ww, device=dev).float().repeat(Hh) + 0.5          # x fastest (the transposed meshgrid of :84-88)
    ys = torch.arange(Hh, device=dev).float().repeat_interleave(Ww) + 0.5
    pts = torch.stack([xs, ys, torch.ones_like(xs)])                    # [3, H*W]
    d = ct[:, :3, :3] @ (torch.inverse(ci) @ pts.unsqueeze(0))
    d = d / torch.norm(d, dim=1, keepdim=True)
    return ct[:, :3, 3].to(dtype), d.transpose(1, 2).reshape(-1, Hh, Ww, 3).to(dtype)


def explicit_view_embedding(sd: SD, camera_intrinsics_norm: Tensor, camera2referego: Tensor, height: int, width: int) -> Tensor:
    """the 'explicit' branch of the forward (crossview_temporal_dit.py:440-458): normalised intrinsics scaled to the token
    grid, rays in the reference ego frame, RayEncoder -> per-token embedding [I, h*w, D]."""
    K = camera_intrinsics_norm.clone()
    K[..., 0, 0] = K[..., 0, 0] * width
    K[..., 1, 1] = K[..., 1, 1] * height
    K[..., 0, 2] = K[..., 0, 2] * width
    K[..., 1, 2] = K[..., 1, 2] * height
    ro, rd = get_rays(K.flatten(0, 2), camera2referego.flatten(0, 2), (height, width))
    return ray_encoder(sd, ro, rd).flatten(1, 2)


def dit_forward(sd: SD, cfg: dict, sample: Tensor, timestep: Tensor,
                encoder_hidden_states: Tensor, pooled_projections: Tensor,
                disable_crossview: Optional[Tensor] = None,
                disable_temporal: Optional[Tensor] = None,
                crossview_attention_mask: Optional[Tensor] = None,
                added_time_ids: Optional[Tensor] = None,
                condition_image_tensor: Optional[Tensor] = None,
                trace: Optional[dict] = None,
                camera_intrinsics_norm: Optional[Tensor] = None,
                camera2referego: Optional[Tensor] = None) -> Tensor:
    """``DiTCrossviewTemporalConditionModel.forward`` (crossview_temporal_dit.py:372-630)
    for 6-D inputs, implicit / no perspective modelling, no image adapter and no
    mask module (the configuration of examples/ctsd_35_6views_video_generation.json).
    Returns the prediction [B,T,V,C,H,W] (element [0][0] of the reference's 3-tuple).
    ``trace``: optional dict that receives named intermediate tensors."""
    B, T, V, _, H, W = sample.shape
    p = cfg["patch_size"]
    height, width = H // p, W // p
    D = cfg["num_attention_heads"] * cfg["attention_head_dim"]

    h = patch_embed(sd, cfg, sample.flatten(0, 2))
    pooled = pooled_projections.flatten(0, 2)
    c = linear(sd, "context_embedder", encoder_hidden_states.flatten(0, 2))

    # CombinedTimestepTextProjEmbeddings
    t_proj = timesteps_sinusoid(timestep.flatten(), 256)
    temb = timestep_embedding_mlp(sd, "time_text_embed.timestep_embedder", t_proj.to(pooled.dtype)) \
        + timestep_embedding_mlp(sd, "time_text_embed.text_embedder", pooled)

    view_cam_emb = 0
    if cfg.get("perspective_modeling_type", "") == "implicit":
        ve = timesteps_sinusoid(added_time_ids.flatten(), 256).to(h.dtype)
        view_cam_emb = timestep_embedding_mlp(sd, "view_embedding", ve.view(B * T * V, -1)).unsqueeze(1)
    elif cfg.get("perspective_modeling_type", "") == "explicit":                 # :440-458, per-token embedding [I, N, D]
        view_cam_emb = explicit_view_embedding(sd, camera_intrinsics_norm, camera2referego, height, width).to(h.dtype)

    condition_residuals = None
    if cfg.get("condition_image_adapter_config") is not None and condition_image_tensor is not None:
        condition_residuals = image_adapter(sd, cfg, condition_image_tensor)     # :459-462

    if trace is not None:
        trace["hidden0"] = h
        trace["context0"] = c
        trace["temb"] = temb

    t_layers = list(cfg.get("temporal_block_layers") or [])
    v_layers = list(cfg.get("crossview_block_layers") or [])
    for i in range(cfg["num_layers"]):
        if condition_residuals is not None and len(condition_residuals) > 0:                    # :491-494
            h = h + condition_residuals.pop(0).flatten(0, 2).flatten(2).permute(0, 2, 1)
        c, h = joint_transformer_block(sd, f"transformer_blocks.{i}", cfg, i, h, c, temb)
        if trace is not None:
            trace[f"joint{i}"] = h

        if cfg.get("enable_temporal") and i in t_layers:
            k = t_layers.index(i)
            idx = torch.arange(T, device=h.device).unsqueeze(0).unsqueeze(-1).repeat(B, 1, V)
            seq_emb = timesteps_sinusoid(idx.flatten(), D).to(h.dtype)
            seq_emb = timestep_embedding_mlp(sd, f"time_pos_embeds.{k}", seq_emb).unsqueeze(1)
            if cfg.get("enable_crossview") and not cfg.get("disable_view_emb_on_temporal_module", False):
                seq_emb = seq_emb + view_cam_emb
            h = temporal_block_and_mix(sd, cfg, k, h, seq_emb, B, T, V, width, disable_temporal)
            if trace is not None:
                trace[f"temporal{i}"] = h

        if cfg.get("enable_crossview") and i in v_layers:
            k = v_layers.index(i)
            idx = torch.arange(V, device=h.device).unsqueeze(0).unsqueeze(0).repeat(B, T, 1)
            view_emb = timesteps_sinusoid(idx.flatten(), D).to(h.dtype)
            view_emb = timestep_embedding_mlp(sd, f"view_pos_embeds.{k}", view_emb).unsqueeze(1)
            view_emb = view_emb + view_cam_emb
            h = crossview_block_and_mix(sd, cfg, k, h, view_emb, B, T, V, width, height,
                                        disable_crossview, crossview_attention_mask)
            if trace is not None:
                trace[f"crossview{i}"] = h

    # norm_out (AdaLayerNormContinuous) + proj_out + unpatchify
    emb = linear(sd, "norm_out.linear", F.silu(temb))
    scale, shift = emb.chunk(2, dim=1)
    h = layer_norm_noaffine(h) * (1 + scale)[:, None] + shift[:, None]
    h = linear(sd, "proj_out", h)
    oc = cfg["out_channels"]
    h = h.reshape(h.shape[0], height, width, p, p, oc)
    h = torch.einsum("nhwpqc->nchpwq", h)
    return h.reshape(B, T, V, oc, height * p, width * p)


# --------------------------------------------------------------------------
# VAE decoder (diffusers AutoencoderKL 0.31.0; called at ctsd.py:1634-1640)
# --------------------------------------------------------------------------

def _vae_resnet(sd: SD, p: str, x: Tensor, groups: int, eps: float) -> Tensor:
    """diffusers ResnetBlock2D with temb_channels=None, output_scale_factor=1."""
    h = F.silu(F.group_norm(x, groups, sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], eps))
    h = F.conv2d(h, sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], padding=1)
    h = F.silu(F.group_norm(h, groups, sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], eps))
    h = F.conv2d(h, sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], padding=1)
    if (p + ".conv_shortcut.weight") in sd:
        x = F.conv2d(x, sd[p + ".conv_shortcut.weight"], sd[p + ".conv_shortcut.bias"])
    return x + h


def _vae_attention(sd: SD, p: str, x: Tensor, groups: int, eps: float) -> Tensor:
    """diffusers Attention in the VAE mid block: heads = 1, GroupNorm first, residual connection."""
    b, c, hh, ww = x.shape
    res = x
    y = F.group_norm(x.view(b, c, hh * ww), groups, sd[p + ".group_norm.weight"], sd[p + ".group_norm.bias"], eps)
    y = y.transpose(1, 2)
    q, k, v = linear(sd, p + ".to_q", y), linear(sd, p + ".to_k", y), linear(sd, p + ".to_v", y)
    o = sdpa(q[:, None], k[:, None], v[:, None])[:, 0]
    o = linear(sd, p + ".to_out.0", o)
    return o.transpose(1, 2).reshape(b, c, hh, ww) + res


def vae_decode(sd: SD, vcfg: dict, z: Tensor) -> Tensor:
    """``AutoencoderKL.decode`` -> ``Decoder.forward`` (no post_quant_conv, SD 3 / 3.5):
    conv_in -> mid (resnet, attention, resnet) -> 4 up blocks (3 resnets [+ nearest-2x + conv]) ->
    GroupNorm -> SiLU -> conv_out."""
    g, eps = vcfg.get("norm_num_groups", 32), 1e-6
    if "post_quant_conv.weight" in sd:         # SD 2.1 VAE: AutoencoderKL.decode applies post_quant_conv (1x1) first
        z = F.conv2d(z, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])
    x = F.conv2d(z, sd["decoder.conv_in.weight"], sd["decoder.conv_in.bias"], padding=1)
    x = _vae_resnet(sd, "decoder.mid_block.resnets.0", x, g, eps)
    if vcfg.get("mid_block_add_attention", True):
        x = _vae_attention(sd, "decoder.mid_block.attentions.0", x, g, eps)
    x = _vae_resnet(sd, "decoder.mid_block.resnets.1", x, g, eps)
    nb = len(vcfg["block_out_channels"])
    for i in range(nb):
        for j in range(vcfg.get("layers_per_block", 2) + 1):
            x = _vae_resnet(sd, f"decoder.up_blocks.{i}.resnets.{j}", x, g, eps)
        if i != nb - 1:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = F.conv2d(x, sd[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"],
                         sd[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"], padding=1)
    x = F.silu(F.group_norm(x, g, sd["decoder.conv_norm_out.weight"], sd["decoder.conv_norm_out.bias"], eps))
    return F.conv2d(x, sd["decoder.conv_out.weight"], sd["decoder.conv_out.bias"], padding=1)


def vae_encode_moments(sd: SD, vcfg: dict, x: Tensor) -> Tensor:
    """``AutoencoderKL.encode`` -> ``Encoder.forward`` (no quant_conv): conv_in -> 4 DownEncoderBlock2D
    (2 resnets [+ Downsample2D: F.pad(0,1,0,1) + 3x3 stride-2 conv]) -> mid -> GroupNorm -> SiLU -> conv_out;
    returns the moments [I, 2*latent, h, w] (mean ‖ logvar) of the DiagonalGaussianDistribution
    (ctsd.py:1213-1218 samples it, :1689-1694 takes the mode)."""
    g, eps = vcfg.get("norm_num_groups", 32), 1e-6
    h = F.conv2d(x, sd["encoder.conv_in.weight"], sd["encoder.conv_in.bias"], padding=1)
    nb = len(vcfg["block_out_channels"])
    for i in range(nb):
        for j in range(vcfg.get("layers_per_block", 2)):
            h = _vae_resnet(sd, f"encoder.down_blocks.{i}.resnets.{j}", h, g, eps)
        if i != nb - 1:
            h = F.pad(h, (0, 1, 0, 1), mode="constant", value=0)
            h = F.conv2d(h, sd[f"encoder.down_blocks.{i}.downsamplers.0.conv.weight"],
                         sd[f"encoder.down_blocks.{i}.downsamplers.0.conv.bias"], stride=2)
    h = _vae_resnet(sd, "encoder.mid_block.resnets.0", h, g, eps)
    if vcfg.get("mid_block_add_attention", True):
        h = _vae_attention(sd, "encoder.mid_block.attentions.0", h, g, eps)
    h = _vae_resnet(sd, "encoder.mid_block.resnets.1", h, g, eps)
    h = F.silu(F.group_norm(h, g, sd["encoder.conv_norm_out.weight"], sd["encoder.conv_norm_out.bias"], eps))
    m = F.conv2d(h, sd["encoder.conv_out.weight"], sd["encoder.conv_out.bias"], padding=1)
    if "quant_conv.weight" in sd:              # SD 2.1 VAE: 1x1 quant_conv on the moments
        m = F.conv2d(m, sd["quant_conv.weight"], sd["quant_conv.bias"])
    return m


def vae_decoder_shapes(vcfg: dict) -> Dict[str, tuple]:
    ch = list(vcfg["block_out_channels"])
    lc, oc = vcfg.get("latent_channels", 16), vcfg.get("out_channels", 3)
    S: Dict[str, tuple] = {}

    def conv(n, i, o, k):
        S[n + ".weight"] = (o, i, k, k)
        S[n + ".bias"] = (o,)

    def norm(n, c):
        S[n + ".weight"] = (c,)
        S[n + ".bias"] = (c,)

    def resnet(n, i, o):
        norm(n + ".norm1", i); conv(n + ".conv1", i, o, 3); norm(n + ".norm2", o); conv(n + ".conv2", o, o, 3)
        if i != o:
            conv(n + ".conv_shortcut", i, o, 1)

    conv("decoder.conv_in", lc, ch[-1], 3)
    resnet("decoder.mid_block.resnets.0", ch[-1], ch[-1])
    resnet("decoder.mid_block.resnets.1", ch[-1], ch[-1])
    if vcfg.get("mid_block_add_attention", True):
        a = "decoder.mid_block.attentions.0"
        norm(a + ".group_norm", ch[-1])
        for nm in ("to_q", "to_k", "to_v", "to_out.0"):
            S[f"{a}.{nm}.weight"] = (ch[-1], ch[-1])
            S[f"{a}.{nm}.bias"] = (ch[-1],)
    prev = ch[-1]
    for i, o in enumerate(ch[::-1]):
        for j in range(vcfg.get("layers_per_block", 2) + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", prev if j == 0 else o, o)
        if i != len(ch) - 1:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", o, o, 3)
        prev = o
    norm("decoder.conv_norm_out", ch[0])
    conv("decoder.conv_out", ch[0], oc, 3)
    # encoder
    conv("encoder.conv_in", vcfg.get("in_channels", 3), ch[0], 3)
    prev = ch[0]
    for i, o in enumerate(ch):
        for j in range(vcfg.get("layers_per_block", 2)):
            resnet(f"encoder.down_blocks.{i}.resnets.{j}", prev if j == 0 else o, o)
        if i != len(ch) - 1:
            conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", o, o, 3)
        prev = o
    resnet("encoder.mid_block.resnets.0", ch[-1], ch[-1])
    resnet("encoder.mid_block.resnets.1", ch[-1], ch[-1])
    if vcfg.get("mid_block_add_attention", True):
        a = "encoder.mid_block.attentions.0"
        norm(a + ".group_norm", ch[-1])
        for nm in ("to_q", "to_k", "to_v", "to_out.0"):
            S[f"{a}.{nm}.weight"] = (ch[-1], ch[-1])
            S[f"{a}.{nm}.bias"] = (ch[-1],)
    norm("encoder.conv_norm_out", ch[-1])
    conv("encoder.conv_out", ch[-1], 2 * lc, 3)
    if vcfg.get("use_quant_conv", False):
        conv("quant_conv", 2 * lc, 2 * lc, 1)
    if vcfg.get("use_post_quant_conv", False):
        conv("post_quant_conv", lc, lc, 1)
    return S


def make_vae_state_dict(vcfg: dict, seed: int = 0) -> SD:
    gen = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in vae_decoder_shapes(vcfg).items():
        if len(shape) == 1:
            v = torch.randn(*shape, generator=gen) * 0.05
            sd[name] = 1.0 + v if name.endswith(".weight") else v
        else:
            fan_in = 1
            for s_ in shape[1:]:
                fan_in *= s_
            std = fan_in ** -0.5
            if ".conv2." in name or ".to_out." in name:
                std *= 0.5
            sd[name] = torch.randn(*shape, generator=gen) * std
    return sd


# --------------------------------------------------------------------------
# scheduler + denoise loop (ctsd.py:1496-1575; diffusers FlowMatchEulerDiscreteScheduler)
# --------------------------------------------------------------------------

def flow_match_sigmas(num_inference_steps: int, shift: float = 3.0,
                      num_train_timesteps: int = 1000) -> Tensor:
    """``FlowMatchEulerDiscreteScheduler.set_timesteps`` (diffusers 0.31.0, SD3.5
    scheduler config shift=3.0, no dynamic shifting): returns sigmas with a
    trailing 0 (length n+1); timesteps = sigmas[:-1] * 1000."""
    ts = torch.linspace(1, num_train_timesteps, num_train_timesteps).flip(0) / num_train_timesteps
    ts = shift * ts / (1 + (shift - 1) * ts)
    sigma_max, sigma_min = ts[0].item(), ts[-1].item()
    t = torch.linspace(sigma_max * num_train_timesteps, sigma_min * num_train_timesteps,
                       num_inference_steps)
    sig = t / num_train_timesteps
    sig = shift * sig / (1 + (shift - 1) * sig)
    return torch.cat([sig, torch.zeros(1)]).float()


def flow_match_train_sigmas(shift: float = 3.0, num_train_timesteps: int = 1000) -> Tensor:
    """sigmas of the *training* ``FlowMatchEulerDiscreteScheduler`` as its constructor builds them
    (diffusers 0.31.0): timesteps = linspace(1, n, n)[::-1]; sigma = t/n; shifted; scheduler.timesteps
    = sigma * n.  ``sd3_get_sigmas`` (ctsd.py:1263-1266) looks sigma up by timestep, i.e. sigmas[idx]."""
    s = torch.linspace(1, num_train_timesteps, num_train_timesteps).flip(0) / num_train_timesteps
    return (shift * s / (1 + (shift - 1) * s)).float()


def train_loss(sd: SD, cfg: dict, latents: Tensor, conditions: dict, timestep_indices: Tensor, noise: Tensor,
               shift: float = 3.0, loss_coef: float = 1.0, num_train_timesteps: int = 1000) -> Tensor:
    """SD 3 branch of ``train_step`` (ctsd.py:1255-1272, 1355-1370) for given noise and timestep indices
    (the reference draws them from a CPU generator, :1229, :1256-1262):
    x_t = sigma*eps + (1-sigma)*x0; pred = model(x_t, 1000*sigma); x0_hat = pred*(-sigma) + x_t;
    loss = mse(x0_hat, x0) * coef.  latents [B,T,V,C,H,W]; timestep_indices [B]."""
    B, T, V = latents.shape[:3]
    sig = flow_match_train_sigmas(shift, num_train_timesteps).to(latents.device)[timestep_indices]
    timesteps = (sig * num_train_timesteps).view(B, 1, 1).expand(B, T, V)
    sg = sig.view(B, 1, 1, 1, 1, 1)
    noisy = sg * noise + (1.0 - sg) * latents
    pred = dit_forward(sd, cfg, noisy, timesteps, **conditions)
    x0_hat = pred * (-sg) + noisy
    return torch.nn.functional.mse_loss(x0_hat.float(), latents.float(), reduction="mean") * loss_coef


def denoise(sd: SD, cfg: dict, latents: Tensor, conditions: dict, steps: int,
            guidance_scale: float, shift: float = 3.0, stop: Optional[int] = None, start: int = 0,
            image_latents: Optional[Tensor] = None, reference_frame_count: int = 0,
            diffusion_forcing: bool = False, take_time: int = 0, clear_reference_frame_count: int = 0):
    """``inference_pipeline`` hot loop (ctsd.py:1496-1575) with classifier-free guidance and the
    FlowMatch-Euler scheduler, in its three modes: full-sequence; reference-frame injection
    (:1514-1526, :1623-1627); diffusion forcing with per-frame timestep indices and
    ``step_by_indices`` (:1498-1507, :1554-1572; schedulers/temporal_independent.py:176-197).
    ``conditions`` hold the CFG-doubled tensors ([2B,...], uncond first).  latents fp32 [B,T,V,C,H,W]."""
    sigmas = flow_match_sigmas(steps, shift)
    timesteps = sigmas[:-1] * 1000
    if diffusion_forcing and image_latents is not None:
        lat = image_latents.float()
        image_latents = None
    else:
        lat = latents.float()
    B, T, V = lat.shape[:3]
    spi = steps // (T - clear_reference_frame_count) if diffusion_forcing else None
    for i in range(start, steps if stop is None else stop):
        if diffusion_forcing:
            idx = torch.tensor([min(i - take_time * spi, max(0, i - j * spi)) for j in range(T)])
            ts = timesteps[idx].view(1, T, 1).repeat(B, 1, V)
        else:
            ts = timesteps[i].reshape(1, 1, 1).repeat(B, T, V)
        x = lat
        if not diffusion_forcing and image_latents is not None:
            x = torch.cat([image_latents[:, :reference_frame_count].float(), lat[:, reference_frame_count:]], 1)
            ts = torch.cat([torch.zeros(B, reference_frame_count, V), ts[:, reference_frame_count:]], 1)
        pred = dit_forward(sd, cfg, torch.cat([x, x]), torch.cat([ts, ts]), **conditions)
        u, cnd = pred.chunk(2)
        noise_pred = u + guidance_scale * (cnd - u)
        if diffusion_forcing:
            ii = idx.view(1, T, 1, 1, 1, 1)
            staging = lat + (sigmas[ii + 1] - sigmas[ii]) * noise_pred.float()
            in_range = torch.tensor([i - j * spi >= 0 for j in range(T)]).view(1, T, 1, 1, 1, 1)
            lat = torch.where(in_range, staging, lat)
        else:
            lat = lat + (sigmas[i + 1] - sigmas[i]) * noise_pred.float()
    if not diffusion_forcing and image_latents is not None:
        lat = torch.cat([image_latents[:, :reference_frame_count].float(), lat[:, reference_frame_count:]], 1)
    return lat


# --------------------------------------------------------------------------
# synthetic weights / inputs: seeded, every branch non-degenerate
# --------------------------------------------------------------------------

def make_config(**over) -> dict:
    """Model kwargs of examples/ctsd_35_6views_video_generation.json:45-107."""
    cfg = dict(
        dual_attention_layers=list(range(13)), attention_head_dim=64,
        caption_projection_dim=1536, in_channels=16, joint_attention_dim=4096,
        num_attention_heads=24, num_layers=24, out_channels=16, patch_size=2,
        pooled_projection_dim=2048, pos_embed_max_size=384, qk_norm="rms_norm",
        qk_norm_on_additional_modules="rms_norm", sample_size=128,
        perspective_modeling_type="implicit", projection_class_embeddings_input_dim=2816,
        enable_crossview=True, crossview_attention_type="rowwise",
        crossview_block_layers=[1, 5, 9, 13, 17, 21],
        enable_temporal=True, temporal_attention_type="rowwise",
        temporal_block_layers=[2, 3, 6, 7, 10, 11, 14, 15, 18, 19, 22, 23],
        mixer_type="AlphaBlender", merge_factor=2,
    )
    cfg.update(over)
    return cfg


def param_shapes(cfg: dict) -> Dict[str, tuple]:
    """Name → shape of every parameter/buffer of the reference module tree."""
    D = cfg["num_attention_heads"] * cfg["attention_head_dim"]
    hd = cfg["attention_head_dim"]
    p = cfg["patch_size"]
    S: Dict[str, tuple] = {}

    def lin(name, i, o, bias=True):
        S[name + ".weight"] = (o, i)
        if bias:
            S[name + ".bias"] = (o,)

    S["pos_embed.proj.weight"] = (D, cfg["in_channels"], p, p)
    S["pos_embed.proj.bias"] = (D,)
    S["pos_embed.pos_embed"] = (1, cfg["pos_embed_max_size"] ** 2, D)
    lin("context_embedder", cfg["joint_attention_dim"], cfg["caption_projection_dim"])
    lin("time_text_embed.timestep_embedder.linear_1", 256, D)
    lin("time_text_embed.timestep_embedder.linear_2", D, D)
    lin("time_text_embed.text_embedder.linear_1", cfg["pooled_projection_dim"], D)
    lin("time_text_embed.text_embedder.linear_2", D, D)
    n = cfg["num_layers"]
    for i in range(n):
        b = f"transformer_blocks.{i}"
        dual = i in cfg.get("dual_attention_layers", ())
        pre = i == n - 1
        lin(b + ".norm1.linear", D, (9 if dual else 6) * D)
        lin(b + ".norm1_context.linear", D, (2 if pre else 6) * D)
        for nm in ("to_q", "to_k", "to_v", "add_q_proj", "add_k_proj", "add_v_proj", "to_out.0"):
            lin(b + ".attn." + nm, D, D)
        if not pre:
            lin(b + ".attn.to_add_out", D, D)
        for nm in ("norm_q", "norm_k", "norm_added_q", "norm_added_k"):
            S[b + f".attn.{nm}.weight"] = (hd,)
        if dual:
            for nm in ("to_q", "to_k", "to_v", "to_out.0"):
                lin(b + ".attn2." + nm, D, D)
            for nm in ("norm_q", "norm_k"):
                S[b + f".attn2.{nm}.weight"] = (hd,)
        lin(b + ".ff.net.0.proj", D, 4 * D)
        lin(b + ".ff.net.2", 4 * D, D)
        if not pre:
            lin(b + ".ff_context.net.0.proj", D, 4 * D)
            lin(b + ".ff_context.net.2", 4 * D, D)
    lin("norm_out.linear", D, 2 * D)
    lin("proj_out", D, p * p * cfg["out_channels"])
    if cfg.get("perspective_modeling_type", "") == "implicit":
        lin("view_embedding.linear_1", cfg["projection_class_embeddings_input_dim"], D)
        lin("view_embedding.linear_2", D, D)
    elif cfg.get("perspective_modeling_type", "") == "explicit":
        S["rayencoder.proj.weight"] = (D, 72)                      # RayEncoder(cond_proj_dim=72), crossview_temporal_dit.py:156-159

    ac = cfg.get("condition_image_adapter_config")
    if ac is not None:
        cin = ac.get("in_channels", 3) * ac.get("downscale_factor", 8) ** 2
        for i, ch in enumerate(ac["channels"]):
            b = f"condition_image_adapter.body.{i}"
            prev = cin if i == 0 else ac["channels"][i - 1]
            if prev != ch:
                S[b + ".in_conv.weight"] = (ch, prev, 1, 1)
                S[b + ".in_conv.bias"] = (ch,)
            for j in range(ac.get("num_res_blocks", 2)):
                S[f"{b}.resnets.{j}.block1.weight"] = (ch, ch, 3, 3)
                S[f"{b}.resnets.{j}.block1.bias"] = (ch,)
                S[f"{b}.resnets.{j}.block2.weight"] = (ch, ch, 1, 1)
                S[f"{b}.resnets.{j}.block2.bias"] = (ch,)
            if ac.get("use_zero_convs", False):
                S[f"condition_image_adapter.zero_convs.{i}.weight"] = (ch, ch, 1, 1)
                S[f"condition_image_adapter.zero_convs.{i}.bias"] = (ch,)

    def vt(b):
        for nm in ("norm_in", "norm1", "norm3"):
            S[b + f".{nm}.weight"] = (D,)
            S[b + f".{nm}.bias"] = (D,)
        lin(b + ".ff_in.net.0.proj", D, 8 * D)
        lin(b + ".ff_in.net.2", 4 * D, D)
        for nm in ("to_q", "to_k", "to_v"):
            lin(b + ".attn1." + nm, D, D, bias=False)
        lin(b + ".attn1.to_out.0", D, D)
        if cfg.get("qk_norm_on_additional_modules") == "rms_norm":
            S[b + ".attn1.norm_q.weight"] = (hd,)
            S[b + ".attn1.norm_k.weight"] = (hd,)
        lin(b + ".ff.net.0.proj", D, 8 * D)
        lin(b + ".ff.net.2", 4 * D, D)

    if cfg.get("enable_crossview"):
        for k in range(len(cfg["crossview_block_layers"])):
            lin(f"view_pos_embeds.{k}.linear_1", D, 4 * D)
            lin(f"view_pos_embeds.{k}.linear_2", 4 * D, D)
            vt(f"crossview_transformer_blocks.{k}")
            S[f"view_mixers.{k}.mix_factor"] = (1,)
    if cfg.get("enable_temporal"):
        for k in range(len(cfg["temporal_block_layers"])):
            lin(f"time_pos_embeds.{k}.linear_1", D, 4 * D)
            lin(f"time_pos_embeds.{k}.linear_2", 4 * D, D)
            vt(f"temporal_transformer_blocks.{k}")
            S[f"time_mixers.{k}.mix_factor"] = (1,)
    return S


def synth_param(name: str, shape: tuple, cfg: dict, gen: torch.Generator,
                device="cpu") -> Tensor:
    """Deterministic synthetic value for one parameter.  Matrices ~ N(0, 1/fan_in);
    norm weights 1 + small noise; biases small noise; modulation (AdaLN) linears
    damped so residual streams stay O(1) over 24 layers; mix_factor = merge_factor."""
    def rn(*s, std=1.0):
        return torch.randn(*s, generator=gen, dtype=torch.float32) * std

    if name == "pos_embed.pos_embed":
        D = shape[-1]
        base = cfg["sample_size"] // cfg["patch_size"]
        return make_pos_embed_table(D, cfg["pos_embed_max_size"], base).to(device)
    if name.endswith("mix_factor"):
        return torch.full(shape, float(cfg.get("merge_factor", 2))).to(device)
    if len(shape) == 1:
        is_norm_w = name.endswith(".weight")
        v = rn(*shape, std=0.05)
        return ((1.0 + v) if is_norm_w else v).to(device)
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    std = fan_in ** -0.5
    if ".norm1.linear" in name or ".norm1_context.linear" in name or name.startswith("norm_out.linear"):
        std *= 0.5
    if name.startswith("condition_image_adapter.") and (".block2." in name or ".zero_convs." in name):
        std *= 0.5
    return rn(*shape, std=std).to(device)


def make_state_dict(cfg: dict, seed: int = 0, device="cpu", dtype=torch.float32) -> SD:
    gen = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in param_shapes(cfg).items():
        sd[name] = synth_param(name, shape, cfg, gen, device).to(dtype)
    return sd


def ring_crossview_mask(B: int, V: int) -> Tensor:
    """Ring mask (self ± 1 view), the literal matrix of
    configs/ctsd/multi_datasets/ctsd_35_tirda_bm_nwao.json:328-343."""
    m = torch.zeros(V, V, dtype=torch.bool)
    for i in range(V):
        for d in (-1, 0, 1):
            m[i, (i + d) % V] = True
    return m[None].repeat(B, 1, 1)


def make_inputs(cfg: dict, B: int, T: int, V: int, H: int, W: int, seed: int = 0,
                text_len: int = 154, n_time_ids: int = 11) -> dict:
    """Synthetic inputs shaped like ``get_conditions`` output (ctsd.py:416-453) for
    the CFG-doubled batch B (pass B=2 for one guided sample)."""
    g = torch.Generator().manual_seed(seed + 1000)
    return dict(
        sample=torch.randn(B, T, V, cfg["in_channels"], H, W, generator=g),
        timestep=torch.full((B, T, V), 500.0),
        encoder_hidden_states=torch.randn(B, T, V, text_len, cfg["joint_attention_dim"], generator=g) * 0.1,
        pooled_projections=torch.randn(B, T, V, cfg["pooled_projection_dim"], generator=g) * 0.1,
        disable_crossview=torch.zeros(B, dtype=torch.bool),
        disable_temporal=torch.zeros(B, dtype=torch.bool),
        crossview_attention_mask=ring_crossview_mask(B, V),
        added_time_ids=torch.rand(B, T, V, n_time_ids, generator=g) * 2 - 1,
    )


def make_camera_inputs(B: int, T: int, V: int, seed: int = 0) -> dict:
    """synthetic `camera_intrinsics_norm` [B,T,V,3,3] (pinhole intrinsics divided by the image size, as get_conditions
    builds them) and `camera2referego` [B,T,V,4,4] (a ring of cameras yawed 360/V degrees apart with a small pitch, 1-2 m
    off the ego origin, the ego advancing a little per frame) for the explicit perspective modelling"""
    g = torch.Generator().manual_seed(seed + 4242)
    K = torch.zeros(B, T, V, 3, 3)
    K[..., 0, 0] = 0.75 + 0.1 * torch.rand(B, T, V, generator=g)
    K[..., 1, 1] = 1.30 + 0.1 * torch.rand(B, T, V, generator=g)
    K[..., 0, 2] = 0.5 + 0.02 * torch.randn(B, T, V, generator=g)
    K[..., 1, 2] = 0.5 + 0.02 * torch.randn(B, T, V, generator=g)
    K[..., 2, 2] = 1.0
    M = torch.zeros(B, T, V, 4, 4)
    for v in range(V):
        yaw = torch.full((B, T), 2 * math.pi * v / max(V, 1)) + 0.05 * torch.randn(B, T, generator=g)
        pitch = 0.03 * torch.randn(B, T, generator=g)
        cy, sy, cp, sp = torch.cos(yaw), torch.sin(yaw), torch.cos(pitch), torch.sin(pitch)
        Rz = torch.stack([torch.stack([cy, -sy, torch.zeros_like(cy)], -1), torch.stack([sy, cy, torch.zeros_like(cy)], -1),
                          torch.stack([torch.zeros_like(cy), torch.zeros_like(cy), torch.ones_like(cy)], -1)], -2)
        Ry = torch.stack([torch.stack([cp, torch.zeros_like(cp), sp], -1), torch.stack([torch.zeros_like(cp), torch.ones_like(cp), torch.zeros_like(cp)], -1),
                          torch.stack([-sp, torch.zeros_like(cp), cp], -1)], -2)
        # camera axes (x right, y down, z forward) into the ego frame (x forward, y left, z up)
        C2E = torch.tensor([[0.0, 0.0, 1.0], [-1.0, 0.0, 0.0], [0.0, -1.0, 0.0]])
        M[:, :, v, :3, :3] = Rz @ Ry @ C2E
        M[:, :, v, 0, 3] = 1.5 * torch.cos(yaw) + 0.4 * torch.arange(T).float().view(1, T)
        M[:, :, v, 1, 3] = 0.8 * torch.sin(yaw)
        M[:, :, v, 2, 3] = 1.5 + 0.05 * torch.randn(B, T, generator=g)
    M[..., 3, 3] = 1.0
    return dict(camera_intrinsics_norm=K, camera2referego=M)


def flops_per_forward(cfg: dict, B: int, T: int, V: int, H: int, W: int, text_len: int = 154) -> dict:
    """Algorithmic FLOP model of SURVEY.md Appendix C (2·MAC linear + 4·L²·d attention)."""
    d = cfg["num_attention_heads"] * cfg["attention_head_dim"]
    p = cfg["patch_size"]
    h, w = H // p, W // p
    N = h * w
    I = B * T * V
    tok, ctx = I * N, I * text_len
    nl = cfg["num_layers"]
    nd = len(cfg.get("dual_attention_layers", ()))
    ncv = len(cfg.get("crossview_block_layers") or []) if cfg.get("enable_crossview") else 0
    ntm = len(cfg.get("temporal_block_layers") or []) if cfg.get("enable_temporal") else 0
    out = {}
    out["joint_linear"] = tok * 2 * 12 * d * d * nl + ctx * 2 * 12 * d * d * (nl - 1) + ctx * 2 * 3 * d * d \
        + tok * 2 * 4 * d * d * nd + I * 2 * d * (9 * d * nd + 6 * d * (nl - nd) + 6 * d * (nl - 1) + 2 * d)
    out["joint_attn"] = I * 4 * (N + text_len) ** 2 * d * nl + I * 4 * N * N * d * nd
    out["vt_linear"] = tok * 2 * (12 + 4 + 12) * d * d * (ncv + ntm)
    cvt = cfg.get("crossview_attention_type")
    out["cv_attn"] = (B * T * h) * 4 * (V * w) ** 2 * d * ncv if cvt == "rowwise" else \
        (B * T) * 4 * (V * N) ** 2 * d * ncv
    tt = cfg.get("temporal_attention_type")
    if tt == "rowwise":
        out["t_attn"] = (B * V * h) * 4 * (T * w) ** 2 * d * ntm
    elif tt == "full":
        out["t_attn"] = (B * V) * 4 * (T * N) ** 2 * d * ntm
    else:
        out["t_attn"] = (B * V * N) * 4 * T * T * d * ntm
    cj = cfg["joint_attention_dim"]
    pd = cfg["pooled_projection_dim"]
    out["embeds"] = tok * 2 * 64 * d + ctx * 2 * cj * d + I * 2 * (256 * d + d * d + pd * d + d * d) \
        + I * 2 * (11 * 256 * d + d * d) + (ncv + ntm) * I * 2 * 8 * d * d + I * 4 * d * d + tok * 2 * 64 * d
    out["attention"] = out["joint_attn"] + out["cv_attn"] + out["t_attn"]
    out["total"] = out["joint_linear"] + out["vt_linear"] + out["attention"] + out["embeds"]
    return out
