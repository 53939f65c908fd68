"""CPU oracle for the CTSD SD-2.1 UNet denoiser.

TEST INFRASTRUCTURE ONLY (same rules as ctsd_oracle.py).  PARITY: the reference ships no tests / golden tensors
for this path and diffusers 0.31.0 cannot be imported here, so the diffusers leaf arithmetic below is UNPINNED; the
composition the reference owns IS pinned: tests/golden/make_reference_unet_fixture.py executes the real
UNetCrossviewTemporalConditionModel.forward with the real block / ResBlock / TransformerModel /
TemporalBasicTransformerBlock classes over this file's leaf functions and reproduces `unet_forward` to 4e-6
(tests/test_reference_fixtures_cpu.py).

Plain-PyTorch fp32 restatement of
  * src/dwm/models/crossview_temporal_unet.py:648-835  UNetCrossviewTemporalConditionModel.forward
    and the block classes at :10-352 (mid / down / cross-attn down / up / cross-attn up);
  * src/dwm/models/crossview_temporal.py:75-164   ResBlock  (ResnetBlock2D + TemporalResnetBlock, AlphaBlender)
  * src/dwm/models/crossview_temporal.py:167-266  TemporalBasicTransformerBlock
  * src/dwm/models/crossview_temporal.py:269-514  TransformerModel
  * the diffusers==0.31.0 modules those instantiate: ResnetBlock2D,
    TemporalResnetBlock, BasicTransformerBlock (AttnProcessor2_0), Downsample2D / Upsample2D,
    Timesteps / TimestepEmbedding, and the UNetSpatioTemporalConditionModel stem / head.
State-dict keys follow the reference module tree, so one set of weights loads into this oracle, the
HIP model (opendwm_amd/unet.py) and the reference class.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from .ctsd_oracle import (SD, Tensor, _heads, _unheads, alpha_blender, feed_forward, linear, rearr, ring_crossview_mask,
                          sdpa, timestep_embedding_mlp, timesteps_sinusoid)


def make_unet_config(**over) -> dict:
    """examples/ctsd_21_6views_video_generation.json model block (reference repo)"""
    cfg = dict(
        in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280), addition_time_embed_dim=256,
        projection_class_embeddings_input_dim=2816, layers_per_block=2, norm_eps=1e-5, cross_attention_dim=1024,
        transformer_layers_per_block=1, num_attention_heads=(5, 10, 20, 20), merge_factor=2,
        down_block_types=("CrossAttnDownBlockCrossviewTemporal",) * 3 + ("DownBlockCrossviewTemporal",),
        up_block_types=("UpBlockCrossviewTemporal",) + ("CrossAttnUpBlockCrossviewTemporal",) * 3,
        enable_crossview=True, enable_temporal=True, enable_rowwise_crossview=True, enable_rowwise_temporal=True)
    cfg.update(over)
    return cfg


# ------------------------------------------------------------------------------------------ diffusers pieces
def conv2d(sd: SD, p: str, x: Tensor, stride: int = 1, padding: int = 1) -> Tensor:
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride=stride, padding=padding)


def resnet_block_2d(sd: SD, p: str, x: Tensor, temb: Tensor, eps: float) -> Tensor:
    """diffusers ResnetBlock2D(groups=32, time_embedding_norm='default', output_scale_factor=1); x [N,C,H,W], temb [N,E]"""
    h = F.silu(F.group_norm(x, 32, sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], eps))
    h = conv2d(sd, p + ".conv1", h)
    h = h + linear(sd, p + ".time_emb_proj", F.silu(temb))[:, :, None, None]
    h = F.silu(F.group_norm(h, 32, sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], eps))
    h = conv2d(sd, p + ".conv2", h)
    if (p + ".conv_shortcut.weight") in sd:
        x = conv2d(sd, p + ".conv_shortcut", x, padding=0)
    return x + h


def temporal_resnet_block(sd: SD, p: str, x: Tensor, temb: Tensor, eps: float) -> Tensor:
    """diffusers TemporalResnetBlock: Conv3d kernel (3,1,1) pad (1,0,0); x [N,C,T,H,W], temb [N,T,E]"""
    h = F.silu(F.group_norm(x, 32, sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], eps))
    h = F.conv3d(h, sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], padding=(1, 0, 0))
    t = linear(sd, p + ".time_emb_proj", F.silu(temb))[:, :, :, None, None].permute(0, 2, 1, 3, 4)
    h = h + t
    h = F.silu(F.group_norm(h, 32, sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], eps))
    h = F.conv3d(h, sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], padding=(1, 0, 0))
    return x + h


def res_block(sd: SD, p: str, x: Tensor, temb: Tensor, disable_temporal: Tensor, eps: float) -> Tensor:
    """ResBlock.forward (crossview_temporal.py:120-164); x [B,T,V,C,H,W], temb [B,T,V,E]"""
    B = x.shape[0]
    s = resnet_block_2d(sd, p + ".spatial_res_block", x.flatten(0, 2), temb.flatten(0, 2), eps).unflatten(0, x.shape[:3])
    if (p + ".temporal_res_block.conv1.weight") not in sd:
        return s
    t = temporal_resnet_block(sd, p + ".temporal_res_block", s.permute(0, 2, 3, 1, 4, 5).flatten(0, 1),
                              temb.transpose(1, 2).flatten(0, 1), eps)
    t = t.unflatten(0, (B, -1)).permute(0, 3, 1, 2, 4, 5)
    return alpha_blender(sd, p + ".time_mixer", s, t, disable_temporal)


def _attention(sd: SD, p: str, heads: int, x: Tensor, ctx: Optional[Tensor] = None, mask: Optional[Tensor] = None) -> Tensor:
    """diffusers Attention + AttnProcessor2_0 (bias=False on q/k/v, out bias)"""
    kv = x if ctx is None else ctx
    q = _heads(linear(sd, p + ".to_q", x), heads)
    k = _heads(linear(sd, p + ".to_k", kv), heads)
    v = _heads(linear(sd, p + ".to_v", kv), heads)
    m = None if mask is None else mask[:, None]
    return linear(sd, p + ".to_out.0", _unheads(sdpa(q, k, v, m)))


def basic_transformer_block(sd: SD, p: str, heads: int, x: Tensor, ctx: Tensor) -> Tensor:
    """diffusers BasicTransformerBlock(dim, heads, head_dim, cross_attention_dim): layer_norm, GEGLU"""
    d = x.shape[-1]
    x = x + _attention(sd, p + ".attn1", heads, F.layer_norm(x, (d,), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], 1e-5))
    x = x + _attention(sd, p + ".attn2", heads, F.layer_norm(x, (d,), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], 1e-5), ctx)
    return x + feed_forward(sd, p + ".ff", F.layer_norm(x, (d,), sd[p + ".norm3.weight"], sd[p + ".norm3.bias"], 1e-5), "geglu")


def temporal_basic_transformer_block(sd: SD, p: str, heads: int, x: Tensor, num_frames: int,
                                     mask: Optional[Tensor] = None) -> Tensor:
    """TemporalBasicTransformerBlock.forward (crossview_temporal.py:217-266), cross_attention_dim=None, is_res=True;
    x [(Bf F), S, C]: attention runs over F for each of the S positions."""
    bf, S, d = x.shape
    b = bf // num_frames
    x = x.unflatten(0, (b, -1)).transpose(1, 2).flatten(0, 1)                    # [(b S), F, C]
    x = feed_forward(sd, p + ".ff_in", F.layer_norm(x, (d,), sd[p + ".norm_in.weight"], sd[p + ".norm_in.bias"], 1e-5), "geglu") + x
    if mask is not None:
        mask = mask.repeat_interleave(S, 0)
    x = _attention(sd, p + ".attn1", heads, F.layer_norm(x, (d,), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], 1e-5),
                   mask=mask) + x
    x = feed_forward(sd, p + ".ff", F.layer_norm(x, (d,), sd[p + ".norm3.weight"], sd[p + ".norm3.bias"], 1e-5), "geglu") + x
    return x.unflatten(0, (b, -1)).transpose(1, 2).flatten(0, 1)


def transformer_model(sd: SD, p: str, cfg: dict, heads: int, x: Tensor, ehs: Tensor, disable_crossview: Tensor,
                      disable_temporal: Tensor, crossview_attention_mask: Optional[Tensor], n_layers: int) -> Tensor:
    """TransformerModel.forward (crossview_temporal.py:398-514); x [B,T,V,C,H,W], ehs [B,T,V,L,Cc]"""
    B, T, V, C, H, W = x.shape
    residual = x
    ctx = ehs.flatten(0, 2)
    h = F.group_norm(x.flatten(0, 2), 32, sd[p + ".norm.weight"], sd[p + ".norm.bias"], 1e-6)
    h = linear(sd, p + ".proj_in", h.flatten(2).transpose(-2, -1))                # [(BTV), HW, C]
    has_cv = (p + ".view_pos_embed.linear_1.weight") in sd
    has_t = (p + ".time_pos_embed.linear_1.weight") in sd
    if has_cv:
        idx = torch.arange(V, device=x.device).view(1, 1, V).repeat(B, T, 1)
        view_emb = timestep_embedding_mlp(sd, p + ".view_pos_embed", timesteps_sinusoid(idx.flatten(), C)).unsqueeze(1)
    if has_t:
        idx = torch.arange(T, device=x.device).view(1, T, 1).repeat(B, 1, V)
        seq_emb = timestep_embedding_mlp(sd, p + ".time_pos_embed", timesteps_sinusoid(idx.flatten(), C)).unsqueeze(1)
    mask = crossview_attention_mask
    if cfg["enable_rowwise_crossview"] and mask is not None:
        mask = mask.repeat_interleave(W, 2).repeat_interleave(W, 1).repeat_interleave(T, 0)
    for l in range(n_layers):
        h = basic_transformer_block(sd, f"{p}.transformer_blocks.{l}", heads, h, ctx)
        if has_cv:
            c = h + view_emb
            q = f"{p}.crossview_transformer_blocks.{l}"
            if cfg["enable_rowwise_crossview"]:
                c = rearr(c, "btv (h w) c -> (btv w) h c", w=W)
                c = temporal_basic_transformer_block(sd, q, heads, c, V * W, mask)
                c = rearr(c, "(btv w) h c -> btv (h w) c", w=W)
            else:
                c = temporal_basic_transformer_block(sd, q, heads, c, V, mask)
            h = alpha_blender(sd, p + ".view_mixer", h.unflatten(0, (B, -1)), c.unflatten(0, (B, -1)), disable_crossview).flatten(0, 1)
        if has_t:
            c = h + seq_emb
            q = f"{p}.temporal_transformer_blocks.{l}"
            if cfg["enable_rowwise_temporal"]:
                c = rearr(c, "(b t v) (h w) c -> (b v t w) h c", b=B, t=T, w=W)
                c = temporal_basic_transformer_block(sd, q, heads, c, T * W)
                c = rearr(c, "(b v t w) h c -> (b t v) (h w) c", b=B, t=T, w=W)
            else:
                c = rearr(c, "(b t v) hw c -> (b v t) hw c", b=B, t=T)
                c = temporal_basic_transformer_block(sd, q, heads, c, T)
                c = rearr(c, "(b v t) hw c -> (b t v) hw c", b=B, t=T)
            h = alpha_blender(sd, p + ".time_mixer", h.unflatten(0, (B, -1)), c.unflatten(0, (B, -1)), disable_temporal).flatten(0, 1)
    h = linear(sd, p + ".proj_out", h)
    return h.transpose(-2, -1).reshape(B, T, V, C, H, W) + residual


# ------------------------------------------------------------------------------------------ the UNet
def _block_plan(cfg: dict):
    """channel bookkeeping of UNetCrossviewTemporalConditionModel.__init__ (crossview_temporal_unet.py:438-560)"""
    boc = list(cfg["block_out_channels"])
    n = len(boc)
    heads = cfg["num_attention_heads"]
    heads = [heads] * n if isinstance(heads, int) else list(heads)
    lpb = cfg["layers_per_block"]
    lpb = [lpb] * n if isinstance(lpb, int) else list(lpb)
    tl = cfg["transformer_layers_per_block"]
    tl = [tl] * n if isinstance(tl, int) else list(tl)
    down = []
    out_c = boc[0]
    for i, typ in enumerate(cfg["down_block_types"]):
        in_c, out_c = out_c, boc[i]
        down.append(dict(attn=typ.startswith("CrossAttn"), resnets=[(in_c if j == 0 else out_c, out_c) for j in range(lpb[i])],
                         heads=heads[i], tlayers=tl[i], downsample=i != n - 1, channels=out_c))
    up = []
    rboc, rheads, rlpb, rtl = boc[::-1], heads[::-1], lpb[::-1], tl[::-1]
    out_c = rboc[0]
    for i, typ in enumerate(cfg["up_block_types"]):
        prev, out_c = out_c, rboc[i]
        in_c = rboc[min(i + 1, n - 1)]
        nl = rlpb[i] + 1
        res = []
        for j in range(nl):
            skip = in_c if j == nl - 1 else out_c
            rin = prev if j == 0 else out_c
            res.append((rin + skip, out_c))
        up.append(dict(attn=typ.startswith("CrossAttn"), resnets=res, heads=rheads[i], tlayers=rtl[i], upsample=i != n - 1,
                       channels=out_c))
    return down, dict(channels=boc[-1], heads=heads[-1], tlayers=tl[-1]), up


def unet_forward(sd: SD, cfg: dict, sample: Tensor, timesteps: Tensor, encoder_hidden_states: Tensor,
                 added_time_ids: Optional[Tensor] = None, disable_crossview: Optional[Tensor] = None,
                 disable_temporal: Optional[Tensor] = None, crossview_attention_mask: Optional[Tensor] = None,
                 condition_image_tensor: Optional[Tensor] = None) -> Tensor:
    """UNetCrossviewTemporalConditionModel.forward (crossview_temporal_unet.py:648-835) without depth net;
    sample [B,T,V,C,H,W] -> noise prediction of the same shape.  With cfg["condition_image_adapter_config"] and a
    condition_image_tensor the layout ImageAdapter residuals are added after conv_in and after every down block
    (:717-755; the last skip of the block is the sum)."""
    B, T, V, _, H, W = sample.shape
    eps = cfg["norm_eps"]
    down, mid, up = _block_plan(cfg)
    c0 = cfg["block_out_channels"][0]
    emb = timestep_embedding_mlp(sd, "time_embedding", timesteps_sinusoid(timesteps.flatten(), c0)).unflatten(0, (B, T, V))
    if added_time_ids is not None:
        aug = timesteps_sinusoid(added_time_ids.flatten(), cfg["addition_time_embed_dim"]).view(B * T * V, -1)
        emb = emb + timestep_embedding_mlp(sd, "add_embedding", aug).view(B, T, V, -1)
    if disable_crossview is None:
        disable_crossview = torch.zeros(B, dtype=torch.bool, device=sample.device)
    if disable_temporal is None:
        disable_temporal = torch.zeros(B, dtype=torch.bool, device=sample.device)

    residuals = []
    if cfg.get("condition_image_adapter_config") is not None and condition_image_tensor is not None:
        from .ctsd_oracle import image_adapter
        residuals = list(image_adapter(sd, cfg, condition_image_tensor))
    x = conv2d(sd, "conv_in", sample.flatten(0, 2)).unflatten(0, (B, T, V))
    if residuals:
        x = x + residuals.pop(0)
    skips = [x]

    def tm(p, heads, nl, h):
        return transformer_model(sd, p, cfg, heads, h, encoder_hidden_states, disable_crossview, disable_temporal,
                                 crossview_attention_mask, nl)

    for i, blk in enumerate(down):
        for j in range(len(blk["resnets"])):
            x = res_block(sd, f"down_blocks.{i}.resnets.{j}", x, emb, disable_temporal, eps)
            if blk["attn"]:
                x = tm(f"down_blocks.{i}.attentions.{j}", blk["heads"], blk["tlayers"], x)
            skips.append(x)
        if blk["downsample"]:
            x = conv2d(sd, f"down_blocks.{i}.downsamplers.0.conv", x.flatten(0, 2), stride=2).unflatten(0, (B, T, V))
            skips.append(x)
        if residuals:
            x = x + residuals.pop(0)
            skips[-1] = x
    x = res_block(sd, "mid_block.resnets.0", x, emb, disable_temporal, eps)
    x = tm("mid_block.attentions.0", mid["heads"], mid["tlayers"], x)
    x = res_block(sd, "mid_block.resnets.1", x, emb, disable_temporal, eps)
    for i, blk in enumerate(up):
        for j in range(len(blk["resnets"])):
            x = torch.cat([x, skips.pop()], dim=-3)
            x = res_block(sd, f"up_blocks.{i}.resnets.{j}", x, emb, disable_temporal, eps)
            if blk["attn"]:
                x = tm(f"up_blocks.{i}.attentions.{j}", blk["heads"], blk["tlayers"], x)
        if blk["upsample"]:
            y = F.interpolate(x.flatten(0, 2), scale_factor=2.0, mode="nearest")
            x = conv2d(sd, f"up_blocks.{i}.upsamplers.0.conv", y).unflatten(0, (B, T, V))
    y = F.silu(F.group_norm(x.flatten(0, 2), 32, sd["conv_norm_out.weight"], sd["conv_norm_out.bias"], 1e-5))
    return conv2d(sd, "conv_out", y).unflatten(0, (B, T, V))


# ------------------------------------------------------------------------------------------ parameters / inputs
def unet_param_shapes(cfg: dict) -> Dict[str, tuple]:
    down, mid, up = _block_plan(cfg)
    c0 = cfg["block_out_channels"][0]
    E = 4 * c0
    cd = cfg["cross_attention_dim"]
    S: Dict[str, tuple] = {}

    def lin(p, i, o, bias=True):
        S[p + ".weight"] = (o, i)
        if bias:
            S[p + ".bias"] = (o,)

    def norm(p, c):
        S[p + ".weight"] = (c,)
        S[p + ".bias"] = (c,)

    def conv(p, i, o, k=3):
        S[p + ".weight"] = (o, i, k, k)
        S[p + ".bias"] = (o,)

    def resblock(p, i, o):
        q = p + ".spatial_res_block"
        norm(q + ".norm1", i); conv(q + ".conv1", i, o); lin(q + ".time_emb_proj", E, o)
        norm(q + ".norm2", o); conv(q + ".conv2", o, o)
        if i != o:
            conv(q + ".conv_shortcut", i, o, 1)
        if cfg["enable_temporal"]:
            q = p + ".temporal_res_block"
            norm(q + ".norm1", o); S[q + ".conv1.weight"] = (o, o, 3, 1, 1); S[q + ".conv1.bias"] = (o,)
            lin(q + ".time_emb_proj", E, o)
            norm(q + ".norm2", o); S[q + ".conv2.weight"] = (o, o, 3, 1, 1); S[q + ".conv2.bias"] = (o,)
            S[p + ".time_mixer.mix_factor"] = (1,)

    def attn(p, d, kvdim=None):
        lin(p + ".to_q", d, d, False); lin(p + ".to_k", kvdim or d, d, False); lin(p + ".to_v", kvdim or d, d, False)
        lin(p + ".to_out.0", d, d)

    def ff(p, d):
        lin(p + ".net.0.proj", d, 8 * d); lin(p + ".net.2", 4 * d, d)

    def tbt(p, d):
        norm(p + ".norm_in", d); ff(p + ".ff_in", d); norm(p + ".norm1", d); attn(p + ".attn1", d)
        norm(p + ".norm3", d); ff(p + ".ff", d)

    def tmodel(p, c, nl):
        norm(p + ".norm", c); lin(p + ".proj_in", c, c)
        for l in range(nl):
            q = f"{p}.transformer_blocks.{l}"
            norm(q + ".norm1", c); attn(q + ".attn1", c); norm(q + ".norm2", c); attn(q + ".attn2", c, cd)
            norm(q + ".norm3", c); ff(q + ".ff", c)
        if cfg["enable_crossview"]:
            lin(p + ".view_pos_embed.linear_1", c, 4 * c); lin(p + ".view_pos_embed.linear_2", 4 * c, c)
            for l in range(nl):
                tbt(f"{p}.crossview_transformer_blocks.{l}", c)
            S[p + ".view_mixer.mix_factor"] = (1,)
        if cfg["enable_temporal"]:
            lin(p + ".time_pos_embed.linear_1", c, 4 * c); lin(p + ".time_pos_embed.linear_2", 4 * c, c)
            for l in range(nl):
                tbt(f"{p}.temporal_transformer_blocks.{l}", c)
            S[p + ".time_mixer.mix_factor"] = (1,)
        lin(p + ".proj_out", c, c)

    conv("conv_in", cfg["in_channels"], c0)
    lin("time_embedding.linear_1", c0, E); lin("time_embedding.linear_2", E, E)
    if cfg.get("projection_class_embeddings_input_dim") is not None:
        lin("add_embedding.linear_1", cfg["projection_class_embeddings_input_dim"], E); lin("add_embedding.linear_2", E, E)
    for i, blk in enumerate(down):
        for j, (ci, co) in enumerate(blk["resnets"]):
            resblock(f"down_blocks.{i}.resnets.{j}", ci, co)
            if blk["attn"]:
                tmodel(f"down_blocks.{i}.attentions.{j}", co, blk["tlayers"])
        if blk["downsample"]:
            conv(f"down_blocks.{i}.downsamplers.0.conv", blk["channels"], blk["channels"])
    resblock("mid_block.resnets.0", mid["channels"], mid["channels"])
    tmodel("mid_block.attentions.0", mid["channels"], mid["tlayers"])
    resblock("mid_block.resnets.1", mid["channels"], mid["channels"])
    for i, blk in enumerate(up):
        for j, (ci, co) in enumerate(blk["resnets"]):
            resblock(f"up_blocks.{i}.resnets.{j}", ci, co)
            if blk["attn"]:
                tmodel(f"up_blocks.{i}.attentions.{j}", co, blk["tlayers"])
        if blk["upsample"]:
            conv(f"up_blocks.{i}.upsamplers.0.conv", blk["channels"], blk["channels"])
    norm("conv_norm_out", c0)
    conv("conv_out", c0, cfg["out_channels"])
    ac = cfg.get("condition_image_adapter_config")
    if ac is not None:                                   # dwm.models.adapters.ImageAdapter (adapters.py:6-38)
        cin = ac.get("in_channels", 3) * ac.get("downscale_factor", 8) ** 2
        for i, ch in enumerate(ac["channels"]):
            b = f"condition_image_adapter.body.{i}"
            prev = cin if i == 0 else ac["channels"][i - 1]
            if prev != ch:
                S[b + ".in_conv.weight"], S[b + ".in_conv.bias"] = (ch, prev, 1, 1), (ch,)
            for j in range(ac.get("num_res_blocks", 2)):
                S[f"{b}.resnets.{j}.block1.weight"], S[f"{b}.resnets.{j}.block1.bias"] = (ch, ch, 3, 3), (ch,)
                S[f"{b}.resnets.{j}.block2.weight"], S[f"{b}.resnets.{j}.block2.bias"] = (ch, ch, 1, 1), (ch,)
            if ac.get("use_zero_convs", False):
                S[f"condition_image_adapter.zero_convs.{i}.weight"] = (ch, ch, 1, 1)
                S[f"condition_image_adapter.zero_convs.{i}.bias"] = (ch,)
    return S


def make_unet_state_dict(cfg: dict, seed: int = 0) -> SD:
    """Seeded synthetic weights: matrices / conv kernels ~ N(0, 1/fan_in) (second conv of every resnet and the
    transformer proj_out damped so the residual streams stay O(1)), norm weights 1 + noise, small biases,
    mix_factor = merge_factor."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in unet_param_shapes(cfg).items():
        if name.endswith("mix_factor"):
            sd[name] = torch.full(shape, float(cfg.get("merge_factor", 2)))
        elif len(shape) == 1:
            v = torch.randn(*shape, generator=g) * 0.05
            sd[name] = 1.0 + v if (name.endswith(".weight")) else v
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            std = fan_in ** -0.5
            if ".conv2." in name or name.endswith("proj_out.weight") or ".net.2." in name or ".to_out.0." in name:
                std *= 0.5
            if name.startswith("condition_image_adapter.") and (".block2." in name or ".zero_convs." in name):
                std *= 0.5
            sd[name] = torch.randn(*shape, generator=g) * std
    return sd


def make_unet_inputs(cfg: dict, B: int, T: int, V: int, H: int, W: int, seed: int = 0, text_len: int = 77,
                     n_time_ids: int = 11) -> dict:
    g = torch.Generator().manual_seed(seed + 2000)
    return dict(
        sample=torch.randn(B, T, V, cfg["in_channels"], H, W, generator=g),
        timesteps=torch.full((B, T, V), 500.0),
        encoder_hidden_states=torch.randn(B, T, V, text_len, cfg["cross_attention_dim"], generator=g) * 0.5,
        added_time_ids=torch.rand(B, T, V, n_time_ids, generator=g) * 2 - 1,
        disable_crossview=torch.zeros(B, dtype=torch.bool),
        disable_temporal=torch.zeros(B, dtype=torch.bool),
        crossview_attention_mask=ring_crossview_mask(B, V),
    )


def unet_flops(cfg: dict, B: int, T: int, V: int, H: int, W: int, text_len: int = 77) -> float:
    """2*MAC of every conv / linear + 4*L^2*64 per attention problem-head of one forward"""
    down, mid, up = _block_plan(cfg)
    I = B * T * V
    E = 4 * cfg["block_out_channels"][0]
    cd = cfg["cross_attention_dim"]
    total = 0.0

    def resblock(ci, co, h, w):
        px = I * h * w
        f = 2.0 * px * co * (9 * ci) + 2.0 * px * co * (9 * co) + (2.0 * px * co * ci if ci != co else 0) + 2.0 * I * E * co
        if cfg["enable_temporal"]:
            f += 2 * (2.0 * px * co * 3 * co) + 2.0 * I * E * co
        return f

    def tbt(c, px, L, nprob):
        return 2.0 * px * c * (8 * c + 4 * c) * 2 + 2.0 * px * c * 4 * c + 4.0 * nprob * (c // 64) * L * L * 64

    def tmodel(c, nl, h, w):
        px, N = I * h * w, h * w
        f = 2 * 2.0 * px * c * c
        for _ in range(nl):
            f += 2.0 * px * c * 4 * c + 4.0 * I * (c // 64) * N * N * 64                      # self attention
            f += 2.0 * px * c * 2 * c + 2 * 2.0 * I * text_len * cd * c + 4.0 * I * (c // 64) * N * text_len * 64
            f += 2.0 * px * c * 12 * c
            if cfg["enable_crossview"]:
                L = V * w if cfg["enable_rowwise_crossview"] else V
                f += tbt(c, px, L, px // L)
            if cfg["enable_temporal"]:
                L = T * w if cfg["enable_rowwise_temporal"] else T
                f += tbt(c, px, L, px // L)
        return f

    h, w = H, W
    total += 2.0 * I * h * w * cfg["block_out_channels"][0] * 9 * cfg["in_channels"]
    for blk in down:
        for (ci, co) in blk["resnets"]:
            total += resblock(ci, co, h, w)
            if blk["attn"]:
                total += tmodel(co, blk["tlayers"], h, w)
        if blk["downsample"]:
            h, w = h // 2, w // 2
            total += 2.0 * I * h * w * blk["channels"] * 9 * blk["channels"]
    total += 2 * resblock(mid["channels"], mid["channels"], h, w) + tmodel(mid["channels"], mid["tlayers"], h, w)
    for blk in up:
        for (ci, co) in blk["resnets"]:
            total += resblock(ci, co, h, w)
            if blk["attn"]:
                total += tmodel(co, blk["tlayers"], h, w)
        if blk["upsample"]:
            h, w = 2 * h, 2 * w
            total += 2.0 * I * h * w * blk["channels"] * 9 * blk["channels"]
    total += 2.0 * I * h * w * cfg["out_channels"] * 9 * cfg["block_out_channels"][0]
    return total


# ------------------------------------------------------------------------------------------ scheduler / denoise loop
def dpm_solver_tables(num_inference_steps: int, num_train_timesteps: int = 1000, beta_start: float = 0.00085,
                      beta_end: float = 0.012):
    """diffusers DPMSolverMultistepScheduler.set_timesteps (0.31.0 defaults of the SD 2.1 scheduler config:
    scaled_linear betas, timestep_spacing 'linspace', final_sigmas_type 'zero'): returns (timesteps [n], sigmas [n+1])."""
    import numpy as np
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
    acp = torch.cumprod(1.0 - betas, 0).numpy().astype(np.float64)
    ts = np.linspace(0, num_train_timesteps - 1, num_inference_steps + 1).round()[::-1][:-1].copy().astype(np.int64)
    sig = ((1 - acp) / acp) ** 0.5
    sig = np.interp(ts, np.arange(num_train_timesteps), sig)
    sig = np.concatenate([sig, [0.0]])
    return torch.from_numpy(ts), torch.from_numpy(sig)


def dpm_solver_coefficients(sigmas: Tensor, i: int, prediction_type: str):
    """per-step scalars of dpmsolver++ (solver_order 2, midpoint, lower_order_final): x0 = kx*x + ko*out;
    x' = A*x + B*x0 + C*x0_prev (DPMSolverMultistepScheduler.convert_model_output / dpm_solver_first_order_update /
    multistep_dpm_solver_second_order_update)."""
    n = sigmas.numel() - 1
    def a_s(s):
        a = 1.0 / math.sqrt(s * s + 1.0)
        return a, s * a
    alpha_s0, sigma_s0 = a_s(float(sigmas[i]))
    kx, ko = (1.0 / alpha_s0, -sigma_s0 / alpha_s0) if prediction_type == "epsilon" else (alpha_s0, -sigma_s0)
    alpha_t, sigma_t = a_s(float(sigmas[i + 1]))
    lam = lambda a, s: math.log(a) - math.log(s)
    if i == n - 1:                                   # final_sigmas_type "zero": x' = x0
        return kx, ko, 0.0, 1.0, 0.0
    h = lam(alpha_t, sigma_t) - lam(alpha_s0, sigma_s0)
    e = math.exp(-h) - 1.0
    A = sigma_t / sigma_s0
    if i == 0:                                       # first step: first order
        return kx, ko, A, -alpha_t * e, 0.0
    alpha_s1, sigma_s1 = a_s(float(sigmas[i - 1]))
    r0 = (lam(alpha_s0, sigma_s0) - lam(alpha_s1, sigma_s1)) / h
    # D0 = m0, D1 = (m0 - m1) / r0:  x' = A x - alpha_t e m0 - 0.5 alpha_t e (m0 - m1) / r0
    return kx, ko, A, -alpha_t * e * (1.0 + 0.5 / r0), 0.5 * alpha_t * e / r0


def unet_denoise(sd: SD, cfg: dict, latents: Tensor, conditions: dict, steps: int, guidance_scale: float,
                 prediction_type: str = "v_prediction", stop: Optional[int] = None) -> Tensor:
    """inference_pipeline hot loop (ctsd.py:1496-1575) for the SD 2.1 configs: CFG + DPMSolverMultistepScheduler.
    conditions: the CFG-doubled tensors ([2B, ...], unconditional first); latents fp32 [B,T,V,C,H,W]."""
    ts, sig = dpm_solver_tables(steps)
    x = latents.float().clone() * 1.0            # init_noise_sigma = 1 for DPM-Solver
    prev = torch.zeros_like(x)
    B, T, V = x.shape[:3]
    for i in range(steps if stop is None else stop):
        t = ts[i].float().expand(2 * B, T, V)
        out = unet_forward(sd, cfg, torch.cat([x, x]), t, **conditions)
        u, c = out.chunk(2)
        o = u + guidance_scale * (c - u)
        kx, ko, A, Bc, Cc = dpm_solver_coefficients(sig, i, prediction_type)
        x0 = kx * x + ko * o
        x = A * x + Bc * x0 + Cc * prev
        prev = x0
    return x
