"""TEST INFRASTRUCTURE (oracle) - CPU fp32 restatement of diffusers 0.31.0 `AutoencoderKLCogVideoX`, the temporal VAE
the reference selects with `"vae": "diffusers.AutoencoderKLCogVideoX"` (src/dwm/pipelines/ctsd.py:953-964; called
at :1206-1218 encode and :1606-1640 decode with the layout "(b v) c t h w";
examples/ctsd_35_tvae_6views_video_generation_with_layout.json:51-52 -> THUDM/CogVideoX-2b).

Only tests/ may import this file.  The arithmetic lives in a third-party dependency that is absent from
/root/reference and from this image (requirements.txt: diffusers==0.31.0), so this follows the published module
structure of that release (diffusers/models/autoencoders/autoencoder_kl_cogvideox.py) from its documented behaviour:
**parity unpinned** - no golden vectors exist in the reference, none can be generated here.

Behaviour restated (names = state-dict prefixes):
  * CogVideoXCausalConv3d: time padding = the previous call's last (kt-1) input frames (`conv_cache`), or the first
    frame repeated (kt-1) times on the first call; spatial padding = zeros; then a plain Conv3d.  The cache lives
    until `_clear_fake_context_parallel_cache()` at the end of encode()/decode(), so the frame-chunked loops below
    see a causal convolution over the whole clip - but every GroupNorm takes its statistics over ONE chunk.
  * encode(): chunks of num_sample_frames_batch_size = 8 frames (first chunk takes the remainder, e.g. 9 + 8 for
    17 frames); decode(): chunks of num_latent_frames_batch_size = 2 latent frames (3 + 2 for 5).
  * CogVideoXResnetBlock3D: norm1 -> SiLU -> conv1 -> norm2 -> SiLU -> conv2 (+ 1x1x1 conv_shortcut); norms are
    GroupNorm (encoder) or CogVideoXSpatialNorm3D (decoder: GroupNorm(f) * conv_y(zq) + conv_b(zq) with zq
    nearest-resized to f, first frame resized separately when f has an odd number (> 1) of frames).
  * CogVideoXDownsample3D: [compress_time: avg_pool1d(2, 2) over frames, keeping the first frame when the count is
    odd] -> F.pad(0, 1, 0, 1) -> Conv2d(3, stride 2) per frame.
  * CogVideoXUpsample3D: [compress_time: first frame 2-D nearest x2, the rest 3-D nearest x2 (odd > 1); 3-D nearest x2
    (even); 2-D (single frame)] or per-frame 2-D nearest x2 -> Conv2d(3, pad 1) per frame.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


def make_cogvideox_config(**over) -> dict:
    """THUDM/CogVideoX-2b vae/config.json"""
    cfg = dict(in_channels=3, out_channels=3, block_out_channels=(128, 256, 256, 512), latent_channels=16,
               layers_per_block=3, norm_eps=1e-6, norm_num_groups=32, temporal_compression_ratio=4,
               scaling_factor=1.15258426, shift_factor=None, use_quant_conv=False, use_post_quant_conv=False,
               num_latent_frames_batch_size=2, num_sample_frames_batch_size=8)
    cfg.update(over)
    return cfg


# ------------------------------------------------------------------------------------------ layers
class ConvCache(dict):
    """per-module `conv_cache` of one encode()/decode() call"""


def causal_conv3d(sd: SD, p: str, x: Tensor, cache: ConvCache) -> Tensor:
    w, b = sd[p + ".conv.weight"], sd[p + ".conv.bias"]
    kt = w.shape[2]
    if kt > 1:
        ctx = cache.get(p)
        ctx = ctx if ctx is not None else x[:, :, :1].repeat(1, 1, kt - 1, 1, 1)
        x = torch.cat([ctx, x], 2)
    cache[p] = x[:, :, x.shape[2] - kt + 1:].clone()
    ph, pw = w.shape[3] // 2, w.shape[4] // 2
    return F.conv3d(F.pad(x, (pw, pw, ph, ph)), w, b)


def _resize_like(zq: Tensor, f: Tensor) -> Tensor:
    if f.shape[2] > 1 and f.shape[2] % 2 == 1:
        z1 = F.interpolate(zq[:, :, :1], size=(1,) + tuple(f.shape[-2:]))
        z2 = F.interpolate(zq[:, :, 1:], size=(f.shape[2] - 1,) + tuple(f.shape[-2:]))
        return torch.cat([z1, z2], 2)
    return F.interpolate(zq, size=tuple(f.shape[-3:]))


def norm3d(sd: SD, p: str, f: Tensor, zq: Optional[Tensor], groups: int, eps: float, cache: ConvCache) -> Tensor:
    if zq is None:
        return F.group_norm(f, groups, sd[p + ".weight"], sd[p + ".bias"], eps)
    z = _resize_like(zq, f)
    n = F.group_norm(f, groups, sd[p + ".norm_layer.weight"], sd[p + ".norm_layer.bias"], 1e-6)
    return n * causal_conv3d(sd, p + ".conv_y", z, cache) + causal_conv3d(sd, p + ".conv_b", z, cache)


def resnet3d(sd: SD, p: str, x: Tensor, zq: Optional[Tensor], cfg: dict, cache: ConvCache) -> Tensor:
    g, eps = cfg["norm_num_groups"], cfg["norm_eps"]
    h = F.silu(norm3d(sd, p + ".norm1", x, zq, g, eps, cache))
    h = causal_conv3d(sd, p + ".conv1", h, cache)
    h = F.silu(norm3d(sd, p + ".norm2", h, zq, g, eps, cache))
    h = causal_conv3d(sd, p + ".conv2", h, cache)
    if (p + ".conv_shortcut.weight") in sd:
        x = F.conv3d(x, sd[p + ".conv_shortcut.weight"], sd[p + ".conv_shortcut.bias"])
    return x + h


def downsample3d(sd: SD, p: str, x: Tensor, compress_time: bool) -> Tensor:
    if compress_time:
        B, C, T, H, W = x.shape
        y = x.permute(0, 3, 4, 1, 2).reshape(B * H * W, C, T)
        if T % 2 == 1:
            first, rest = y[..., 0], y[..., 1:]
            if rest.shape[-1] > 0:
                rest = F.avg_pool1d(rest, 2, 2)
            y = torch.cat([first[..., None], rest], -1)
        else:
            y = F.avg_pool1d(y, 2, 2)
        x = y.reshape(B, H, W, C, y.shape[-1]).permute(0, 3, 4, 1, 2)
    x = F.pad(x, (0, 1, 0, 1))
    B, C, T, H, W = x.shape
    y = F.conv2d(x.permute(0, 2, 1, 3, 4).reshape(B * T, C, H, W), sd[p + ".conv.weight"], sd[p + ".conv.bias"], stride=2)
    return y.reshape(B, T, *y.shape[1:]).permute(0, 2, 1, 3, 4)


def upsample3d(sd: SD, p: str, x: Tensor, compress_time: bool) -> Tensor:
    if compress_time:
        T = x.shape[2]
        if T > 1 and T % 2 == 1:
            first = F.interpolate(x[:, :, 0], scale_factor=2.0)
            rest = F.interpolate(x[:, :, 1:], scale_factor=2.0)
            x = torch.cat([first[:, :, None], rest], 2)
        elif T > 1:
            x = F.interpolate(x, scale_factor=2.0)
        else:
            x = F.interpolate(x.squeeze(2), scale_factor=2.0)[:, :, None]
    else:
        B, C, T, H, W = x.shape
        y = F.interpolate(x.permute(0, 2, 1, 3, 4).reshape(B * T, C, H, W), scale_factor=2.0)
        x = y.reshape(B, T, C, *y.shape[-2:]).permute(0, 2, 1, 3, 4)
    B, C, T, H, W = x.shape
    y = F.conv2d(x.permute(0, 2, 1, 3, 4).reshape(B * T, C, H, W), sd[p + ".conv.weight"], sd[p + ".conv.bias"], padding=1)
    return y.reshape(B, T, *y.shape[1:]).permute(0, 2, 1, 3, 4)


# ------------------------------------------------------------------------------------------ encoder / decoder
def _time_levels(cfg: dict) -> int:
    return int(math.log2(cfg["temporal_compression_ratio"]))


def encoder(sd: SD, cfg: dict, x: Tensor, cache: ConvCache) -> Tensor:
    ch, L = list(cfg["block_out_channels"]), cfg["layers_per_block"]
    h = causal_conv3d(sd, "encoder.conv_in", x, cache)
    for i in range(len(ch)):
        for j in range(L):
            h = resnet3d(sd, f"encoder.down_blocks.{i}.resnets.{j}", h, None, cfg, cache)
        if i != len(ch) - 1:
            h = downsample3d(sd, f"encoder.down_blocks.{i}.downsamplers.0", h, i < _time_levels(cfg))
    for j in range(2):
        h = resnet3d(sd, f"encoder.mid_block.resnets.{j}", h, None, cfg, cache)
    h = F.silu(F.group_norm(h, cfg["norm_num_groups"], sd["encoder.norm_out.weight"], sd["encoder.norm_out.bias"], 1e-6))
    return causal_conv3d(sd, "encoder.conv_out", h, cache)


def decoder(sd: SD, cfg: dict, z: Tensor, cache: ConvCache) -> Tensor:
    ch, L = list(cfg["block_out_channels"])[::-1], cfg["layers_per_block"] + 1
    h = causal_conv3d(sd, "decoder.conv_in", z, cache)
    for j in range(2):
        h = resnet3d(sd, f"decoder.mid_block.resnets.{j}", h, z, cfg, cache)
    for i in range(len(ch)):
        for j in range(L):
            h = resnet3d(sd, f"decoder.up_blocks.{i}.resnets.{j}", h, z, cfg, cache)
        if i != len(ch) - 1:
            h = upsample3d(sd, f"decoder.up_blocks.{i}.upsamplers.0", h, i < _time_levels(cfg))
    h = F.silu(norm3d(sd, "decoder.norm_out", h, z, cfg["norm_num_groups"], 1e-6, cache))
    return causal_conv3d(sd, "decoder.conv_out", h, cache)


def _chunks(n: int, size: int):
    nb = max(n // size, 1)
    rem = n % size
    return [(size * i + (0 if i == 0 else rem), size * (i + 1) + rem) for i in range(nb)]


def encode_moments(sd: SD, cfg: dict, x: Tensor) -> Tensor:
    """AutoencoderKLCogVideoX.encode -> moments [B, 2*latent, T', h, w] (mean || logvar); x [B, 3, T, H, W]"""
    cache, out = ConvCache(), []
    for a, b in _chunks(x.shape[2], cfg["num_sample_frames_batch_size"]):
        m = encoder(sd, cfg, x[:, :, a:b], cache)
        if "quant_conv.weight" in sd:
            m = F.conv3d(m, sd["quant_conv.weight"], sd["quant_conv.bias"])
        out.append(m)
    return torch.cat(out, 2)


def decode(sd: SD, cfg: dict, z: Tensor) -> Tensor:
    """AutoencoderKLCogVideoX.decode; z [B, latent, T', h, w] -> [B, 3, T, H, W]"""
    cache, out = ConvCache(), []
    for a, b in _chunks(z.shape[2], cfg["num_latent_frames_batch_size"]):
        zi = z[:, :, a:b]
        if "post_quant_conv.weight" in sd:
            zi = F.conv3d(zi, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])
        out.append(decoder(sd, cfg, zi, cache))
    return torch.cat(out, 2)


# ------------------------------------------------------------------------------------------ synthetic weights
def param_shapes(cfg: dict) -> Dict[str, tuple]:
    ch = list(cfg["block_out_channels"])
    lc, L = cfg["latent_channels"], cfg["layers_per_block"]
    S: Dict[str, tuple] = {}

    def cconv(n, i, o, k):
        S[n + ".conv.weight"] = (o, i, k, k, k)
        S[n + ".conv.bias"] = (o,)

    def gn(n, c):
        S[n + ".weight"] = (c,)
        S[n + ".bias"] = (c,)

    def norm(n, c, zq):
        if zq is None:
            gn(n, c)
        else:
            gn(n + ".norm_layer", c)
            cconv(n + ".conv_y", zq, c, 1)
            cconv(n + ".conv_b", zq, c, 1)

    def resnet(n, i, o, zq):
        norm(n + ".norm1", i, zq)
        cconv(n + ".conv1", i, o, 3)
        norm(n + ".norm2", o, zq)
        cconv(n + ".conv2", o, o, 3)
        if i != o:
            S[n + ".conv_shortcut.weight"] = (o, i, 1, 1, 1)
            S[n + ".conv_shortcut.bias"] = (o,)

    def conv2d(n, c):
        S[n + ".conv.weight"] = (c, c, 3, 3)
        S[n + ".conv.bias"] = (c,)

    cconv("encoder.conv_in", cfg["in_channels"], ch[0], 3)
    prev = ch[0]
    for i, o in enumerate(ch):
        for j in range(L):
            resnet(f"encoder.down_blocks.{i}.resnets.{j}", prev if j == 0 else o, o, None)
        if i != len(ch) - 1:
            conv2d(f"encoder.down_blocks.{i}.downsamplers.0", o)
        prev = o
    for j in range(2):
        resnet(f"encoder.mid_block.resnets.{j}", ch[-1], ch[-1], None)
    gn("encoder.norm_out", ch[-1])
    cconv("encoder.conv_out", ch[-1], 2 * lc, 3)
    rch = ch[::-1]
    cconv("decoder.conv_in", lc, rch[0], 3)
    for j in range(2):
        resnet(f"decoder.mid_block.resnets.{j}", rch[0], rch[0], lc)
    prev = rch[0]
    for i, o in enumerate(rch):
        for j in range(L + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", prev if j == 0 else o, o, lc)
        if i != len(rch) - 1:
            conv2d(f"decoder.up_blocks.{i}.upsamplers.0", o)
        prev = o
    norm("decoder.norm_out", rch[-1], lc)
    cconv("decoder.conv_out", rch[-1], cfg["out_channels"], 3)
    if cfg.get("use_quant_conv"):
        S["quant_conv.weight"], S["quant_conv.bias"] = (2 * lc, 2 * lc, 1, 1, 1), (2 * lc,)
    if cfg.get("use_post_quant_conv"):
        S["post_quant_conv.weight"], S["post_quant_conv.bias"] = (lc, lc, 1, 1, 1), (lc,)
    return S


def make_state_dict(cfg: dict, seed: int = 0) -> SD:
    gen = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in param_shapes(cfg).items():
        if len(shape) == 1:
            v = torch.randn(*shape, generator=gen) * 0.05
            is_scale = name.endswith(".weight") or name.endswith("conv_y.conv.bias")     # gains around 1
            sd[name] = 1.0 + v if is_scale else v
        else:
            fan_in = 1
            for s_ in shape[1:]:
                fan_in *= s_
            std = fan_in ** -0.5
            if ".conv2." in name:
                std *= 0.5
            if ".conv_y." in name or ".conv_b." in name:
                std *= 0.3
            sd[name] = torch.randn(*shape, generator=gen) * std
    return sd
