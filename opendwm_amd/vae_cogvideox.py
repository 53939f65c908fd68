"""MI355X-native AutoencoderKLCogVideoX: the temporal VAE (4x frames, 8x8 pixels -> 16 latent channels) the CTSD
pipeline builds with `"vae": "diffusers.AutoencoderKLCogVideoX"` (src/dwm/pipelines/ctsd.py:953-964; encode at
:1206-1218, decode at :1606-1640, layout "(b v) c t h w"; examples/ctsd_35_tvae_6views_video_generation_with_layout.json).

Module tree / state-dict keys follow diffusers 0.31.0 (encoder.conv_in.conv, encoder.down_blocks.N.resnets.M.{norm1,
conv1.conv,norm2,conv2.conv,conv_shortcut}, ...downsamplers.0.conv, decoder...norm1.{norm_layer,conv_y.conv,conv_b.conv},
...upsamplers.0.conv, decoder.norm_out, decoder.conv_out.conv), so `vae/diffusion_pytorch_model.safetensors` of
THUDM/CogVideoX-2b loads with strict=True.

Data layout: activations are token-major bf16 rows ordered (frame t, video b, y, x) - one frame of every video is a
contiguous slab - so that
  * a causal 3x3x3 convolution is ONE implicit GEMM of dwm_gemm_bf16 with 27 taps over `ops.Grid3D` (two context frames in
    front of the chunk: the previous chunk's last two input frames = diffusers' `conv_cache`, or the first frame twice);
  * GroupNorm over (channels/group, t, y, x) of one video is the row-mapped GroupNorm kernel, whose output lands directly
    in the interior of the next convolution's padded grid; the decoder's CogVideoXSpatialNorm3D gathers
    conv_y(zq) / conv_b(zq) - evaluated once at latent resolution by a GEMM - through the nearest-resize index
    (dwm_groupnorm_spatial);
  * temporal average pooling / nearest upsampling are frame-slab mixes (dwm_frame_mix_bf16); the per-frame stride-2 and
    nearest-2x + 3x3 convolutions are the 2-D VAE's implicit GEMMs with (t, b) as the image index.
Frames are processed in the reference implementation's chunks (8 frames encoding, 2 latent frames decoding, remainder in
the first chunk): the GroupNorm statistics are per chunk there, so the chunking is part of the function being computed.
"""
from __future__ import annotations

import math
import types
from typing import Dict, List, Optional, Tuple

import torch
from torch import nn

from . import ops
from .blocks import STORE, _bf
from .ops import EPI_RESID, Grid3D, PaddedGrid
from .vae import AutoencoderKL, DiagonalGaussianDistribution, _conv3_w

bf16 = torch.bfloat16


class _CausalConv3d(nn.Module):
    """diffusers CogVideoXCausalConv3d: holds `.conv` (Conv3d without padding)"""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int):
        super().__init__()
        self.conv = nn.Conv3d(in_channels, out_channels, kernel_size)


class _SpatialNorm3D(nn.Module):
    """diffusers CogVideoXSpatialNorm3D(f_channels, zq_channels, groups)"""

    def __init__(self, f_channels: int, zq_channels: int, groups: int):
        super().__init__()
        self.norm_layer = nn.GroupNorm(groups, f_channels, eps=1e-6)
        self.conv_y = _CausalConv3d(zq_channels, f_channels, 1)
        self.conv_b = _CausalConv3d(zq_channels, f_channels, 1)


class _Ctx:
    """state of one encode() / decode() call: geometry, the causal-convolution caches, the latent rows of the chunk"""

    def __init__(self, owner: "AutoencoderKLCogVideoX", videos: int, device):
        self.owner, self.B, self.dev = owner, videos, device
        self.cache: Dict[int, torch.Tensor] = {}
        self.zrows: Optional[torch.Tensor] = None
        self.Tz = self.hz = self.wz = 0

    def zt(self, T: int) -> List[int]:
        """frame of the latent chunk that f frame t reads (F.interpolate nearest; the first frame of an odd clip is
        resized separately, CogVideoXSpatialNorm3D.forward)"""
        Tz = self.Tz
        if T > 1 and T % 2 == 1:
            return [0] + [1 + ((t - 1) * (Tz - 1)) // (T - 1) for t in range(1, T)]
        return [(t * Tz) // T for t in range(T)]


class _Resnet3D(nn.Module):
    """diffusers CogVideoXResnetBlock3D (temb_channels = 0, conv_shortcut = 1x1x1 CogVideoXSafeConv3d)"""

    def __init__(self, in_channels: int, out_channels: int, groups: int, eps: float, zq_channels: Optional[int]):
        super().__init__()
        self.groups, self.eps = groups, eps
        mk = (lambda c: nn.GroupNorm(groups, c, eps=eps)) if zq_channels is None else (lambda c: _SpatialNorm3D(c, zq_channels, groups))
        self.norm1 = mk(in_channels)
        self.conv1 = _CausalConv3d(in_channels, out_channels, 3)
        self.norm2 = mk(out_channels)
        self.conv2 = _CausalConv3d(out_channels, out_channels, 3)
        self.conv_shortcut = nn.Conv3d(in_channels, out_channels, 1) if in_channels != out_channels else None

    def run(self, ctx: _Ctx, x: torch.Tensor, T: int, h: int, w: int) -> torch.Tensor:
        own = ctx.owner
        grid = Grid3D(T, ctx.B, h, w)
        b1 = own.norm_to_grid(ctx, self.norm1, x, grid, self.groups, self.eps)
        h1 = own.causal_conv(ctx, self.conv1, b1, grid)
        b2 = own.norm_to_grid(ctx, self.norm2, h1, grid, self.groups, self.eps)
        sc = x
        if self.conv_shortcut is not None:
            ws = own.packed(self.conv_shortcut, lambda: _bf(self.conv_shortcut.weight).reshape(self.conv_shortcut.weight.shape[0], -1).contiguous())
            sc = ops.gemm(x, ws, _bf(self.conv_shortcut.bias))
        return own.causal_conv(ctx, self.conv2, b2, grid, epilogue=EPI_RESID, res=sc, out=h1)


class _Sampler3D(nn.Module):
    """CogVideoXDownsample3D (Conv2d 3x3 stride 2, no padding) / CogVideoXUpsample3D (Conv2d 3x3 padding 1): holds `.conv`"""

    def __init__(self, channels: int, down: bool, compress_time: bool):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, stride=2 if down else 1, padding=0 if down else 1)
        self.compress_time = compress_time


class _Block3D(nn.Module):
    def __init__(self):
        super().__init__()
        self.resnets = nn.ModuleList()
        self.downsamplers = None
        self.upsamplers = None


class Encoder3D(nn.Module):
    def __init__(self, in_channels, latent_channels, block_out_channels, layers_per_block, groups, eps, time_levels):
        super().__init__()
        ch = list(block_out_channels)
        self.conv_in = _CausalConv3d(in_channels, ch[0], 3)
        self.down_blocks = nn.ModuleList()
        prev = ch[0]
        for i, o in enumerate(ch):
            blk = _Block3D()
            for j in range(layers_per_block):
                blk.resnets.append(_Resnet3D(prev if j == 0 else o, o, groups, eps, None))
            if i != len(ch) - 1:
                blk.downsamplers = nn.ModuleList([_Sampler3D(o, True, i < time_levels)])
            self.down_blocks.append(blk)
            prev = o
        self.mid_block = _Block3D()
        for _ in range(2):
            self.mid_block.resnets.append(_Resnet3D(ch[-1], ch[-1], groups, eps, None))
        self.norm_out = nn.GroupNorm(groups, ch[-1], eps=1e-6)
        self.conv_out = _CausalConv3d(ch[-1], 2 * latent_channels, 3)


class Decoder3D(nn.Module):
    def __init__(self, latent_channels, out_channels, block_out_channels, layers_per_block, groups, eps, time_levels):
        super().__init__()
        ch = list(block_out_channels)[::-1]
        self.conv_in = _CausalConv3d(latent_channels, ch[0], 3)
        self.mid_block = _Block3D()
        for _ in range(2):
            self.mid_block.resnets.append(_Resnet3D(ch[0], ch[0], groups, eps, latent_channels))
        self.up_blocks = nn.ModuleList()
        prev = ch[0]
        for i, o in enumerate(ch):
            blk = _Block3D()
            for j in range(layers_per_block + 1):
                blk.resnets.append(_Resnet3D(prev if j == 0 else o, o, groups, eps, latent_channels))
            if i != len(ch) - 1:
                blk.upsamplers = nn.ModuleList([_Sampler3D(o, False, i < time_levels)])
            self.up_blocks.append(blk)
            prev = o
        self.norm_out = _SpatialNorm3D(ch[-1], latent_channels, groups)
        self.conv_out = _CausalConv3d(ch[-1], out_channels, 3)
        self.out_channels = out_channels


def _chunks(n: int, size: int) -> List[Tuple[int, int]]:
    """frame ranges of AutoencoderKLCogVideoX.encode / _decode: the first chunk takes the remainder"""
    nb, rem = max(n // size, 1), n % size
    return [(size * i + (0 if i == 0 else rem), size * (i + 1) + rem) for i in range(nb)]


class AutoencoderKLCogVideoX(nn.Module):
    """encode / decode stand-in for diffusers.AutoencoderKLCogVideoX (THUDM/CogVideoX-2b VAE defaults)."""

    def __init__(self, in_channels: int = 3, out_channels: int = 3, down_block_types=None, up_block_types=None,
                 block_out_channels=(128, 256, 256, 512), latent_channels: int = 16, layers_per_block: int = 3,
                 act_fn: str = "silu", norm_eps: float = 1e-6, norm_num_groups: int = 32, temporal_compression_ratio: float = 4,
                 sample_height: int = 480, sample_width: int = 720, scaling_factor: float = 1.15258426,
                 shift_factor: Optional[float] = None, latents_mean=None, latents_std=None, force_upcast: bool = True,
                 use_quant_conv: bool = False, use_post_quant_conv: bool = False, **unused):
        super().__init__()
        if act_fn != "silu":
            raise NotImplementedError("AutoencoderKLCogVideoX: act_fn != silu")
        if use_quant_conv or use_post_quant_conv:
            raise NotImplementedError("AutoencoderKLCogVideoX: quant_conv / post_quant_conv (no released CogVideoX VAE has them)")
        tl = int(math.log2(temporal_compression_ratio))
        self.encoder = Encoder3D(in_channels, latent_channels, block_out_channels, layers_per_block, norm_num_groups, norm_eps, tl)
        self.decoder = Decoder3D(latent_channels, out_channels, block_out_channels, layers_per_block, norm_num_groups, norm_eps, tl)
        self.quant_conv = self.post_quant_conv = None
        self.config = types.SimpleNamespace(
            in_channels=in_channels, out_channels=out_channels, block_out_channels=tuple(block_out_channels),
            latent_channels=latent_channels, layers_per_block=layers_per_block, norm_eps=norm_eps,
            norm_num_groups=norm_num_groups, temporal_compression_ratio=temporal_compression_ratio,
            scaling_factor=scaling_factor, shift_factor=shift_factor, sample_height=sample_height, sample_width=sample_width,
            down_block_types=down_block_types or ("CogVideoXDownBlock3D",) * len(block_out_channels),
            up_block_types=up_block_types or ("CogVideoXUpBlock3D",) * len(block_out_channels))
        self.num_latent_frames_batch_size = 2
        self.num_sample_frames_batch_size = 8
        # bf16 (storage bf16, fp32 accumulation / statistics), or torch.float32: the fp32 accuracy path (north_star's 1e-3) - the
        # causal 3x3x3 convolutions through dwm_gemm_f32 (27 taps as three groups of 9), GroupNorm / spatial norm / frame mixes /
        # resampling in their fp32 forms; the reference runs this VAE in whatever dtype the caller loaded it in
        self.compute_dtype = bf16
        self._scratch: Dict[tuple, torch.Tensor] = {}
        self._packed: Dict[int, torch.Tensor] = {}

    from_pretrained = classmethod(AutoencoderKL.from_pretrained.__func__)

    @property
    def dtype(self):
        return self.decoder.conv_in.conv.weight.dtype

    def _apply(self, fn, *a, **kw):
        out = super()._apply(fn, *a, **kw)
        self._scratch, self._packed = {}, {}
        return out

    def load_state_dict(self, state_dict, *a, **kw):
        out = super().load_state_dict(state_dict, *a, **kw)
        self._packed = {}
        return out

    # ------------------------------------------------------------------ helpers
    def packed(self, module: nn.Module, make) -> torch.Tensor:
        key = (id(module), STORE.precision)                  # one set per compute precision
        t = self._packed.get(key)
        if t is None:
            t = self._packed[key] = make()
        return t

    def scratch(self, grid, channels: int) -> torch.Tensor:
        """zero-bordered padded buffers, reused: every producer rewrites the whole interior, the border stays zero"""
        key = (type(grid).__name__,) + tuple(getattr(grid, f) for f in grid.__dataclass_fields__) + (channels, STORE.precision)
        dev = self.decoder.conv_in.conv.weight.device
        buf = self._scratch.get(key)
        if buf is None or buf.device != dev:
            buf = self._scratch[key] = torch.zeros((grid.rows, channels), dtype=STORE.precision, device=dev)
        return buf

    @staticmethod
    def _conv27_w(conv: nn.Conv3d, k_pad: Optional[int] = None, n_pad: Optional[int] = None) -> torch.Tensor:
        """[N, C, 3, 3, 3] -> tap-major [Np, 27*Cp] (dt, dy, dx, c)"""
        w = _bf(conv.weight)
        n, c = w.shape[:2]
        cp, npad = k_pad or c, n_pad or n
        t = torch.zeros((npad, 3, 3, 3, cp), dtype=w.dtype, device=w.device)
        t[:n, ..., :c] = w.permute(0, 2, 3, 4, 1)
        return t.reshape(npad, 27 * cp).contiguous()

    def causal_conv(self, ctx: _Ctx, mod: _CausalConv3d, buf: torch.Tensor, grid: Grid3D, n_pad: Optional[int] = None, **kw):
        """buf: padded Grid3D rows whose interior (frames 2 .. T+1) holds this call's input.  Context frames = the cache
        of the previous chunk, else the first frame twice; the last two frames become the next chunk's cache."""
        fr = grid.frame_rows
        prev = ctx.cache.get(id(mod))
        if prev is None:
            buf[:fr].copy_(buf[2 * fr:3 * fr])
            buf[fr:2 * fr].copy_(buf[2 * fr:3 * fr])
        else:
            buf[:2 * fr].copy_(prev)
        ctx.cache[id(mod)] = buf[grid.T * fr:(grid.T + 2) * fr].clone()
        w = self.packed(mod, lambda: self._conv27_w(mod.conv, buf.shape[1], n_pad))
        bias = _bf(mod.conv.bias)
        if n_pad is not None and n_pad != bias.shape[0]:
            bias = self.packed(mod.conv, lambda: torch.cat([bias, torch.zeros(n_pad - bias.shape[0], dtype=bias.dtype, device=bias.device)]))
        return ops.gemm(buf, w, bias, a_grid=grid, conv_taps=grid.tap_shifts(), **kw)

    def norm_to_grid(self, ctx: _Ctx, norm: nn.Module, x: torch.Tensor, grid: Grid3D, groups: int, eps: float) -> torch.Tensor:
        """SiLU(norm(x)) written into the interior of the convolution's padded grid"""
        B, T, h, w, C_ = ctx.B, grid.T, grid.h, grid.w, x.shape[1]
        imap = (B, h * w, 0, h * w, B * h * w)              # video b, pixel (t, y, x) -> row (t*B + b)*h*w + y*w + x
        out = self.scratch(grid, C_)
        if isinstance(norm, nn.GroupNorm):
            return ops.groupnorm_silu(x, B, T * h * w, _bf(norm.weight), _bf(norm.bias), groups, eps, out=out, out_grid=grid,
                                      img_map=imap)
        def make():
            zc = norm.conv_y.conv.weight.shape[1]
            wyb = torch.zeros((2 * C_, ctx.zrows.shape[1]), dtype=STORE.precision, device=x.device)
            wyb[:C_, :zc] = _bf(norm.conv_y.conv.weight).reshape(C_, zc)
            wyb[C_:, :zc] = _bf(norm.conv_b.conv.weight).reshape(C_, zc)
            return wyb
        wyb = self.packed(norm, make)
        byb = self.packed(norm.conv_y, lambda: torch.cat([_bf(norm.conv_y.conv.bias), _bf(norm.conv_b.conv.bias)]).contiguous())
        mod = ops.gemm(ctx.zrows, wyb, byb)                  # [Tz*B*hz*wz, 2C]: conv_y(zq) | conv_b(zq)
        shift = int(math.log2(h // ctx.hz))
        if (ctx.hz << shift) != h or (ctx.wz << shift) != w:
            raise RuntimeError("CogVideoXSpatialNorm3D: feature map is not a power-of-two multiple of the latent")
        zmap = dict(mod=mod, frames=T, videos=B, h=h, w=w, shift=shift, zt=ctx.zt(T))
        nl = norm.norm_layer
        return ops.groupnorm_silu(x, B, T * h * w, _bf(nl.weight), _bf(nl.bias), groups, 1e-6, out=out, out_grid=grid,
                                  img_map=imap, zmap=zmap)

    def _downsample(self, ctx: _Ctx, ds: _Sampler3D, x: torch.Tensor, T: int, h: int, w: int):
        C_ = x.shape[1]
        if ds.compress_time:
            if T % 2 == 1:
                f0 = [0] + list(range(1, T, 2))
                f1 = [0] + list(range(2, T, 2))
                w0 = [1.0] + [0.5] * ((T - 1) // 2)
                w1 = [0.0] + [0.5] * ((T - 1) // 2)
            else:
                f0, f1 = list(range(0, T, 2)), list(range(1, T, 2))
                w0 = w1 = [0.5] * (T // 2)
            if len(f0) != T:
                x = ops.frame_mix(x, ctx.B * h * w * C_, f0, f1, w0, w1).view(-1, C_)
                T = len(f0)
        if h % 2 or w % 2:
            raise NotImplementedError("CogVideoX encoder needs even feature-map sizes at every downsample")
        g2 = PaddedGrid(T * ctx.B, h, w)
        xp = ops.pad_tokens(x, g2, out=self.scratch(g2, C_))
        wd = self.packed(ds, lambda: _conv3_w(ds.conv))
        return ops.gemm(xp, wd, _bf(ds.conv.bias), a_grid=g2, conv3x3=True, stride2=True), T, h // 2, w // 2

    def _upsample(self, ctx: _Ctx, up: _Sampler3D, x: torch.Tensor, T: int, h: int, w: int):
        C_ = x.shape[1]
        if up.compress_time and T > 1:
            src = ([0] + [1 + j // 2 for j in range(2 * (T - 1))]) if T % 2 == 1 else [j // 2 for j in range(2 * T)]
            x = ops.frame_mix(x, ctx.B * h * w * C_, src, src, [1.0] * len(src), [0.0] * len(src)).view(-1, C_)
            T = len(src)
        g2 = PaddedGrid(T * ctx.B, 2 * h, 2 * w)
        xp = ops.upsample2_padded(x, T * ctx.B, h, w, out=self.scratch(g2, C_))
        wu = self.packed(up, lambda: _conv3_w(up.conv))
        return ops.gemm(xp, wu, _bf(up.conv.bias), a_grid=g2, conv3x3=True), T, 2 * h, 2 * w

    # ------------------------------------------------------------------ encode
    @torch.no_grad()
    def encode(self, x: torch.Tensor, return_dict: bool = True):
        """x [B, 3, T, H, W] in [-1, 1] -> object with .latent_dist over the moments [B, 2*latent, T', H/8, W/8]
        (ctsd.py:1206-1218: `.latent_dist.sample()`; :1689-1694 `.mode()`)."""
        if not x.is_cuda:
            raise RuntimeError("opendwm_amd VAE runs on an MI355X (HIP) device only")
        if x.dim() != 5:
            raise ValueError("AutoencoderKLCogVideoX.encode expects [B, C, T, H, W]")
        if self.compute_dtype not in (bf16, torch.float32):
            raise ValueError("compute_dtype must be torch.bfloat16 or torch.float32")
        STORE.set_precision(self.compute_dtype)
        try:
            ctx = _Ctx(self, x.shape[0], x.device)
            parts = [self._encode_chunk(ctx, x[:, :, a:b]) for a, b in _chunks(x.shape[2], self.num_sample_frames_batch_size)]
        finally:
            STORE.set_precision(bf16)
        dist = DiagonalGaussianDistribution(torch.cat(parts, 2).float())
        if return_dict:
            return types.SimpleNamespace(latent_dist=dist)
        return (dist,)

    def _encode_chunk(self, ctx: _Ctx, x: torch.Tensor) -> torch.Tensor:
        e = self.encoder
        B, ic, T, H, W = x.shape
        cd = STORE.precision
        xt = x.permute(2, 0, 1, 3, 4).reshape(T * B, ic, H, W).contiguous()
        if cd == torch.float32:
            xt = xt.float()
        elif xt.dtype not in (torch.float32, bf16):
            xt = xt.to(bf16)
        tok = ops.unshuffle_tokens(xt, 1, 64 * ((ic + 63) // 64), dtype=cd)
        grid = Grid3D(T, B, H, W)
        buf = ops.pad_tokens(tok, grid, out=self.scratch(grid, tok.shape[1]))
        hcur = self.causal_conv(ctx, e.conv_in, buf, grid)
        h, w = H, W
        for blk in e.down_blocks:
            for res in blk.resnets:
                hcur = res.run(ctx, hcur, T, h, w)
            if blk.downsamplers is not None:
                hcur, T, h, w = self._downsample(ctx, blk.downsamplers[0], hcur, T, h, w)
        for res in e.mid_block.resnets:
            hcur = res.run(ctx, hcur, T, h, w)
        grid = Grid3D(T, B, h, w)
        g = e.down_blocks[0].resnets[0].groups
        buf = self.norm_to_grid(ctx, e.norm_out, hcur, grid, g, 1e-6)
        m = self.causal_conv(ctx, e.conv_out, buf, grid)                     # [T*B*h*w, 2*latent]
        return m.reshape(T, B, h, w, -1).permute(1, 4, 0, 2, 3).contiguous()

    # ------------------------------------------------------------------ decode
    @torch.no_grad()
    def decode(self, z: torch.Tensor, return_dict: bool = False):
        """z [B, latent, T', h, w] -> frames [B, 3, T, 8h, 8w] (in compute_dtype); 1-tuple like diffusers' decode(return_dict=False)"""
        if not z.is_cuda:
            raise RuntimeError("opendwm_amd VAE runs on an MI355X (HIP) device only")
        if z.dim() != 5:
            raise ValueError("AutoencoderKLCogVideoX.decode expects [B, C, T, h, w]")
        if self.compute_dtype not in (bf16, torch.float32):
            raise ValueError("compute_dtype must be torch.bfloat16 or torch.float32")
        STORE.set_precision(self.compute_dtype)
        try:
            ctx = _Ctx(self, z.shape[0], z.device)
            parts = [self._decode_chunk(ctx, z[:, :, a:b]) for a, b in _chunks(z.shape[2], self.num_latent_frames_batch_size)]
        finally:
            STORE.set_precision(bf16)
        out = torch.cat(parts, 2)
        if return_dict:
            return types.SimpleNamespace(sample=out)
        return (out,)

    def _decode_chunk(self, ctx: _Ctx, z: torch.Tensor) -> torch.Tensor:
        d = self.decoder
        B, lc, T, h, w = z.shape
        cd = STORE.precision
        zt = z.permute(2, 0, 1, 3, 4).reshape(T * B, lc, h, w).contiguous()
        if cd == torch.float32:
            zt = zt.float()
        elif zt.dtype not in (torch.float32, bf16):
            zt = zt.to(bf16)
        ctx.zrows = ops.unshuffle_tokens(zt, 1, 64 * ((lc + 63) // 64), dtype=cd)     # [T*B*h*w, 64]: zq of every spatial norm
        ctx.Tz, ctx.hz, ctx.wz = T, h, w
        grid = Grid3D(T, B, h, w)
        buf = ops.pad_tokens(ctx.zrows, grid, out=self.scratch(grid, ctx.zrows.shape[1]))
        x = self.causal_conv(ctx, d.conv_in, buf, grid)
        for res in d.mid_block.resnets:
            x = res.run(ctx, x, T, h, w)
        for blk in d.up_blocks:
            for res in blk.resnets:
                x = res.run(ctx, x, T, h, w)
            if blk.upsamplers is not None:
                x, T, h, w = self._upsample(ctx, blk.upsamplers[0], x, T, h, w)
        grid = Grid3D(T, B, h, w)
        g = d.mid_block.resnets[0].groups
        buf = self.norm_to_grid(ctx, d.norm_out, x, grid, g, 1e-6)
        oc = d.out_channels
        y = self.causal_conv(ctx, d.conv_out, buf, grid, n_pad=8 * ((oc + 7) // 8))           # [T*B*h*w, 8]
        return y[:, :oc].reshape(T, B, h, w, oc).permute(1, 4, 0, 2, 3).contiguous()
