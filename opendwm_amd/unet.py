"""MI355X-native SD 2.1 cross-view temporal UNet: drop-in for
`dwm.models.crossview_temporal_unet.UNetCrossviewTemporalConditionModel`
(src/dwm/models/crossview_temporal_unet.py:355-835; blocks :10-352; ResBlock / TransformerModel /
TemporalBasicTransformerBlock src/dwm/models/crossview_temporal.py:75-514) - same constructor kwargs, forward
signature, return structure and state-dict keys.  This file is the inference path; in train() mode the forward runs
opendwm_amd.train_unet (block Functions with hand-written HIP backward).

Activations stay token-major `[(b t v)(h w), C]` bf16 for the whole network:
  * every 3x3 convolution (ResnetBlock2D, Downsample2D stride 2, Upsample2D, conv_in / conv_out) is an implicit GEMM
    of dwm_gemm_bf16 over a zero-bordered padded token grid that the preceding GroupNorm+SiLU kernel writes directly;
    the time-embedding add is the GEMM's per-image residual;
  * TemporalResnetBlock's Conv3d (3,1,1) is the same GEMM with 3 taps over a T-padded row layout, its GroupNorm over
    (T,H,W) a row-mapped GroupNorm; the AlphaBlender mix is the last conv's epilogue;
  * TransformerModel: GroupNorm -> proj_in -> BasicTransformerBlock (fused qkv GEMM, flash self-attention,
    cross-attention to the text tokens, GEGLU) -> cross-view / temporal TemporalBasicTransformerBlock (the same
    VTSelfAttentionBlock kernels as the MMDiT, row-wise rearranges folded into attention addressing) -> proj_out (+x);
  * skip concatenations are column-slice copies into one buffer (the next GroupNorm needs the concatenated rows).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
from torch import nn

from . import ops
from . import train_ops as T
from .blocks import AlphaBlender, Attention, FeedForward, TimestepEmbedding, VTSelfAttentionBlock, _bf, geglu_pack, STORE
from .ops import EPI_GEGLU, EPI_RESID, PaddedGrid, TimeGrid

bf16 = torch.bfloat16

try:
    import diffusers as _diffusers   # noqa: F401
    _Base = _diffusers.UNetSpatioTemporalConditionModel
except Exception:
    _Base = nn.Module


def _conv3_w(w: torch.Tensor, c_pad: Optional[int] = None, n_pad: Optional[int] = None) -> torch.Tensor:
    """[N, C, 3, 3] -> tap-major [Np, 9*Cp] in the compute dtype (zero padded)"""
    w = _bf(w)
    n, c = w.shape[:2]
    t = torch.zeros((n_pad or n, 3, 3, c_pad or c), dtype=w.dtype, device=w.device)
    t[:n, :, :, :c] = w.permute(0, 2, 3, 1)
    return t.reshape(t.shape[0], -1).contiguous()


def _conv3d_w(w: torch.Tensor) -> torch.Tensor:
    """Conv3d (3,1,1) weight [N, C, 3, 1, 1] -> tap-major [N, 3*C]"""
    w = _bf(w)
    n, c = w.shape[:2]
    return w.reshape(n, c, 3).permute(0, 2, 1).reshape(n, 3 * c).contiguous()


class _Scratch:
    """zero-bordered padded buffers, reused (kernels only ever write their interiors)"""

    def __init__(self):
        self.buf: Dict[tuple, torch.Tensor] = {}

    def get(self, tag: str, rows: int, channels: int, device) -> torch.Tensor:
        dt = STORE.precision                                  # bf16, or fp32 while the accuracy path runs
        k = (tag, rows, channels, str(device), dt)
        t = self.buf.get(k)
        if t is None:
            t = torch.zeros((rows, channels), dtype=dt, device=device)
            self.buf[k] = t
        return t


class _Geom:
    """per-call geometry: batch, frames, views and the level's (h, w)"""

    def __init__(self, B, T, V, h, w, scratch: _Scratch):
        self.B, self.T, self.V, self.h, self.w, self.scratch = B, T, V, h, w, scratch
        self.I, self.N = B * T * V, h * w

    def at(self, h, w):
        return _Geom(self.B, self.T, self.V, h, w, self.scratch)


class _TimeProj:
    """`time_emb_proj(silu(emb))` of every residual block (ResnetBlock2D / TemporalResnetBlock) in ONE GEMM: they
    all read the same [images, temb_channels] input, so their weights are stacked once per weight version and each
    block takes its column slice of the [images, sum C_out] result (41 skinny launches -> 1)."""

    def __init__(self, layers, silu_emb: torch.Tensor):
        anchor = layers[0].weight
        wcat = STORE.derived(anchor, "tproj_w", lambda: torch.cat([_bf(l.weight) for l in layers], 0).contiguous())
        bcat = STORE.derived(anchor, "tproj_b", lambda: torch.cat([_bf(l.bias) for l in layers], 0).contiguous())
        self.all = ops.gemm(silu_emb, wcat, bcat)
        self.slices, off = {}, 0
        for l in layers:
            self.slices[id(l)] = (off, l.weight.shape[0])
            off += l.weight.shape[0]

    def of(self, layer) -> torch.Tensor:
        off, n = self.slices[id(layer)]
        return self.all[:, off:off + n]


class ResnetBlock2D(nn.Module):
    """diffusers ResnetBlock2D(in, out, temb_channels, eps, groups=32)"""

    def __init__(self, in_channels: int, out_channels: int, temb_channels: int, eps: float):
        super().__init__()
        self.eps = eps
        self.norm1 = nn.GroupNorm(32, in_channels, eps=eps)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels)
        self.norm2 = nn.GroupNorm(32, out_channels, eps=eps)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 1) if in_channels != out_channels else None

    def run(self, x: torch.Tensor, tproj: "_TimeProj", g: _Geom) -> torch.Tensor:
        grid = PaddedGrid(g.I, g.h, g.w)
        w1 = STORE.derived(self.conv1.weight, "c3", lambda: _conv3_w(self.conv1.weight))
        w2 = STORE.derived(self.conv2.weight, "c3", lambda: _conv3_w(self.conv2.weight))
        p1 = ops.groupnorm_silu(x, g.I, g.N, _bf(self.norm1.weight), _bf(self.norm1.bias), 32, self.eps,
                                out=g.scratch.get("s1", grid.rows, x.shape[1], x.device), out_grid=grid)
        tp = tproj.of(self.time_emb_proj)
        h1 = ops.gemm(p1, w1, _bf(self.conv1.bias), a_grid=grid, conv3x3=True, epilogue=EPI_RESID, res=tp, res_mod=-g.N)
        p2 = ops.groupnorm_silu(h1, g.I, g.N, _bf(self.norm2.weight), _bf(self.norm2.bias), 32, self.eps,
                                out=g.scratch.get("s2", grid.rows, h1.shape[1], x.device), out_grid=grid)
        if self.conv_shortcut is not None:
            ws = STORE.derived(self.conv_shortcut.weight, "c1", lambda: _bf(self.conv_shortcut.weight).reshape(self.conv_shortcut.weight.shape[0], -1).contiguous())
            x = ops.gemm(x, ws, _bf(self.conv_shortcut.bias))
        return ops.gemm(p2, w2, _bf(self.conv2.bias), a_grid=grid, conv3x3=True, epilogue=EPI_RESID, res=x, out=h1)


class TemporalResnetBlock(nn.Module):
    """diffusers TemporalResnetBlock(in, out, temb_channels, eps): Conv3d kernel (3,1,1)"""

    def __init__(self, channels: int, temb_channels: int, eps: float):
        super().__init__()
        self.eps = eps
        self.norm1 = nn.GroupNorm(32, channels, eps=eps)
        self.conv1 = nn.Conv3d(channels, channels, (3, 1, 1), padding=(1, 0, 0))
        self.time_emb_proj = nn.Linear(temb_channels, channels)
        self.norm2 = nn.GroupNorm(32, channels, eps=eps)
        self.conv2 = nn.Conv3d(channels, channels, (3, 1, 1), padding=(1, 0, 0))

    def run(self, s: torch.Tensor, tproj: "_TimeProj", g: _Geom, alpha: torch.Tensor) -> torch.Tensor:
        """returns AlphaBlender(s, s + temporal_resnet(s)) (crossview_temporal.py:146-162)"""
        tg = TimeGrid(g.B, g.T, g.V * g.N)
        imap = (g.V, g.N, g.T * g.V * g.N, g.N, g.V * g.N)
        Cc = s.shape[1]
        w1 = STORE.derived(self.conv1.weight, "c3d", lambda: _conv3d_w(self.conv1.weight))
        w2 = STORE.derived(self.conv2.weight, "c3d", lambda: _conv3d_w(self.conv2.weight))
        t1 = ops.groupnorm_silu(s, g.B * g.V, g.T * g.N, _bf(self.norm1.weight), _bf(self.norm1.bias), 32, self.eps,
                                out=g.scratch.get("t1", tg.rows, Cc, s.device), out_grid=tg, img_map=imap)
        tp = tproj.of(self.time_emb_proj)
        u1 = ops.gemm(t1, w1, _bf(self.conv1.bias), a_grid=tg, conv_taps=tg.tap_shifts(), epilogue=EPI_RESID, res=tp, res_mod=-g.N)
        t2 = ops.groupnorm_silu(u1, g.B * g.V, g.T * g.N, _bf(self.norm2.weight), _bf(self.norm2.bias), 32, self.eps,
                                out=g.scratch.get("t2", tg.rows, Cc, s.device), out_grid=tg, img_map=imap)
        return ops.gemm(t2, w2, _bf(self.conv2.bias), a_grid=tg, conv_taps=tg.tap_shifts(), epilogue=EPI_RESID, res=s, blend=s,
                        alpha=alpha, rows_per_alpha=g.T * g.V * g.N, out=u1)


class ResBlock(nn.Module):
    """dwm.models.crossview_temporal.ResBlock (:75-164)"""

    def __init__(self, in_channels: int, out_channels: int, temb_channels: int, eps: float, enable_temporal: bool,
                 merge_factor: float):
        super().__init__()
        self.spatial_res_block = ResnetBlock2D(in_channels, out_channels, temb_channels, eps)
        if enable_temporal:
            self.temporal_res_block = TemporalResnetBlock(out_channels, temb_channels, eps)
            self.time_mixer = AlphaBlender(merge_factor, merge_strategy="learned_with_images")
        else:
            self.temporal_res_block = None

    def run(self, x, tproj: "_TimeProj", g: _Geom, disable_temporal):
        s = self.spatial_res_block.run(x, tproj, g)
        if self.temporal_res_block is None:
            return s
        alpha = self.time_mixer.get_alpha(disable_temporal, g.B)
        return self.temporal_res_block.run(s, tproj, g, alpha)


class _CrossAttention(nn.Module):
    """diffusers Attention(query_dim, cross_attention_dim, heads, dim_head): bias-free q/k/v, biased out"""

    def __init__(self, dim: int, cross_dim: int, heads: int):
        super().__init__()
        self.heads = heads
        self.to_q = nn.Linear(dim, dim, bias=False)
        self.to_k = nn.Linear(cross_dim, dim, bias=False)
        self.to_v = nn.Linear(cross_dim, dim, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(dim, dim), nn.Identity()])

    def wkv(self):
        return STORE.derived(self.to_k.weight, "kv", lambda: torch.cat([_bf(self.to_k.weight), _bf(self.to_v.weight)], 0).contiguous())


class _TextContext:
    """Text tokens [images * L, cross_attention_dim] of one `encoder_hidden_states` tensor and the K/V projections of
    every text cross-attention layer on them.  Both are functions of the text embeddings and the weights only, so the
    UNet keeps one instance across denoise steps while the caller passes the same (unmodified) tensor."""

    def __init__(self, encoder_hidden_states: torch.Tensor):
        self.source = encoder_hidden_states                  # held: its storage cannot be recycled under the cache key
        self.key = (encoder_hidden_states.data_ptr(), encoder_hidden_states._version, tuple(encoder_hidden_states.shape),
                    encoder_hidden_states.dtype, STORE.step, STORE.precision)
        ehs = encoder_hidden_states.flatten(0, -3)
        ehs = ehs if ehs.dtype == STORE.precision else ehs.to(STORE.precision)
        self.rows = ehs.reshape(ehs.shape[0] * ehs.shape[1], -1).contiguous()
        self._kv = {}

    @classmethod
    def from_rows(cls, rows: torch.Tensor) -> "_TextContext":
        """a context over text rows that are already flattened to [images * L, cross_attention_dim] bf16 (training path)"""
        self = cls.__new__(cls)
        self.source, self.key, self.rows, self._kv = None, None, rows, {}
        return self

    def matches(self, encoder_hidden_states: torch.Tensor) -> bool:
        t = encoder_hidden_states
        return t is self.source and self.key == (t.data_ptr(), t._version, tuple(t.shape), t.dtype, STORE.step, STORE.precision)

    def kv(self, attn: "_CrossAttention") -> torch.Tensor:
        out = self._kv.get(id(attn))
        if out is None:
            out = self._kv[id(attn)] = ops.gemm(self.rows, attn.wkv())
        return out


class BasicTransformerBlock(nn.Module):
    """diffusers BasicTransformerBlock(dim, heads, head_dim, cross_attention_dim): LayerNorm, GEGLU FF"""

    def __init__(self, dim: int, heads: int, head_dim: int, cross_attention_dim: int):
        super().__init__()
        if head_dim != 64:
            raise NotImplementedError("attention kernels are built for head_dim 64 (SD 2.1: channels / heads = 64)")
        self.dim, self.heads = dim, heads
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = Attention(dim, heads, head_dim, bias=False)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = _CrossAttention(dim, cross_attention_dim, heads)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim, activation_fn="geglu")

    def run(self, h: torch.Tensor, ctx: "_TextContext", n_img: int) -> torch.Tensor:
        D = self.dim
        N = h.shape[0] // n_img
        ln = lambda x, n, **kw: ops.layernorm(x, eps=1e-5, weight=_bf(n.weight), bias=_bf(n.bias), **kw)
        y = ln(h, self.norm1)
        qkv = self.attn1.project_qkv(y)
        ao = y
        ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], ao, ops.rowmap_identity(n_img, N), self.heads)
        o1 = self.attn1.to_out[0]
        ops.gemm(ao, _bf(o1.weight), _bf(o1.bias), epilogue=EPI_RESID, res=h, out=h)
        y = ln(h, self.norm2, out=y)
        q = ops.gemm(y, _bf(self.attn2.to_q.weight))
        kv = ctx.kv(self.attn2)
        ops.cross_attention(q, kv[:, :D], kv[:, D:], ao, n_img, self.heads)
        o2 = self.attn2.to_out[0]
        ops.gemm(ao, _bf(o2.weight), _bf(o2.bias), epilogue=EPI_RESID, res=h, out=h)
        y = ln(h, self.norm3, out=y)
        p = self.ff.net[0].proj
        wp = STORE.derived(p.weight, "geglu", lambda: geglu_pack(_bf(p.weight)))
        bp = STORE.derived(p.bias, "geglu", lambda: geglu_pack(_bf(p.bias)))
        gg = ops.gemm(y, wp, bp, epilogue=EPI_GEGLU)
        l2 = self.ff.net[2]
        ops.gemm(gg, _bf(l2.weight), _bf(l2.bias), epilogue=EPI_RESID, res=h, out=h)
        return h


class TransformerModel(nn.Module):
    """dwm.models.crossview_temporal.TransformerModel (:269-514)"""

    def __init__(self, num_attention_heads: int, attention_head_dim: int, in_channels: int, enable_crossview: bool,
                 enable_temporal: bool, enable_rowwise_crossview: bool, enable_rowwise_temporal: bool, num_layers: int,
                 cross_attention_dim: int, merge_factor: float):
        super().__init__()
        inner = num_attention_heads * attention_head_dim
        if inner != in_channels:
            raise NotImplementedError("TransformerModel: inner_dim != in_channels is never built by the UNet")
        self.heads, self.in_channels = num_attention_heads, in_channels
        self.rowwise_cv, self.rowwise_t = enable_rowwise_crossview, enable_rowwise_temporal
        self.norm = nn.GroupNorm(32, in_channels, eps=1e-6)
        self.proj_in = nn.Linear(in_channels, inner)
        self.transformer_blocks = nn.ModuleList([
            BasicTransformerBlock(inner, num_attention_heads, attention_head_dim, cross_attention_dim) for _ in range(num_layers)])
        if enable_crossview:
            self.view_pos_embed = TimestepEmbedding(in_channels, in_channels * 4, out_dim=in_channels)
            self.crossview_transformer_blocks = nn.ModuleList([
                VTSelfAttentionBlock(inner, inner, num_attention_heads, attention_head_dim) for _ in range(num_layers)])
            self.view_mixer = AlphaBlender(merge_factor, merge_strategy="learned_with_images")
        else:
            self.view_pos_embed = None
        if enable_temporal:
            self.time_pos_embed = TimestepEmbedding(in_channels, in_channels * 4, out_dim=in_channels)
            self.temporal_transformer_blocks = nn.ModuleList([
                VTSelfAttentionBlock(inner, inner, num_attention_heads, attention_head_dim) for _ in range(num_layers)])
            self.time_mixer = AlphaBlender(merge_factor, merge_strategy="learned_with_images")
        else:
            self.time_pos_embed = None
        self.proj_out = nn.Linear(inner, in_channels)
        self._emb_cache = (None, None, None)

    def run(self, x: torch.Tensor, ctx: "_TextContext", g: _Geom, disable_crossview, disable_temporal, mask) -> torch.Tensor:
        B, Tn, V, C = g.B, g.T, g.V, self.in_channels
        dev = x.device
        hn = ops.groupnorm_silu(x, g.I, g.N, _bf(self.norm.weight), _bf(self.norm.bias), 32, 1e-6, silu=False)
        h = ops.gemm(hn, _bf(self.proj_in.weight), _bf(self.proj_in.bias))
        # view / frame index embeddings depend on (B, T, V) and the weights only: kept across denoise steps
        key = (STORE.step, B, Tn, V, str(dev), STORE.precision)
        if self._emb_cache[0] != key:
            view_emb = seq_emb = None
            if self.view_pos_embed is not None:
                idx = torch.arange(V, device=dev).view(1, 1, V).expand(B, Tn, V)
                view_emb = self.view_pos_embed.run(ops.timestep_sinusoid(idx, C, dtype=STORE.precision))
            if self.time_pos_embed is not None:
                idx = torch.arange(Tn, device=dev).view(1, Tn, 1).expand(B, Tn, V)
                seq_emb = self.time_pos_embed.run(ops.timestep_sinusoid(idx, C, dtype=STORE.precision))
            self._emb_cache = (key, view_emb, seq_emb)
        _, view_emb, seq_emb = self._emb_cache
        if self.view_pos_embed is not None:
            alpha_v = self.view_mixer.get_alpha(disable_crossview, B)
        if self.time_pos_embed is not None:
            alpha_t = self.time_mixer.get_alpha(disable_temporal, B)
        for l, blk in enumerate(self.transformer_blocks):
            h = blk.run(h, ctx, g.I)
            if self.view_pos_embed is not None:
                rm = ops.rowmap_crossview_rowwise(B, Tn, V, g.h, g.w) if self.rowwise_cv else \
                    ops.rowmap_crossview_pointwise(B, Tn, V, g.h, g.w)
                self.crossview_transformer_blocks[l].run(h, rm, emb=view_emb, rows_per_emb=g.N, group_mask=mask,
                                                         blend_alpha=alpha_v, rows_per_alpha=Tn * V * g.N, blend_into=h)
            if self.time_pos_embed is not None:
                rm = ops.rowmap_temporal_rowwise(B, Tn, V, g.h, g.w) if self.rowwise_t else \
                    ops.rowmap_temporal_pointwise(B, Tn, V, g.h, g.w)
                self.temporal_transformer_blocks[l].run(h, rm, emb=seq_emb, rows_per_emb=g.N,
                                                        blend_alpha=alpha_t, rows_per_alpha=Tn * V * g.N, blend_into=h)
        # (the output must not alias the A operand `h`: column tiles of one row panel do not all run at the same time once the
        # grid exceeds the chip - 6 views x 6 frames at 32x56 - and a finished tile would overwrite rows another still reads)
        return ops.gemm(h, _bf(self.proj_out.weight), _bf(self.proj_out.bias), epilogue=EPI_RESID, res=x, out=hn)


class _Sampler(nn.Module):
    """Downsample2D(padding=1) / Upsample2D: holds `.conv`"""

    def __init__(self, channels: int, stride: int):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, stride=stride, padding=1)


class _Block(nn.Module):
    """down / up / mid block container with the reference attribute names"""

    def __init__(self):
        super().__init__()
        self.resnets = nn.ModuleList()
        self.attentions = None
        self.downsamplers = None
        self.upsamplers = None


def _concat_cols(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """torch.cat([a, b], channel) on token-major rows: two strided copies into one buffer"""
    out = torch.empty((a.shape[0], a.shape[1] + b.shape[1]), dtype=a.dtype, device=a.device)
    if a.dtype == torch.float32:                              # (fp32 accuracy path: plain strided copies)
        out[:, :a.shape[1]].copy_(a)
        out[:, a.shape[1]:].copy_(b)
        return out
    T.rowcombine(a, out=out[:, :a.shape[1]])
    T.rowcombine(b, out=out[:, a.shape[1]:])
    return out


class UNetCrossviewTemporalConditionModel(_Base):
    """Constructor kwargs of crossview_temporal_unet.py:379-404."""

    def __init__(self, sample_size=None, in_channels: int = 8, out_channels: int = 4,
                 down_block_types=("CrossAttnDownBlockCrossviewTemporal",) * 3 + ("DownBlockCrossviewTemporal",),
                 up_block_types=("UpBlockCrossviewTemporal",) + ("CrossAttnUpBlockCrossviewTemporal",) * 3,
                 block_out_channels=(320, 640, 1280, 1280), addition_time_embed_dim: int = 256,
                 projection_class_embeddings_input_dim=768, layers_per_block=2, norm_eps: float = 1e-5,
                 cross_attention_dim: int = 1024, transformer_layers_per_block=1, num_attention_heads=(5, 10, 20, 20),
                 merge_factor: float = 0.5, enable_crossview: bool = True, enable_temporal: bool = True,
                 enable_rowwise_crossview: bool = False, enable_rowwise_temporal: bool = False,
                 condition_image_adapter_config=None, depth_net_config=None, depth_frustum_range=None,
                 enforce_align_projection=None):
        nn.Module.__init__(self)
        if depth_net_config is not None or enforce_align_projection is not None:
            raise NotImplementedError("UNet: depth_net / align projection are not built")
        self.compute_dtype = bf16           # torch.float32 selects the fp32 accuracy path of the inference forward
        n = len(block_out_channels)
        as_list = lambda v: [v] * n if isinstance(v, int) else list(v)
        heads, lpb, tl = as_list(num_attention_heads), as_list(layers_per_block), as_list(transformer_layers_per_block)
        self.in_channels_, self.out_channels_ = in_channels, out_channels
        self.block_out_channels = tuple(block_out_channels)
        self.addition_time_embed_dim = addition_time_embed_dim
        c0 = block_out_channels[0]
        E = 4 * c0
        self.conv_in = nn.Conv2d(in_channels, c0, 3, padding=1)
        self.time_embedding = TimestepEmbedding(c0, E)
        self.add_embedding = TimestepEmbedding(projection_class_embeddings_input_dim, E) \
            if projection_class_embeddings_input_dim is not None else None
        common = dict(enable_crossview=enable_crossview, enable_temporal=enable_temporal,
                      enable_rowwise_crossview=enable_rowwise_crossview, enable_rowwise_temporal=enable_rowwise_temporal,
                      cross_attention_dim=cross_attention_dim, merge_factor=merge_factor)

        def res(i, o):
            return ResBlock(i, o, E, norm_eps, enable_temporal, merge_factor)

        def tm(c, h, nl):
            return TransformerModel(h, c // h, c, num_layers=nl, **common)

        self.down_blocks = nn.ModuleList()
        out_c = c0
        for i, typ in enumerate(down_block_types):
            in_c, out_c = out_c, block_out_channels[i]
            blk = _Block()
            for j in range(lpb[i]):
                blk.resnets.append(res(in_c if j == 0 else out_c, out_c))
            if typ.startswith("CrossAttn"):
                blk.attentions = nn.ModuleList([tm(out_c, heads[i], tl[i]) for _ in range(lpb[i])])
            if i != n - 1:
                blk.downsamplers = nn.ModuleList([_Sampler(out_c, 2)])
            self.down_blocks.append(blk)
        self.mid_block = _Block()
        cm = block_out_channels[-1]
        self.mid_block.resnets.extend([res(cm, cm), res(cm, cm)])
        self.mid_block.attentions = nn.ModuleList([tm(cm, heads[-1], tl[-1])])
        self.up_blocks = nn.ModuleList()
        rboc, rheads, rlpb, rtl = list(block_out_channels)[::-1], heads[::-1], lpb[::-1], tl[::-1]
        out_c = rboc[0]
        for i, typ in enumerate(up_block_types):
            prev, out_c = out_c, rboc[i]
            in_c = rboc[min(i + 1, n - 1)]
            nl = rlpb[i] + 1
            blk = _Block()
            for j in range(nl):
                skip = in_c if j == nl - 1 else out_c
                rin = prev if j == 0 else out_c
                blk.resnets.append(res(rin + skip, out_c))
            if typ.startswith("CrossAttn"):
                blk.attentions = nn.ModuleList([tm(out_c, rheads[i], rtl[i]) for _ in range(nl)])
            if i != n - 1:
                blk.upsamplers = nn.ModuleList([_Sampler(out_c, 1)])
            self.up_blocks.append(blk)
        self.conv_norm_out = nn.GroupNorm(32, c0, eps=1e-5)
        self.conv_out = nn.Conv2d(c0, out_channels, 3, padding=1)
        self.condition_image_adapter = None
        if condition_image_adapter_config is not None:                  # crossview_temporal_unet.py: layout ImageAdapter
            from .adapters import ImageAdapter
            self.condition_image_adapter = ImageAdapter(**condition_image_adapter_config)
        self._adapter_cache = (None, None)
        self.depth_net = None
        self.gradient_checkpointing = False
        self.depth_frustum_range = depth_frustum_range
        self._scratch = _Scratch()
        self._text_ctx = None
        self._tproj_layers = None

    def enable_gradient_checkpointing(self):
        """module protocol of the pipeline (ctsd.py:867-875).  The training path keeps only each block's inputs whatever
        this flag says (opendwm_amd.train_unet), as the reference's blocks do once it is set."""
        self.gradient_checkpointing = True

    def disable_gradient_checkpointing(self):
        self.gradient_checkpointing = False

    def _time_proj_layers(self):
        """every residual block's `time_emb_proj`, in module order (the column layout of the stacked projection)"""
        if self._tproj_layers is None:
            self._tproj_layers = [m.time_emb_proj for m in self.modules() if isinstance(m, (ResnetBlock2D, TemporalResnetBlock))]
        return self._tproj_layers

    @staticmethod
    def try_to_convert_state_dict(state_dict: dict):
        """SD 2.1 -> this tree key renaming (crossview_temporal_unet.py:358-373)"""
        import re
        pat = re.compile(r"resnets.(\d+).conv")
        if any(pat.search(k) for k in state_dict):
            p2 = re.compile(r"resnets.(\d+)")
            return {(p2.sub(r"resnets.\1.spatial_res_block", k) if "resnets" in k else k): v for k, v in state_dict.items()}
        return state_dict

    def _apply(self, fn, *a, **kw):
        out = super()._apply(fn, *a, **kw)
        STORE.bump()
        self._scratch = _Scratch()
        self._text_ctx = None
        self._adapter_cache = (None, None)
        return out

    def load_state_dict(self, state_dict, *a, **kw):
        out = super().load_state_dict(state_dict, *a, **kw)
        STORE.bump()
        return out

    def forward(self, sample: torch.Tensor, timesteps, frustum_bev_residuals=None, encoder_hidden_states=None,
                condition_image_tensor=None, disable_crossview=None, disable_temporal=None, crossview_attention_mask=None,
                camera_intrinsics=None, camera_transforms=None, added_time_ids=None, camera_intrinsics_norm=None,
                camera2referego=None, return_dict=False):
        """Reference signature (crossview_temporal_unet.py:655-675).  In train() mode with autograd enabled the prediction
        carries a grad_fn (opendwm_amd.train_unet: checkpointed block Functions with HIP backward kernels - the SD 2.1
        branch of the train step, ctsd.py:1240-1253); otherwise the fused inference path runs under no_grad."""
        if self.training and torch.is_grad_enabled():
            from . import train_unet as _tu
            if not sample.is_cuda:
                raise RuntimeError("opendwm_amd runs on an MI355X (HIP) device only; got a CPU tensor")
            if isinstance(encoder_hidden_states, dict):
                raise NotImplementedError("dict encoder_hidden_states (align projection) is not built")
            if encoder_hidden_states is None:
                raise ValueError("UNet training forward: encoder_hidden_states (the text condition) is required")
            # inputs the reference's forward consumes (crossview_temporal_unet.py:730-755: the frustum BEV residuals are added to
            # the down path; the camera inputs feed the depth net) but the training graph here does not: refuse, do not drop
            dropped = {"frustum_bev_residuals": frustum_bev_residuals, "camera_intrinsics": camera_intrinsics,
                       "camera_transforms": camera_transforms, "camera_intrinsics_norm": camera_intrinsics_norm,
                       "camera2referego": camera2referego}
            used = [k for k, v in dropped.items() if v is not None]
            if used:
                raise NotImplementedError(f"UNet training forward: {', '.join(used)} (depth net / frustum BEV branch) is not built; "
                                          "no shipped CTSD config enables it")
            squeeze = sample.dim() < 6
            if squeeze:
                sample, timesteps = sample.unsqueeze(2), timesteps.unsqueeze(2)
                if encoder_hidden_states is not None:
                    encoder_hidden_states = encoder_hidden_states.unsqueeze(2)
                if disable_temporal is not None:
                    disable_temporal = disable_temporal.unsqueeze(2)
            out = _tu.forward_train(self, sample, timesteps, encoder_hidden_states, disable_crossview=disable_crossview,
                                    disable_temporal=disable_temporal, crossview_attention_mask=crossview_attention_mask,
                                    added_time_ids=added_time_ids, condition_image_tensor=condition_image_tensor)
            if squeeze:
                out = out.squeeze(2)
            if return_dict:
                return {"noise_pred": out}
            return (out,), None, None
        with torch.no_grad():
            try:
                return self._forward_infer(sample, timesteps, frustum_bev_residuals, encoder_hidden_states, condition_image_tensor,
                                           disable_crossview, disable_temporal, crossview_attention_mask, camera_intrinsics,
                                           camera_transforms, added_time_ids, camera_intrinsics_norm, camera2referego, return_dict)
            finally:
                STORE.set_precision(bf16)   # the fp32 accuracy path is scoped to this forward (compute_dtype = torch.float32)

    def _forward_infer(self, sample: torch.Tensor, timesteps, frustum_bev_residuals=None, encoder_hidden_states=None,
                       condition_image_tensor=None, disable_crossview=None, disable_temporal=None, crossview_attention_mask=None,
                       camera_intrinsics=None, camera_transforms=None, added_time_ids=None, camera_intrinsics_norm=None,
                       camera2referego=None, return_dict=False):
        # compute dtype: bf16, or - `model.compute_dtype = torch.float32` - the fp32 accuracy path (north_star's 1e-3 tolerance;
        # BASELINE.json configs[0] is the reference's fp32 CPU denoise of this model, ctsd.py:1189-1193): every activation and
        # weight in fp32, contractions by dwm_gemm_f32 (incl. the stride-2 / 3-tap temporal implicit convolutions), fp32 GroupNorm /
        # LayerNorm / attention kernels
        cd = getattr(self, "compute_dtype", bf16)
        STORE.set_precision(cd)              # reset to bf16 by `forward` on the way out
        if not sample.is_cuda:
            raise RuntimeError("opendwm_amd runs on an MI355X (HIP) device only; got a CPU tensor")
        if isinstance(encoder_hidden_states, dict):
            raise NotImplementedError("dict encoder_hidden_states (align projection) is not built")
        squeeze = sample.dim() < 6
        if squeeze:
            sample, timesteps = sample.unsqueeze(2), timesteps.unsqueeze(2)
            if encoder_hidden_states is not None:
                encoder_hidden_states = encoder_hidden_states.unsqueeze(2)
            if disable_temporal is not None:
                disable_temporal = disable_temporal.unsqueeze(2)
        B, Tn, V, Cin, H, W = sample.shape
        dev = sample.device
        I = B * Tn * V
        g0 = _Geom(B, Tn, V, H, W, self._scratch)
        if disable_crossview is None:
            disable_crossview = torch.zeros(B, dtype=torch.bool, device=dev)
        if disable_temporal is None:
            disable_temporal = torch.zeros(B, dtype=torch.bool, device=dev)
        c0 = self.block_out_channels[0]

        # 1. time embeddings (crossview_temporal_unet.py:708-715)
        emb = self.time_embedding.run(ops.timestep_sinusoid(timesteps.flatten(), c0, dtype=cd))
        if added_time_ids is not None and self.add_embedding is not None:
            aug = ops.timestep_sinusoid(added_time_ids.flatten(), self.addition_time_embed_dim, dtype=cd).view(I, -1)
            emb = self.add_embedding.run(aug, res=emb)
        silu_emb = _TimeProj(self._time_proj_layers(), ops.silu(emb))
        if self._text_ctx is None or not self._text_ctx.matches(encoder_hidden_states):
            self._text_ctx = _TextContext(encoder_hidden_states)
        ctx = self._text_ctx

        # 2. conv_in: NCHW -> token-major rows with the channels zero-padded to 64, 3x3 implicit GEMM
        xin = sample.flatten(0, 2).contiguous()
        if cd == torch.float32:
            xin = xin.float()
        elif xin.dtype not in (torch.float32, bf16):
            xin = xin.to(bf16)
        tok = ops.unshuffle_tokens(xin, 1, 64, dtype=cd)
        grid = PaddedGrid(I, H, W)
        pin = ops.pad_tokens(tok, grid, out=self._scratch.get("in", grid.rows, 64, dev))
        wci = STORE.derived(self.conv_in.weight, "c3", lambda: _conv3_w(self.conv_in.weight, c_pad=64))
        x = ops.gemm(pin, wci, _bf(self.conv_in.bias), a_grid=grid, conv3x3=True)
        # layout residuals (:717-729, 748-750): one after conv_in, one after every down block - added in place, so the
        # block's last skip connection carries the sum as in the reference.  Step-invariant: cached on the tensor identity.
        residuals = []
        if self.condition_image_adapter is not None and condition_image_tensor is not None:
            key = (condition_image_tensor.data_ptr(), condition_image_tensor._version, tuple(condition_image_tensor.shape), STORE.step)
            if self._adapter_cache[0] != key:
                self._adapter_cache = (key, self.condition_image_adapter.run(condition_image_tensor, precise=True), condition_image_tensor)
            residuals = list(self._adapter_cache[1])

        def add_residual(t):
            if residuals:
                f = residuals.pop(0)
                if f.shape != t.shape:
                    raise RuntimeError(f"UNet: layout residual {tuple(f.shape)} does not match the feature map {tuple(t.shape)}")
                ops.add_(t, f)
        add_residual(x)

        # 3. down
        skips = [(x, H, W)]
        h_, w_ = H, W
        for blk in self.down_blocks:
            g = g0.at(h_, w_)
            for j, rb in enumerate(blk.resnets):
                x = rb.run(x, silu_emb, g, disable_temporal)
                if blk.attentions is not None:
                    x = blk.attentions[j].run(x, ctx, g, disable_crossview, disable_temporal, crossview_attention_mask)
                skips.append((x, h_, w_))
            if blk.downsamplers is not None:
                conv = blk.downsamplers[0].conv
                gr = PaddedGrid(I, h_, w_)
                pad = ops.pad_tokens(x, gr, out=self._scratch.get("ds", gr.rows, x.shape[1], dev))
                wd = STORE.derived(conv.weight, "c3", lambda: _conv3_w(conv.weight))
                x = ops.gemm(pad, wd, _bf(conv.bias), a_grid=gr, conv3x3=True, stride2="sym")
                h_, w_ = h_ // 2, w_ // 2
                skips.append((x, h_, w_))
            add_residual(x)
        # 4. mid
        g = g0.at(h_, w_)
        x = self.mid_block.resnets[0].run(x, silu_emb, g, disable_temporal)
        x = self.mid_block.attentions[0].run(x, ctx, g, disable_crossview, disable_temporal, crossview_attention_mask)
        x = self.mid_block.resnets[1].run(x, silu_emb, g, disable_temporal)
        # 5. up
        for blk in self.up_blocks:
            g = g0.at(h_, w_)
            for j, rb in enumerate(blk.resnets):
                sk, sh, sw = skips.pop()
                if (sh, sw) != (h_, w_):
                    raise RuntimeError("UNet: skip resolution mismatch (latent height / width must be divisible by 8)")
                x = rb.run(_concat_cols(x, sk), silu_emb, g, disable_temporal)
                if blk.attentions is not None:
                    x = blk.attentions[j].run(x, ctx, g, disable_crossview, disable_temporal, crossview_attention_mask)
            if blk.upsamplers is not None:
                conv = blk.upsamplers[0].conv
                gr = PaddedGrid(I, 2 * h_, 2 * w_)
                up = ops.upsample2_padded(x, I, h_, w_, out=self._scratch.get("us", gr.rows, x.shape[1], dev))
                wu = STORE.derived(conv.weight, "c3", lambda: _conv3_w(conv.weight))
                x = ops.gemm(up, wu, _bf(conv.bias), a_grid=gr, conv3x3=True)
                h_, w_ = 2 * h_, 2 * w_
        # 6. head: GroupNorm -> SiLU -> conv_out (output channels padded to 8 for the GEMM, dropped by the un-tokenizer)
        gr = PaddedGrid(I, H, W)
        pn = ops.groupnorm_silu(x, I, H * W, _bf(self.conv_norm_out.weight), _bf(self.conv_norm_out.bias), 32, 1e-5,
                                out=self._scratch.get("s1", gr.rows, x.shape[1], dev), out_grid=gr)
        co = self.out_channels_
        cop = (co + 7) // 8 * 8
        wco = STORE.derived(self.conv_out.weight, "c3", lambda: _conv3_w(self.conv_out.weight, n_pad=cop))
        bco = STORE.derived(self.conv_out.bias, "pad", lambda: torch.cat([_bf(self.conv_out.bias), torch.zeros(cop - co, dtype=cd, device=dev)]))
        y = ops.gemm(pn, wco, bco, a_grid=gr, conv3x3=True)
        out = ops.unpatchify(y, I, co, H, W, 1).view(B, Tn, V, co, H, W)
        if squeeze:
            out = out.squeeze(2)
        if return_dict:
            return {"noise_pred": out}
        # the reference returns (result, maskgit_up, maskgit_down) whenever those lists are non-empty (:831-833); the
        # intermediate activations are not re-materialised in NCHW here
        return (out,), None, None


# ------------------------------------------------------------------------------------------ FLOP model (bench)
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
