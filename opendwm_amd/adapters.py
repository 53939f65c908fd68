"""MI355X-native layout condition adapter: drop-in for dwm.models.adapters.ImageAdapter
(src/dwm/models/adapters.py:6-60; diffusers T2I AdapterBlock / AdapterResnetBlock underneath).

State-dict keys equal the reference's: body.{i}.in_conv.*, body.{i}.resnets.{j}.block1.* /
block2.*, zero_convs.{i}.*, zero_gates.  All features are produced token-major
([I*h*w, C], the layout the MMDiT adds them in, crossview_temporal_dit.py:491-494); the 3x3
convolutions run as implicit GEMMs of dwm_gemm_bf16 over a zero-padded token grid."""
from __future__ import annotations

from typing import List, Optional

import torch
from torch import nn

from . import ops
from .blocks import STORE, _bf, stream32
from .ops import ACT_RELU, EPI_RESID, PaddedGrid

bf16 = torch.bfloat16


class AdapterResnetBlock(nn.Module):
    """x + block2(relu(block1(x))): Conv3x3(pad 1) -> ReLU -> Conv1x1 (diffusers)."""

    def __init__(self, channels: int):
        super().__init__()
        self.block1 = nn.Conv2d(channels, channels, kernel_size=3, padding=1)
        self.act = nn.ReLU()
        self.block2 = nn.Conv2d(channels, channels, kernel_size=1)

    def packed(self):
        """tap-major 3x3 weight [N, 9*C] and the 1x1 weight [N, C] in the store's precision (bf16, or fp32 for the accuracy
        path); rebuilt after every optimizer step / state-dict load (STORE.derived keys on the optimizer step and the
        parameter version, as every other packed weight does)"""
        w3, w1 = self.block1.weight, self.block2.weight
        return {"w3": STORE.derived(w3, "c3tap", lambda: _bf(w3).permute(0, 2, 3, 1).reshape(w3.shape[0], -1).contiguous()),
                "w1": STORE.derived(w1, "c1", lambda: _bf(w1).reshape(w1.shape[0], -1).contiguous())}

    def run_f32(self, x_pad: torch.Tensor, grid: PaddedGrid) -> torch.Tensor:
        """the fp32 accuracy path: x_pad fp32 padded grid, updated in place (dwm_gemm_f32 with the 3x3 taps)"""
        pk = self.packed()
        h1 = ops.gemm(x_pad, pk["w3"], _bf(self.block1.bias), act=ACT_RELU, a_grid=grid, conv3x3=True)
        ops.gemm(h1, pk["w1"], _bf(self.block2.bias), epilogue=EPI_RESID, res=x_pad, out=x_pad, c_grid=grid)
        return x_pad

    def run(self, x_pad: torch.Tensor, grid: PaddedGrid, x32_pad: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x_pad: padded token grid [grid.rows, C] (zero border), updated in place.  With x32_pad (the fp32 master of the same
        grid) the skip sum is taken and kept in fp32; x_pad receives its rounding - what the next convolution reads."""
        pk = self.packed()
        h1 = ops.gemm(x_pad, pk["w3"], _bf(self.block1.bias), act=ACT_RELU, a_grid=grid, conv3x3=True)
        if x32_pad is not None:
            ops.gemm(h1, pk["w1"], _bf(self.block2.bias), epilogue=EPI_RESID, res=x32_pad, out=x_pad, out32=x32_pad, c_grid=grid)
        else:
            ops.gemm(h1, pk["w1"], _bf(self.block2.bias), epilogue=EPI_RESID, res=x_pad, out=x_pad, c_grid=grid)
        return x_pad


class AdapterBlock(nn.Module):
    """diffusers AdapterBlock(in_channels, out_channels, num_res_blocks, down)."""

    def __init__(self, in_channels: int, out_channels: int, num_res_blocks: int, down: bool = False):
        super().__init__()
        self.downsample = nn.AvgPool2d(kernel_size=2, stride=2, ceil_mode=True) if down else None
        self.in_conv = nn.Conv2d(in_channels, out_channels, kernel_size=1) if in_channels != out_channels else None
        self.resnets = nn.Sequential(*[AdapterResnetBlock(out_channels) for _ in range(num_res_blocks)])
        self.out_channels = out_channels


class ImageAdapter(nn.Module):
    def __init__(
        self, in_channels: int = 3,
        channels: list = [320, 320, 640, 1280, 1280],
        is_downblocks: list = [False, True, True, True, False],
        num_res_blocks: int = 2, downscale_factor: int = 8,
        use_zero_convs: bool = False, zero_gate_coef: Optional[float] = None,
        gradient_checkpointing: bool = True
    ):
        super().__init__()
        self.downscale_factor = downscale_factor
        in_channels = in_channels * downscale_factor ** 2
        self.unshuffle = nn.PixelUnshuffle(downscale_factor)
        self.body = nn.ModuleList([
            AdapterBlock(in_channels if i == 0 else channels[i - 1], channels[i], num_res_blocks,
                         down=is_downblocks[i])
            for i in range(len(channels))])
        self.gradient_checkpointing = gradient_checkpointing
        self.zero_convs = nn.ModuleList([nn.Conv2d(c, c, 1) for c in channels]) if use_zero_convs \
            else [None for _ in channels]
        for z in self.zero_convs:
            if z is not None:
                nn.init.zeros_(z.weight)
                nn.init.zeros_(z.bias)
        self.zero_gate_coef = zero_gate_coef
        self.zero_gates = nn.Parameter(torch.zeros(len(channels))) if zero_gate_coef else None
        if any(c % 64 != 0 for c in channels):
            raise NotImplementedError("ImageAdapter channels must be multiples of 64 (GEMM K granularity)")

    def _levels(self, x: torch.Tensor, precise: bool):
        """the adapter body, level by level: yields (x16_pad, grid, zero_conv) when a level's resnets are done; x16_pad is
        the bf16 padded token grid of the level's output (the next level continues in it: consume it before advancing).

        precise: the running feature map - the skip path through all num_res_blocks x len(channels) resnets - is kept in
        fp32 next to its bf16 rounding (dwm_gemm_args.C32): the convolutions still read bf16, but the storage rounding of
        the skip sum no longer accumulates block after block.  The adapter's input does not change across denoise steps,
        so its error is the SAME at every step and adds up linearly over the 40 steps instead of in quadrature
        (profiles/r3_drift_bisect.json): it is the one place of the forward where bf16 storage shows in the final latents."""
        if self.zero_gates is not None:
            raise NotImplementedError("zero_gates (zero_gate_coef) is not used by any shipped CTSD config")
        x = x.flatten(0, -4).contiguous()
        if x.dtype not in (torch.float32, bf16):
            x = x.to(bf16)
        I, _, H, W = x.shape
        r = self.downscale_factor
        h, w = H // r, W // r
        cur = ops.unshuffle_tokens(x, r)                    # compact tokens [I*h*w, Cin padded to 64]
        cur_pad: Optional[torch.Tensor] = None
        cur32: Optional[torch.Tensor] = None
        grid: Optional[PaddedGrid] = None

        def conv1x1(a, wi, bias, **kw):
            """-> (bf16 padded grid, fp32 master or None)"""
            if not precise:
                return ops.gemm(a, wi, bias, c_grid=grid, **kw), None
            o16 = torch.zeros((grid.rows, wi.shape[0]), dtype=bf16, device=a.device)
            o32 = torch.zeros((grid.rows, wi.shape[0]), dtype=torch.float32, device=a.device)
            ops.gemm(a, wi, bias, epilogue=EPI_RESID, out=o16, out32=o32, c_grid=grid, **kw)
            return o16, o32

        for blk, zc in zip(self.body, self.zero_convs):
            if blk.downsample is not None:
                if cur is None:                             # leave the padded grid of the previous level
                    cur = cur_pad[grid.interior_index().to(cur_pad.device)]
                if h % 2 or w % 2:
                    raise NotImplementedError("AvgPool2d(ceil_mode) on odd sizes")
                cur = ops.avgpool2_tokens(cur, I, h, w)
                h, w = h // 2, w // 2
                cur_pad = None
            new_grid = PaddedGrid(I, h, w)
            if cur_pad is None:
                grid = new_grid
                if blk.in_conv is not None:
                    wi = _bf(blk.in_conv.weight).reshape(blk.in_conv.weight.shape[0], -1)
                    if wi.shape[1] != cur.shape[1]:          # K padded to a multiple of 64 by unshuffle_tokens
                        wp = torch.zeros((wi.shape[0], cur.shape[1]), dtype=bf16, device=wi.device)
                        wp[:, :wi.shape[1]] = wi
                        wi = wp
                    cur_pad, cur32 = conv1x1(cur, wi.contiguous(), _bf(blk.in_conv.bias))
                else:
                    cur_pad = torch.zeros((grid.rows, cur.shape[1]), dtype=bf16, device=cur.device)
                    cur_pad[grid.interior_index().to(cur.device)] = cur
                    cur32 = cur_pad.float() if precise else None
                cur = None
            elif blk.in_conv is not None:
                wi = _bf(blk.in_conv.weight).reshape(blk.in_conv.weight.shape[0], -1).contiguous()
                cur_pad, cur32 = conv1x1(cur_pad, wi, _bf(blk.in_conv.bias), a_grid=grid)
            for res in blk.resnets:
                res.run(cur_pad, grid, cur32)
            yield cur_pad, grid, zc

    def _levels_f32(self, x: torch.Tensor):
        """the fp32 accuracy path of the body (model.compute_dtype = torch.float32): every activation and weight in fp32, the
        convolutions by dwm_gemm_f32, PixelUnshuffle / AvgPool2d / the padded-grid scatter by the fp32 forms of the kernels the
        bf16 path uses (dwm_unshuffle_tokens_f32, dwm_avgpool2_tokens_f32, dwm_pad_tokens_f32)"""
        if self.zero_gates is not None:
            raise NotImplementedError("zero_gates (zero_gate_coef) is not used by any shipped CTSD config")
        f32 = torch.float32
        x = x.flatten(0, -4).to(f32).contiguous()
        I, _, H, W = x.shape
        r = self.downscale_factor
        h, w = H // r, W // r
        cur = ops.unshuffle_tokens(x, r, dtype=f32)          # compact tokens [I*h*w, Cin padded to 64]
        cur_pad, grid = None, None
        for blk, zc in zip(self.body, self.zero_convs):
            if blk.downsample is not None:
                if cur is None:                              # leave the padded grid of the previous level
                    cur = cur_pad[grid.interior_index().to(cur_pad.device)].contiguous()
                if h % 2 or w % 2:
                    raise NotImplementedError("AvgPool2d(ceil_mode) on odd sizes")
                cur = ops.avgpool2_tokens(cur, I, h, w)
                h, w = h // 2, w // 2
                cur_pad = None
            if cur_pad is None:
                grid = PaddedGrid(I, h, w)
                cur_pad = ops.pad_tokens(cur, grid)
                cur = None
            if blk.in_conv is not None:
                wi = _bf(blk.in_conv.weight).reshape(blk.in_conv.weight.shape[0], -1)
                if wi.shape[1] != cur_pad.shape[1]:          # input channels padded to a multiple of 64
                    wp = torch.zeros((wi.shape[0], cur_pad.shape[1]), dtype=f32, device=wi.device)
                    wp[:, :wi.shape[1]] = wi
                    wi = wp
                cur_pad = ops.gemm(cur_pad, wi.contiguous(), _bf(blk.in_conv.bias), a_grid=grid, c_grid=grid)
            for res in blk.resnets:
                res.run_f32(cur_pad, grid)
            yield cur_pad, grid, zc

    @torch.no_grad()
    def run(self, x: torch.Tensor, precise: bool = False) -> List[torch.Tensor]:
        """x [..., C, H, W] -> list of token-major features [I*h_i*w_i, channels[i]], I = prod(leading): bf16, or - precise -
        fp32 (fp32 skip path inside the adapter, fp32 output of the zero convolutions: what the inference forwards cache
        across denoise steps and add with ops.add_)."""
        feats = []
        if STORE.precision == torch.float32:                 # the fp32 accuracy path: fp32 residuals
            for cur_pad, grid, zc in self._levels_f32(x):
                if zc is not None:
                    feats.append(ops.gemm(cur_pad, _bf(zc.weight).reshape(zc.weight.shape[0], -1).contiguous(), _bf(zc.bias), a_grid=grid))
                else:
                    feats.append(cur_pad[grid.interior_index().to(cur_pad.device)].contiguous())
            return feats
        for cur_pad, grid, zc in self._levels(x, precise):
            if zc is not None:
                wz = _bf(zc.weight).reshape(zc.weight.shape[0], -1).contiguous()
                if precise:
                    o16 = torch.empty((grid.pixels, wz.shape[0]), dtype=bf16, device=cur_pad.device)
                    o32 = torch.empty((grid.pixels, wz.shape[0]), dtype=torch.float32, device=cur_pad.device)
                    ops.gemm(cur_pad, wz, _bf(zc.bias), a_grid=grid, epilogue=EPI_RESID, out=o16, out32=o32)
                    feats.append(o32)
                else:
                    feats.append(ops.gemm(cur_pad, wz, _bf(zc.bias), a_grid=grid))
            else:
                f = cur_pad[grid.interior_index().to(cur_pad.device)]
                feats.append(f.float() if precise else f)
        return feats

    @torch.no_grad()
    def residual_adders(self, x: torch.Tensor):
        """generator for forwards that recompute the adapter at every call: the i-th item is a function add(h) that adds the
        i-th feature to the token-major hidden state h [I*h_i*w_i, C] IN PLACE - the zero convolution runs as a GEMM whose
        epilogue adds h in fp32 (h <- bf16(h + W x + b): no residual tensor, no separate add, one rounding).  Each add must
        be called before the next item is drawn (the next level overwrites the grid the zero convolution reads)."""
        for cur_pad, grid, zc in (self._levels_f32(x) if STORE.precision == torch.float32 else self._levels(x, True)):
            if zc is not None:
                wz = _bf(zc.weight).reshape(zc.weight.shape[0], -1).contiguous()

                def add(h, cur_pad=cur_pad, grid=grid, wz=wz, zc=zc):
                    if h.shape != (grid.pixels, wz.shape[0]):
                        raise RuntimeError(f"condition residual {(grid.pixels, wz.shape[0])} does not match hidden states {tuple(h.shape)}")
                    if stream32(h):        # the fp32 hidden stream of the bf16 forward: fp32 in, fp32 out, no bf16 copy
                        ops.gemm(cur_pad, wz, _bf(zc.bias), a_grid=grid, epilogue=EPI_RESID, res=h, out32=h, mirror=False)
                    else:
                        ops.gemm(cur_pad, wz, _bf(zc.bias), a_grid=grid, epilogue=EPI_RESID, res=h, out=h)
            else:
                def add(h, cur_pad=cur_pad, grid=grid):
                    ops.add_(h, cur_pad[grid.interior_index().to(cur_pad.device)].contiguous())      # (any bf16 / fp32 pairing)
            yield add

    def forward(self, x: torch.Tensor, return_features: bool = False):
        """Reference signature (adapters.py:40): features shaped [*base_shape, C, h, w]."""
        base_shape = x.shape[:-3]
        r = self.downscale_factor
        h, w = x.shape[-2] // r, x.shape[-1] // r
        out = []
        for blk, f in zip(self.body, self.run(x)):
            if blk.downsample is not None:
                h, w = h // 2, w // 2
            out.append(f.view(*base_shape, h, w, f.shape[-1]).movedim(-1, -3))
        return out if not return_features else out[-1]
