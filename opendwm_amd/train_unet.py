"""Training path of the SD 2.1 cross-view temporal UNet - the `UNetSpatioTemporalConditionModel` branch of
`CrossviewTemporalSD.train_step` (src/dwm/pipelines/ctsd.py:1240-1253: DDPM add_noise, epsilon / v_prediction target;
forward under autocast, `loss.backward()` :1401-1404).  Same design as opendwm_amd.train (the MMDiT branch):

  * one `torch.autograd.Function` per residual block / transformer sub-block / sampler.  The forward is the fused inference
    path of opendwm_amd.unet and keeps only the block inputs (gradient checkpointing, as the reference's blocks do,
    crossview_temporal_unet.py:84-93); the backward recomputes what it needs and walks the block in reverse with HIP kernels:
      - 3x3 / stride-2 / (3,1,1) convolutions: weight gradient = one GEMM per tap over the tap-shifted rows of the padded
        input grid, input gradient = the same implicit GEMM with mirrored taps over the padded (or zero-stuffed) output
        gradient;
      - GroupNorm (+SiLU): dwm_groupnorm_bwd, also with the (b v) x (t h w) row map of TemporalResnetBlock;
      - BasicTransformerBlock: LayerNorm / flash self-attention / text cross-attention (dwm_attention_bwd in cross mode) /
        GEGLU backward kernels; the cross-view / temporal blocks + AlphaBlender are opendwm_amd.train's VTBlockFn.
  * parameters are Function inputs (DDP sees each block's gradients as soon as the block is done); torch orchestrates, all
    arithmetic on token-sized tensors is HIP.  Row gathers of padded grids (`x_pad[idx]`) are index copies.
"""
from __future__ import annotations

from typing import List

import torch

from . import ops
from . import train_ops as T
from .blocks import STORE, _bf
from .ops import EPI_RESID, PaddedGrid, TimeGrid
from .train import (AdapterFn, AddFn, Grads, SiluFn, _conv3_flip, _grads_for, _params, alpha_train, lin_bwd, lin_fwd, mlp_train,
                    project_qkv_train, qkv_bwd, vt_block_train)
from . import unet as UM

bf16 = torch.bfloat16


# ------------------------------------------------------------------------------------------ convolution helpers
def _interior(grid: PaddedGrid, dev) -> torch.Tensor:
    return grid.interior_index().to(dev)


def _time_interior(tg: TimeGrid, dev) -> torch.Tensor:
    """padded-row index of every compact row (b, t, r) of a TimeGrid: b (T+2) vn + (t+1) vn + r"""
    b = torch.arange(tg.B, device=dev)[:, None, None]
    t = torch.arange(tg.T, device=dev)[None, :, None]
    r = torch.arange(tg.vn, device=dev)[None, None, :]
    return (b * (tg.T + 2) * tg.vn + (t + 1) * tg.vn + r).reshape(-1)


def conv_wgrad(dy: torch.Tensor, x_pad: torch.Tensor, idx: torch.Tensor, shifts) -> torch.Tensor:
    """dW [N, taps*C] (tap-major, bf16): dW[n, t, c] = sum_pixels dy[pixel, n] * x_pad[idx[pixel] + shift_t, c]
    (train_ops.conv_wgrad: one dwm_gemm_tn launch for all taps)."""
    return T.conv_wgrad(dy, x_pad, idx, shifts)


def _conv3_w_to_param(dw: torch.Tensor, n: int, c: int) -> torch.Tensor:
    """tap-major [Np, 9*Cp] -> [n, c, 3, 3]"""
    cp = dw.shape[1] // 9
    return dw.view(dw.shape[0], 3, 3, cp)[:n, :, :, :c].permute(0, 3, 1, 2)


def _conv3d_flip(w: torch.Tensor) -> torch.Tensor:
    """input-gradient weight of a Conv3d (3,1,1): [N, C, 3, 1, 1] -> tap-major [C, 3*N] with the taps mirrored"""
    return STORE.derived(w, "c3dflip", lambda: _bf(w).reshape(w.shape[0], w.shape[1], 3).flip(2).permute(1, 2, 0)
                         .reshape(w.shape[1], -1).contiguous())


def _conv3_flip_pad(w: torch.Tensor, n_pad: int) -> torch.Tensor:
    """_conv3_flip with the OUTPUT channels (the contraction of the input-gradient GEMM) zero-padded to n_pad"""
    def make():
        wb = _bf(w)
        n, c = wb.shape[:2]
        t = torch.zeros((c, 3, 3, n_pad), dtype=bf16, device=wb.device)
        t[..., :n] = wb.flip(2, 3).permute(1, 2, 3, 0)
        return t.reshape(c, -1).contiguous()
    return STORE.derived(w, f"c3flip{n_pad}", make)


def _ln_bwd(G: Grads, x, n, dyy, dx):
    D = x.shape[1]
    dg = torch.zeros(1, D, dtype=torch.float32, device=x.device)
    db = torch.zeros(1, D, dtype=torch.float32, device=x.device)
    T.layernorm_bwd(x, dyy, eps=1e-5, dx=dx, accumulate=True, weight=_bf(n.weight), dgamma=dg, dbeta=db)
    G.add(n.weight, dg[0])
    G.add(n.bias, db[0])


def _gn_bwd(G: Grads, norm, x, dz, I, P, eps, **kw):
    Cc = x.shape[1]
    dg = torch.zeros(Cc, dtype=torch.float32, device=x.device)
    db = torch.zeros(Cc, dtype=torch.float32, device=x.device)
    dx = T.groupnorm_bwd(x, dz, I, P, _bf(norm.weight), _bf(norm.bias), 32, eps, dg, db, **kw)
    G.add(norm.weight, dg)
    G.add(norm.bias, db)
    return dx


# ------------------------------------------------------------------------------------------ ResBlock
def resnet2d_backward(G: Grads, rb: UM.ResnetBlock2D, x: torch.Tensor, tp: torch.Tensor, g: UM._Geom, dout: torch.Tensor):
    """ResnetBlock2D.run backwards.  x [I*N, Ci] block input, tp [I, Co] = time_emb_proj(silu(emb)); dout [I*N, Co].
    Returns (dx, dtp fp32 [I, Co])."""
    grid = PaddedGrid(g.I, g.h, g.w)
    idx = _interior(grid, x.device)
    w1 = STORE.derived(rb.conv1.weight, "c3", lambda: UM._conv3_w(rb.conv1.weight))
    Ci, Co = x.shape[1], dout.shape[1]
    # recompute
    p1 = ops.groupnorm_silu(x, g.I, g.N, _bf(rb.norm1.weight), _bf(rb.norm1.bias), 32, rb.eps, out_grid=grid)
    h1 = ops.gemm(p1, w1, _bf(rb.conv1.bias), a_grid=grid, conv3x3=True, epilogue=EPI_RESID, res=tp, res_mod=-g.N)
    p2 = ops.groupnorm_silu(h1, g.I, g.N, _bf(rb.norm2.weight), _bf(rb.norm2.bias), 32, rb.eps, out_grid=grid)
    # conv2 (+ shortcut)
    G.add(rb.conv2.bias, T.segsum(dout)[0])
    G.add(rb.conv2.weight, _conv3_w_to_param(conv_wgrad(dout, p2, idx, grid.tap_shifts()), Co, Co))
    dp2 = ops.gemm(ops.pad_tokens(dout, grid), _conv3_flip(rb.conv2.weight), None, a_grid=grid, conv3x3=True)
    del p2
    dh1 = _gn_bwd(G, rb.norm2, h1, dp2, g.I, g.N, rb.eps)
    del dp2, h1
    # conv1 + per-image time embedding row
    G.add(rb.conv1.bias, T.segsum(dh1)[0])
    dtp = T.segsum(dh1, rows_per_group=g.N)
    G.add(rb.conv1.weight, _conv3_w_to_param(conv_wgrad(dh1, p1, idx, grid.tap_shifts()), Co, Ci))
    dp1 = ops.gemm(ops.pad_tokens(dh1, grid), _conv3_flip(rb.conv1.weight), None, a_grid=grid, conv3x3=True)
    del p1, dh1
    dx = _gn_bwd(G, rb.norm1, x, dp1, g.I, g.N, rb.eps)
    if rb.conv_shortcut is not None:
        cs = rb.conv_shortcut
        dws, dbs = T.linear_wgrad(dout, x, want_bias=True)
        G.add(cs.weight, dws)
        G.add(cs.bias, dbs)
        wst = STORE.derived(cs.weight, "T1", lambda: T.transpose(_bf(cs.weight).reshape(cs.weight.shape[0], -1), rows_pad=cs.weight.shape[0]))
        dx = T.linear_dgrad(dout, wst, epilogue=EPI_RESID, res=dx)
    else:
        dx = T.rowcombine(dx, b=dout)
    return dx, dtp


def temporal_resnet_backward(G: Grads, tb: UM.TemporalResnetBlock, s: torch.Tensor, tp: torch.Tensor, g: UM._Geom,
                             alpha: torch.Tensor, dout: torch.Tensor):
    """TemporalResnetBlock.run (+ AlphaBlender) backwards: out = alpha s + (1 - alpha) (s + temporal_resnet(s)).
    Returns (ds, dtp fp32 [I, C], dalpha fp32 [B])."""
    tg = TimeGrid(g.B, g.T, g.V * g.N)
    imap = (g.V, g.N, g.T * g.V * g.N, g.N, g.V * g.N)
    idx = _time_interior(tg, s.device)
    shifts = tg.tap_shifts()
    Cc = s.shape[1]
    IB, PB = g.B * g.V, g.T * g.N
    rpa = g.T * g.V * g.N
    w1 = STORE.derived(tb.conv1.weight, "c3d", lambda: UM._conv3d_w(tb.conv1.weight))
    w2 = STORE.derived(tb.conv2.weight, "c3d", lambda: UM._conv3d_w(tb.conv2.weight))
    # recompute
    t1 = ops.groupnorm_silu(s, IB, PB, _bf(tb.norm1.weight), _bf(tb.norm1.bias), 32, tb.eps, out_grid=tg, img_map=imap)
    u1 = ops.gemm(t1, w1, _bf(tb.conv1.bias), a_grid=tg, conv_taps=shifts, epilogue=EPI_RESID, res=tp, res_mod=-g.N)
    t2 = ops.groupnorm_silu(u1, IB, PB, _bf(tb.norm2.weight), _bf(tb.norm2.bias), 32, tb.eps, out_grid=tg, img_map=imap)
    y = ops.gemm(t2, w2, _bf(tb.conv2.bias), a_grid=tg, conv_taps=shifts, epilogue=EPI_RESID, res=s)
    # mixer: out = alpha s + (1 - alpha) y
    dalpha = T.segsum_diff(dout, s, y, rows_per_group=rpa).double().sum(-1).float()
    del y
    one_minus = (1.0 - alpha).contiguous()
    dy = T.rowcombine(dout, coef_a=one_minus, rows_per_coef_a=rpa)
    ds = dout.clone()               # alpha dout + dy (the residual `+ s` inside y) = dout
    # conv2
    G.add(tb.conv2.bias, T.segsum(dy)[0])
    G.add(tb.conv2.weight, conv_wgrad(dy, t2, idx, shifts).view(Cc, 3, Cc).permute(0, 2, 1).reshape(tb.conv2.weight.shape))
    dt2 = ops.gemm(ops.pad_tokens(dy, tg), _conv3d_flip(tb.conv2.weight), None, a_grid=tg, conv_taps=shifts)
    del t2, dy
    du1 = _gn_bwd(G, tb.norm2, u1, dt2, IB, PB, tb.eps, img_map=imap)
    del dt2, u1
    G.add(tb.conv1.bias, T.segsum(du1)[0])
    dtp = T.segsum(du1, rows_per_group=g.N)
    G.add(tb.conv1.weight, conv_wgrad(du1, t1, idx, shifts).view(Cc, 3, Cc).permute(0, 2, 1).reshape(tb.conv1.weight.shape))
    dt1 = ops.gemm(ops.pad_tokens(du1, tg), _conv3d_flip(tb.conv1.weight), None, a_grid=tg, conv_taps=shifts)
    del t1, du1
    dg = torch.zeros(Cc, dtype=torch.float32, device=s.device)
    db = torch.zeros(Cc, dtype=torch.float32, device=s.device)
    T.groupnorm_bwd(s, dt1, IB, PB, _bf(tb.norm1.weight), _bf(tb.norm1.bias), 32, tb.eps, dg, db, dx=ds, accumulate=True,
                    img_map=imap)
    G.add(tb.norm1.weight, dg)
    G.add(tb.norm1.bias, db)
    return ds, dtp, dalpha


class ResBlockFn(torch.autograd.Function):
    """dwm.models.crossview_temporal.ResBlock (:75-164): ResnetBlock2D [+ TemporalResnetBlock, AlphaBlender].
    Gradient-carrying inputs: x, silu_emb [I, E], alpha [B]."""

    @staticmethod
    def forward(ctx, rb, geom, x, silu_emb, alpha, *params):
        ctx.rb, ctx.geom = rb, geom
        ctx.save_for_backward(x, silu_emb, alpha)
        with torch.no_grad():
            sp = rb.spatial_res_block
            tps = lin_fwd(silu_emb, sp.time_emb_proj)
            s = _resnet2d_fwd(sp, x, tps, geom)
            if rb.temporal_res_block is None:
                return s
            tb = rb.temporal_res_block
            return _temporal_fwd(tb, s, lin_fwd(silu_emb, tb.time_emb_proj), geom, alpha)

    @staticmethod
    def backward(ctx, dout):
        x, silu_emb, alpha = ctx.saved_tensors
        rb, g = ctx.rb, ctx.geom
        G = Grads()
        dout = dout.contiguous()
        sp, tb = rb.spatial_res_block, rb.temporal_res_block
        tps = lin_fwd(silu_emb, sp.time_emb_proj)
        dsilu = None
        dalpha = None
        if tb is not None:
            s = _resnet2d_fwd(sp, x, tps, g)
            tpt = lin_fwd(silu_emb, tb.time_emb_proj)
            dout, dtpt, dalpha = temporal_resnet_backward(G, tb, s, tpt, g, alpha, dout)
            del s
            dsilu = lin_bwd(G, tb.time_emb_proj, silu_emb, ops.cast_bf16(dtpt))
        dx, dtps = resnet2d_backward(G, sp, x, tps, g, dout)
        d2 = lin_bwd(G, sp.time_emb_proj, silu_emb, ops.cast_bf16(dtps))
        dsilu = d2 if dsilu is None else T.rowcombine(dsilu, b=d2)
        return (None, None, dx, dsilu, None if dalpha is None else dalpha.to(alpha.dtype)) + \
            _grads_for(G, _params(rb), ctx.needs_input_grad[5:])


def _resnet2d_fwd(rb: UM.ResnetBlock2D, x, tp, g: UM._Geom):
    """ResnetBlock2D.run with the time-embedding row given (no stacked projection) and fresh buffers"""
    grid = PaddedGrid(g.I, g.h, g.w)
    w1 = STORE.derived(rb.conv1.weight, "c3", lambda: UM._conv3_w(rb.conv1.weight))
    w2 = STORE.derived(rb.conv2.weight, "c3", lambda: UM._conv3_w(rb.conv2.weight))
    p1 = ops.groupnorm_silu(x, g.I, g.N, _bf(rb.norm1.weight), _bf(rb.norm1.bias), 32, rb.eps, out_grid=grid)
    h1 = ops.gemm(p1, w1, _bf(rb.conv1.bias), a_grid=grid, conv3x3=True, epilogue=EPI_RESID, res=tp, res_mod=-g.N)
    p2 = ops.groupnorm_silu(h1, g.I, g.N, _bf(rb.norm2.weight), _bf(rb.norm2.bias), 32, rb.eps, out=p1 if p1.shape[1] == h1.shape[1] else None,
                            out_grid=grid)
    if rb.conv_shortcut is not None:
        cs = rb.conv_shortcut
        ws = STORE.derived(cs.weight, "c1", lambda: _bf(cs.weight).reshape(cs.weight.shape[0], -1).contiguous())
        x = ops.gemm(x, ws, _bf(cs.bias))
    return ops.gemm(p2, w2, _bf(rb.conv2.bias), a_grid=grid, conv3x3=True, epilogue=EPI_RESID, res=x, out=h1)


def _temporal_fwd(tb: UM.TemporalResnetBlock, s, tp, g: UM._Geom, alpha):
    tg = TimeGrid(g.B, g.T, g.V * g.N)
    imap = (g.V, g.N, g.T * g.V * g.N, g.N, g.V * g.N)
    w1 = STORE.derived(tb.conv1.weight, "c3d", lambda: UM._conv3d_w(tb.conv1.weight))
    w2 = STORE.derived(tb.conv2.weight, "c3d", lambda: UM._conv3d_w(tb.conv2.weight))
    t1 = ops.groupnorm_silu(s, g.B * g.V, g.T * g.N, _bf(tb.norm1.weight), _bf(tb.norm1.bias), 32, tb.eps, out_grid=tg, img_map=imap)
    u1 = ops.gemm(t1, w1, _bf(tb.conv1.bias), a_grid=tg, conv_taps=tg.tap_shifts(), epilogue=EPI_RESID, res=tp, res_mod=-g.N)
    t2 = ops.groupnorm_silu(u1, g.B * g.V, g.T * g.N, _bf(tb.norm2.weight), _bf(tb.norm2.bias), 32, tb.eps, out=t1, out_grid=tg, img_map=imap)
    return ops.gemm(t2, w2, _bf(tb.conv2.bias), a_grid=tg, conv_taps=tg.tap_shifts(), epilogue=EPI_RESID, res=s, blend=s,
                    alpha=alpha, rows_per_alpha=g.T * g.V * g.N, out=u1)


def res_block_train(rb: UM.ResBlock, x, silu_emb, g: UM._Geom, disable_temporal):
    if rb.temporal_res_block is not None:
        alpha = alpha_train(rb.time_mixer, disable_temporal, g.B)
    else:
        alpha = torch.zeros(g.B, dtype=torch.float32, device=x.device)
    return ResBlockFn.apply(rb, g, x, silu_emb, alpha, *_params(rb))


# ------------------------------------------------------------------------------------------ BasicTransformerBlock
def basic_block_backward(G: Grads, blk: UM.BasicTransformerBlock, h: torch.Tensor, ctx_rows: torch.Tensor, n_img: int,
                         dout: torch.Tensor) -> torch.Tensor:
    """BasicTransformerBlock.run backwards (diffusers BasicTransformerBlock: norm1 -> self-attention -> norm2 -> text
    cross-attention -> norm3 -> GEGLU feed-forward, each with a residual).  h [I*N, D] block input, ctx_rows [I*L, Dc]
    the text tokens (no gradient).  Returns dh."""
    D = blk.dim
    N = h.shape[0] // n_img
    Lk = ctx_rows.shape[0] // n_img
    dev = h.device
    ln = lambda x, n: ops.layernorm(x, eps=1e-5, weight=_bf(n.weight), bias=_bf(n.bias))
    id_map = ops.rowmap_identity(n_img, N)
    a1, a2 = blk.attn1, blk.attn2
    # ---------------- recompute
    y1 = ln(h, blk.norm1)
    qkv, _ = project_qkv_train(a1, y1)
    ao = torch.empty_like(h)
    lse = torch.empty(n_img * blk.heads * N, dtype=torch.float32, device=dev)
    ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], ao, id_map, blk.heads, lse=lse)
    o1 = a1.to_out[0]
    h1 = lin_fwd(ao, o1, epilogue=EPI_RESID, res=h)
    y2 = ln(h1, blk.norm2)
    q2 = ops.gemm(y2, _bf(a2.to_q.weight))
    kv = ops.gemm(ctx_rows, a2.wkv())
    ao2 = torch.empty_like(h)
    lse2 = torch.zeros(n_img * blk.heads * (N + Lk), dtype=torch.float32, device=dev)
    ops.cross_attention(q2, kv[:, :D], kv[:, D:], ao2, n_img, blk.heads, lse=lse2)
    o2 = a2.to_out[0]
    h2 = lin_fwd(ao2, o2, epilogue=EPI_RESID, res=h1)
    y3 = ln(h2, blk.norm3)
    p, l2 = blk.ff.net[0].proj, blk.ff.net[2]
    u = lin_fwd(y3, p)
    gg = T.geglu_fwd(u)
    # ---------------- out = h2 + ff(norm3(h2))
    dh2 = dout
    dgg = lin_bwd(G, l2, gg, dout)
    du = T.geglu_bwd(u, dgg)
    del gg, dgg, u
    dy3 = lin_bwd(G, p, y3, du)
    del du
    _ln_bwd(G, h2, blk.norm3, dy3, dh2)
    # ---------------- h2 = h1 + to_out(cross_attention(norm2(h1), text))
    dao2 = lin_bwd(G, o2, ao2, dh2)
    dq2 = torch.empty_like(q2)
    dkv = torch.empty_like(kv)
    ops.cross_attention_bwd(q2, kv[:, :D], kv[:, D:], ao2, dao2, dq2, dkv[:, :D], dkv[:, D:], n_img, blk.heads, lse2)
    if a2.to_k.weight.requires_grad:
        dwkv, _ = T.linear_wgrad(dkv, ctx_rows, want_bias=False)
        G.add(a2.to_k.weight, dwkv[:D])
        G.add(a2.to_v.weight, dwkv[D:])
    dy2 = lin_bwd(G, a2.to_q, y2, dq2)
    dh1 = dh2
    _ln_bwd(G, h1, blk.norm2, dy2, dh1)
    # ---------------- h1 = h + to_out(self_attention(norm1(h)))
    dao = lin_bwd(G, o1, ao, dh1)
    dqkv = torch.empty_like(qkv)
    ops.attention_bwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], ao, dao, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:],
                      id_map, blk.heads, lse)
    dy1 = qkv_bwd(G, a1, y1, dqkv)
    dh = dh1
    _ln_bwd(G, h, blk.norm1, dy1, dh)
    return dh


class BasicBlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, blk, n_img, ctx_rows, h, *params):
        ctx.blk, ctx.n_img, ctx.ctx_rows = blk, n_img, ctx_rows
        ctx.save_for_backward(h)
        with torch.no_grad():
            return blk.run(h.clone(), UM._TextContext.from_rows(ctx_rows), n_img)

    @staticmethod
    def backward(ctx, dout):
        (h,) = ctx.saved_tensors
        G = Grads()
        dh = basic_block_backward(G, ctx.blk, h, ctx.ctx_rows, ctx.n_img, dout.contiguous().clone())
        return (None, None, None, dh) + _grads_for(G, _params(ctx.blk), ctx.needs_input_grad[4:])


# ------------------------------------------------------------------------------------------ TransformerModel ends
class TMInFn(torch.autograd.Function):
    """TransformerModel head: GroupNorm(32, eps 1e-6) -> proj_in"""

    @staticmethod
    def forward(ctx, tm, geom, x, *params):
        ctx.tm, ctx.geom = tm, geom
        ctx.save_for_backward(x)
        g = geom
        with torch.no_grad():
            hn = ops.groupnorm_silu(x, g.I, g.N, _bf(tm.norm.weight), _bf(tm.norm.bias), 32, 1e-6, silu=False)
            return lin_fwd(hn, tm.proj_in)

    @staticmethod
    def backward(ctx, dh):
        (x,) = ctx.saved_tensors
        tm, g = ctx.tm, ctx.geom
        G = Grads()
        hn = ops.groupnorm_silu(x, g.I, g.N, _bf(tm.norm.weight), _bf(tm.norm.bias), 32, 1e-6, silu=False)
        dhn = lin_bwd(G, tm.proj_in, hn, dh.contiguous())
        dx = _gn_bwd(G, tm.norm, x, dhn, g.I, g.N, 1e-6, silu=False)
        ps = _params(tm.norm) + _params(tm.proj_in)
        return (None, None, dx) + _grads_for(G, ps, ctx.needs_input_grad[3:])


class TMOutFn(torch.autograd.Function):
    """TransformerModel tail: proj_out(h) + x"""

    @staticmethod
    def forward(ctx, tm, h, x, *params):
        ctx.tm = tm
        ctx.save_for_backward(h)
        with torch.no_grad():
            return lin_fwd(h, tm.proj_out, epilogue=EPI_RESID, res=x)

    @staticmethod
    def backward(ctx, dout):
        (h,) = ctx.saved_tensors
        G = Grads()
        dout = dout.contiguous()
        dh = lin_bwd(G, ctx.tm.proj_out, h, dout)
        return (None, dh, dout) + _grads_for(G, _params(ctx.tm.proj_out), ctx.needs_input_grad[3:])


def transformer_model_train(tm: UM.TransformerModel, x, ctx_rows, g: UM._Geom, disable_crossview, disable_temporal, mask):
    B, Tn, V, Cc = g.B, g.T, g.V, tm.in_channels
    dev = x.device
    h = TMInFn.apply(tm, g, x, *(_params(tm.norm) + _params(tm.proj_in)))
    view_emb = seq_emb = None
    if tm.view_pos_embed is not None:
        idx = torch.arange(V, device=dev).view(1, 1, V).expand(B, Tn, V)
        view_emb = mlp_train(tm.view_pos_embed, ops.timestep_sinusoid(idx, Cc))
        alpha_v = alpha_train(tm.view_mixer, disable_crossview, B)
    if tm.time_pos_embed is not None:
        idx = torch.arange(Tn, device=dev).view(1, Tn, 1).expand(B, Tn, V)
        seq_emb = mlp_train(tm.time_pos_embed, ops.timestep_sinusoid(idx, Cc))
        alpha_t = alpha_train(tm.time_mixer, disable_temporal, B)
    for l, blk in enumerate(tm.transformer_blocks):
        h = BasicBlockFn.apply(blk, g.I, ctx_rows, h, *_params(blk))
        if tm.view_pos_embed is not None:
            rm = ops.rowmap_crossview_rowwise(B, Tn, V, g.h, g.w) if tm.rowwise_cv else ops.rowmap_crossview_pointwise(B, Tn, V, g.h, g.w)
            h = vt_block_train(tm.crossview_transformer_blocks[l], h, rm, view_emb, g.N, alpha_v, Tn * V * g.N, group_mask=mask)
        if tm.time_pos_embed is not None:
            rm = ops.rowmap_temporal_rowwise(B, Tn, V, g.h, g.w) if tm.rowwise_t else ops.rowmap_temporal_pointwise(B, Tn, V, g.h, g.w)
            h = vt_block_train(tm.temporal_transformer_blocks[l], h, rm, seq_emb, g.N, alpha_t, Tn * V * g.N)
    return TMOutFn.apply(tm, h, x, *_params(tm.proj_out))


# ------------------------------------------------------------------------------------------ samplers, stem, head
class DownsampleFn(torch.autograd.Function):
    """Downsample2D(padding=1): 3x3 stride-2 convolution with symmetric padding"""

    @staticmethod
    def forward(ctx, conv, I, h, w, x, *params):
        ctx.conv, ctx.dims = conv, (I, h, w)
        ctx.save_for_backward(x)
        with torch.no_grad():
            gr = PaddedGrid(I, h, w)
            wd = STORE.derived(conv.weight, "c3", lambda: UM._conv3_w(conv.weight))
            return ops.gemm(ops.pad_tokens(x, gr), wd, _bf(conv.bias), a_grid=gr, conv3x3=True, stride2="sym")

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        conv = ctx.conv
        I, h, w = ctx.dims
        dev = x.device
        G = Grads()
        dy = dy.contiguous()
        gr = PaddedGrid(I, h, w)
        Cc = x.shape[1]
        # output pixel (i, y, x) reads padded rows (2y + dy, 2x + dx): base row + tap shift
        i_ = torch.arange(I, device=dev)[:, None, None]
        y_ = torch.arange(h // 2, device=dev)[None, :, None]
        x_ = torch.arange(w // 2, device=dev)[None, None, :]
        base = (i_ * (h + 2) * (w + 2) + 2 * y_ * (w + 2) + 2 * x_).reshape(-1)
        xp = ops.pad_tokens(x, gr)
        G.add(conv.bias, T.segsum(dy)[0])
        G.add(conv.weight, _conv3_w_to_param(conv_wgrad(dy, xp, base, gr.tap_shifts_stride2()), Cc, Cc))
        del xp
        # input gradient: dy zero-stuffed onto the even interior pixels of the input grid, then the mirrored 3x3 convolution
        z = torch.zeros((gr.rows, dy.shape[1]), dtype=bf16, device=dev)
        z[base + (w + 2) + 1] = dy
        dx = ops.gemm(z, _conv3_flip(conv.weight), None, a_grid=gr, conv3x3=True)
        return (None, None, None, None, dx) + _grads_for(G, _params(conv), ctx.needs_input_grad[5:])


class UpsampleFn(torch.autograd.Function):
    """Upsample2D: nearest 2x, then a 3x3 convolution"""

    @staticmethod
    def forward(ctx, conv, I, h, w, x, *params):
        ctx.conv, ctx.dims = conv, (I, h, w)
        ctx.save_for_backward(x)
        with torch.no_grad():
            gr = PaddedGrid(I, 2 * h, 2 * w)
            wu = STORE.derived(conv.weight, "c3", lambda: UM._conv3_w(conv.weight))
            return ops.gemm(ops.upsample2_padded(x, I, h, w), wu, _bf(conv.bias), a_grid=gr, conv3x3=True)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        conv = ctx.conv
        I, h, w = ctx.dims
        G = Grads()
        dy = dy.contiguous()
        gr = PaddedGrid(I, 2 * h, 2 * w)
        Cc = x.shape[1]
        up = ops.upsample2_padded(x, I, h, w)
        G.add(conv.bias, T.segsum(dy)[0])
        G.add(conv.weight, _conv3_w_to_param(conv_wgrad(dy, up, _interior(gr, x.device), gr.tap_shifts()), Cc, Cc))
        del up
        dup = ops.gemm(ops.pad_tokens(dy, gr), _conv3_flip(conv.weight), None, a_grid=gr, conv3x3=True)
        four = torch.full((1,), 4.0, dtype=torch.float32, device=x.device)
        pooled = ops.avgpool2_tokens(dup, I, 2 * h, 2 * w)                       # mean of the 2x2 children; the sum is wanted
        dx = T.rowcombine(pooled, coef_a=four, rows_per_coef_a=pooled.shape[0])
        return (None, None, None, None, dx) + _grads_for(G, _params(conv), ctx.needs_input_grad[5:])


class ConvInFn(torch.autograd.Function):
    """conv_in on the NCHW latents (no gradient to the latents)"""

    @staticmethod
    def forward(ctx, conv, xin, *params):
        ctx.conv = conv
        ctx.save_for_backward(xin)
        I, _, H, W = xin.shape
        with torch.no_grad():
            grid = PaddedGrid(I, H, W)
            pin = ops.pad_tokens(ops.unshuffle_tokens(xin, 1, 64), grid)
            wci = STORE.derived(conv.weight, "c3", lambda: UM._conv3_w(conv.weight, c_pad=64))
            return ops.gemm(pin, wci, _bf(conv.bias), a_grid=grid, conv3x3=True)

    @staticmethod
    def backward(ctx, dy):
        (xin,) = ctx.saved_tensors
        conv = ctx.conv
        I, Ci, H, W = xin.shape
        G = Grads()
        dy = dy.contiguous()
        grid = PaddedGrid(I, H, W)
        pin = ops.pad_tokens(ops.unshuffle_tokens(xin, 1, 64), grid)
        G.add(conv.bias, T.segsum(dy)[0])
        G.add(conv.weight, _conv3_w_to_param(conv_wgrad(dy, pin, _interior(grid, dy.device), grid.tap_shifts()), dy.shape[1], Ci))
        return (None, None) + _grads_for(G, _params(conv), ctx.needs_input_grad[2:])


class HeadFn(torch.autograd.Function):
    """conv_norm_out (GroupNorm + SiLU) -> conv_out -> NCHW"""

    @staticmethod
    def forward(ctx, model, dims, x, *params):
        ctx.model, ctx.dims = model, dims
        ctx.save_for_backward(x)
        I, H, W = dims
        with torch.no_grad():
            gr = PaddedGrid(I, H, W)
            nrm, conv = model.conv_norm_out, model.conv_out
            pn = ops.groupnorm_silu(x, I, H * W, _bf(nrm.weight), _bf(nrm.bias), 32, 1e-5, out_grid=gr)
            co = model.out_channels_
            cop = (co + 7) // 8 * 8
            wco = STORE.derived(conv.weight, "c3", lambda: UM._conv3_w(conv.weight, n_pad=cop))
            bco = STORE.derived(conv.bias, "pad", lambda: torch.cat([_bf(conv.bias), torch.zeros(cop - co, dtype=bf16, device=x.device)]))
            y = ops.gemm(pn, wco, bco, a_grid=gr, conv3x3=True)
            return ops.unpatchify(y, I, co, H, W, 1)

    @staticmethod
    def backward(ctx, dout):
        (x,) = ctx.saved_tensors
        model = ctx.model
        I, H, W = ctx.dims
        nrm, conv = model.conv_norm_out, model.conv_out
        co, Cc = model.out_channels_, x.shape[1]
        G = Grads()
        gr = PaddedGrid(I, H, W)
        pn = ops.groupnorm_silu(x, I, H * W, _bf(nrm.weight), _bf(nrm.bias), 32, 1e-5, out_grid=gr)
        dy = ops.unshuffle_tokens(dout.contiguous(), 1, 64)                      # [px, 64]: output channels zero-padded
        G.add(conv.bias, T.segsum(dy)[0][:co])
        G.add(conv.weight, _conv3_w_to_param(conv_wgrad(dy, pn, _interior(gr, x.device), gr.tap_shifts()), co, Cc))
        dpn = ops.gemm(ops.pad_tokens(dy, gr), _conv3_flip_pad(conv.weight, 64), None, a_grid=gr, conv3x3=True)
        del pn
        dx = _gn_bwd(G, nrm, x, dpn, I, H * W, 1e-5)
        ps = _params(nrm) + _params(conv)
        return (None, None, dx) + _grads_for(G, ps, ctx.needs_input_grad[3:])


class ConcatFn(torch.autograd.Function):
    """torch.cat([a, b], channel) on token-major rows"""

    @staticmethod
    def forward(ctx, a, b):
        ctx.ca = a.shape[1]
        with torch.no_grad():
            return UM._concat_cols(a, b)

    @staticmethod
    def backward(ctx, d):
        d = d.contiguous()
        return T.rowcombine(d[:, :ctx.ca]), T.rowcombine(d[:, ctx.ca:])


# ------------------------------------------------------------------------------------------ model forward
def forward_train(model: UM.UNetCrossviewTemporalConditionModel, sample, timesteps, encoder_hidden_states=None,
                  disable_crossview=None, disable_temporal=None, crossview_attention_mask=None, added_time_ids=None,
                  condition_image_tensor=None):
    """Autograd-enabled forward of UNetCrossviewTemporalConditionModel (crossview_temporal_unet.py:655-835), layout
    ImageAdapter included (the shipped SD 2.1 training configs, configs/ctsd/*/ctsd_21_*_tirda_bm_*.json, use it).  Returns the
    prediction [B, T, V, C_out, H, W] (bf16) with a grad_fn."""
    STORE.set_precision(bf16)
    B, Tn, V, _, H, W = sample.shape
    dev = sample.device
    I = B * Tn * V
    g0 = UM._Geom(B, Tn, V, H, W, model._scratch)
    if disable_crossview is None:
        disable_crossview = torch.zeros(B, dtype=torch.bool, device=dev)
    if disable_temporal is None:
        disable_temporal = torch.zeros(B, dtype=torch.bool, device=dev)
    c0 = model.block_out_channels[0]
    emb = mlp_train(model.time_embedding, ops.timestep_sinusoid(timesteps.flatten(), c0))
    if added_time_ids is not None and model.add_embedding is not None:
        aug = ops.timestep_sinusoid(added_time_ids.flatten(), model.addition_time_embed_dim).view(I, -1)
        emb = mlp_train(model.add_embedding, aug, res=emb)
    silu_emb = SiluFn.apply(emb)
    ehs = encoder_hidden_states.flatten(0, -3)
    ehs = ehs if ehs.dtype == bf16 else ehs.to(bf16)
    ctx_rows = ehs.reshape(ehs.shape[0] * ehs.shape[1], -1).contiguous()

    xin = sample.flatten(0, 2).contiguous()
    if xin.dtype not in (torch.float32, bf16):
        xin = xin.to(bf16)
    x = ConvInFn.apply(model.conv_in, xin, *_params(model.conv_in))
    # layout residuals (crossview_temporal_unet.py:717-729, 748-750): one after conv_in, one after every down block; the
    # block's last skip connection carries the sum, as in the reference
    residuals: List[torch.Tensor] = []
    if model.condition_image_adapter is not None and condition_image_tensor is not None:
        ad = model.condition_image_adapter
        residuals = list(AdapterFn.apply(ad, condition_image_tensor, *_params(ad)))

    def add_residual(t):
        if not residuals:
            return t
        f = residuals.pop(0)
        if f.shape != t.shape:
            raise RuntimeError(f"UNet: layout residual {tuple(f.shape)} does not match the feature map {tuple(t.shape)}")
        return AddFn.apply(t, f)
    x = add_residual(x)

    def attn(tm, x, g):
        return transformer_model_train(tm, x, ctx_rows, g, disable_crossview, disable_temporal, crossview_attention_mask)

    skips: List[tuple] = [(x, H, W)]
    h_, w_ = H, W
    for blk in model.down_blocks:
        g = g0.at(h_, w_)
        for j, rb in enumerate(blk.resnets):
            x = res_block_train(rb, x, silu_emb, g, disable_temporal)
            if blk.attentions is not None:
                x = attn(blk.attentions[j], x, g)
            skips.append((x, h_, w_))
        if blk.downsamplers is not None:
            conv = blk.downsamplers[0].conv
            x = DownsampleFn.apply(conv, I, h_, w_, x, *_params(conv))
            h_, w_ = h_ // 2, w_ // 2
            skips.append((x, h_, w_))
        if residuals:
            x = add_residual(x)
            skips[-1] = (x, h_, w_)
    g = g0.at(h_, w_)
    x = res_block_train(model.mid_block.resnets[0], x, silu_emb, g, disable_temporal)
    x = attn(model.mid_block.attentions[0], x, g)
    x = res_block_train(model.mid_block.resnets[1], x, silu_emb, g, disable_temporal)
    for blk in model.up_blocks:
        g = g0.at(h_, w_)
        for j, rb in enumerate(blk.resnets):
            sk, sh, sw = skips.pop()
            if (sh, sw) != (h_, w_):
                raise RuntimeError("UNet: skip resolution mismatch (latent height / width must be divisible by 8)")
            x = res_block_train(rb, ConcatFn.apply(x, sk), silu_emb, g, disable_temporal)
            if blk.attentions is not None:
                x = attn(blk.attentions[j], x, g)
        if blk.upsamplers is not None:
            conv = blk.upsamplers[0].conv
            x = UpsampleFn.apply(conv, I, h_, w_, x, *_params(conv))
            h_, w_ = 2 * h_, 2 * w_
    out = HeadFn.apply(model, (I, H, W), x, *(_params(model.conv_norm_out) + _params(model.conv_out)))
    return out.view(B, Tn, V, model.out_channels_, H, W)
