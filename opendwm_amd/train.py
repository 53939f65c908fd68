"""Training path of the CTSD MMDiT (reference: `train_step`, src/dwm/pipelines/ctsd.py:1195-1437,
forward under autocast with gradient checkpointing `:497-521`, `loss.backward()` `:1401-1404`,
optimizer `:1406-1432`; DDP wrap `:1051-1054`).

Design (MI355X-first rather than op-by-op autograd):
  * one `torch.autograd.Function` per transformer block.  Its forward is the fused inference path
    (opendwm_amd.blocks.*.run) and keeps only the block inputs - the reference's gradient
    checkpointing; its backward re-runs the block un-fused (pre-activations are needed) and then
    walks it backwards with the hand-written HIP kernels of include/dwm_hip.h "Training".
  * parameters enter the Functions as inputs, so autograd's AccumulateGrad nodes - and with them
    `DistributedDataParallel`'s bucketed RCCL all-reduce - see every gradient as soon as its block
    is done; torch only orchestrates, all arithmetic on token-sized tensors is HIP.
  * master parameters may be fp32; compute copies are bf16 shadows (blocks.STORE) that the AdamW
    kernel refreshes in the same pass.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from . import ops
from . import train_ops as T
from .blocks import (STORE, AlphaBlender, JointTransformerBlock, TimestepEmbedding, VTSelfAttentionBlock, _bf)
from .ops import ACT_GELU_TANH, ACT_SILU

bf16 = torch.bfloat16


# ------------------------------------------------------------------------------------------ helpers
class Grads:
    """fp32 gradient accumulator keyed by parameter (one block's worth)."""

    def __init__(self):
        self.g: Dict[int, torch.Tensor] = {}

    def add(self, p: Optional[torch.Tensor], g: Optional[torch.Tensor]) -> None:
        """g: bf16 or fp32, any shape with p.numel() elements"""
        if p is None or g is None or not p.requires_grad:
            return
        cur = self.g.get(id(p))
        if g.dtype == bf16:
            g2 = g.reshape(-1, g.shape[-1]) if g.dim() > 1 else g.reshape(1, -1)
            if cur is None:
                self.g[id(p)] = T.cast_f32(g2.contiguous()).view(p.shape)
            else:
                T.cast_f32(g2.contiguous(), out=cur.view(g2.shape), accumulate=True)
        else:
            g = g.reshape(p.shape).to(torch.float32)
            if cur is None:
                self.g[id(p)] = g.clone()
            else:
                cur.add_(g)

    def take(self, p: torch.Tensor) -> Optional[torch.Tensor]:
        g = self.g.get(id(p))
        if g is None:
            return None
        return g if g.dtype == p.dtype else g.to(p.dtype)


def w_t(w: torch.Tensor) -> torch.Tensor:
    """bf16 W^T [K, N] of a 2-D weight [N, K] (N % 8 == 0), cached for one optimizer step"""
    return STORE.derived(w, "T", lambda: T.transpose(_bf(w).reshape(w.shape[0], -1), rows_pad=w.shape[0]))


def lin_fwd(x: torch.Tensor, lin: torch.nn.Linear, **epi) -> torch.Tensor:
    return ops.gemm(x, _bf(lin.weight), _bf(lin.bias), **epi)


def lin_bwd(G: Grads, lin: torch.nn.Linear, x: torch.Tensor, dy: torch.Tensor, need_dx: bool = True,
            **epi) -> Optional[torch.Tensor]:
    """gradients of y = x W^T + b; `epi` = GEMM epilogue of the input-gradient GEMM (e.g. fused residual add)"""
    want_w = lin.weight.requires_grad
    want_b = lin.bias is not None and lin.bias.requires_grad
    if want_w:
        dw, db = T.linear_wgrad(dy, x, want_bias=want_b)
        G.add(lin.weight, dw)
        G.add(lin.bias, db)
    elif want_b:
        G.add(lin.bias, T.segsum(dy)[0])
    return T.linear_dgrad(dy, w_t(lin.weight), **epi) if need_dx else None


def _fused_t(attn, added: bool) -> torch.Tensor:
    """transpose of the fused qkv projection weight [3D, D] -> [D, 3D]"""
    pk = attn.packed()
    key = "wadd" if added else "wqkv"
    anchor = (attn.add_q_proj if added else attn.to_q).weight
    return STORE.derived(anchor, "fusedT", lambda: T.transpose(pk[key], rows_pad=pk[key].shape[0]))


def qkv_bwd(G: Grads, attn, x: torch.Tensor, dqkv: torch.Tensor, added: bool = False) -> torch.Tensor:
    """backward of the fused q/k/v projection: dqkv [rows, 3D] -> dx, parameter gradients split per projection"""
    lins = (attn.add_q_proj, attn.add_k_proj, attn.add_v_proj) if added else (attn.to_q, attn.to_k, attn.to_v)
    D = lins[0].weight.shape[0]
    if lins[0].weight.requires_grad:
        want_b = lins[0].bias is not None
        dw, db = T.linear_wgrad(dqkv, x, want_bias=want_b)
        for i, l in enumerate(lins):
            G.add(l.weight, dw[i * D:(i + 1) * D])
            if want_b:
                G.add(l.bias, db[i * D:(i + 1) * D])
    return T.linear_dgrad(dqkv, _fused_t(attn, added))


def qk_norm_bwd(G: Grads, attn, qkv: torch.Tensor, rinv: Optional[torch.Tensor], dqkv: torch.Tensor, added: bool = False) -> None:
    """in place on dqkv[:, :2D]; folds the per-column weight gradient back onto norm_{q,k}.weight [64]"""
    if rinv is None:
        return
    pk = attn.packed()
    rms = pk["rms_add" if added else "rms"]
    D2 = rms.numel()
    dw = torch.zeros(D2, dtype=torch.float32, device=qkv.device)
    T.rmsnorm_heads_bwd_(qkv[:, :D2], rinv, rms, dqkv[:, :D2], dw)
    nq, nk = (attn.norm_added_q, attn.norm_added_k) if added else (attn.norm_q, attn.norm_k)
    dw = dw.view(2, attn.heads, 64).sum(1)
    G.add(nq.weight, dw[0])
    G.add(nk.weight, dw[1])


def project_qkv_train(attn, x: torch.Tensor, added: bool = False):
    """fused projection + in-place per-head RMSNorm keeping 1/rms for the backward"""
    pk = attn.packed()
    w, b, rms = (pk["wadd"], pk["badd"], pk.get("rms_add")) if added else (pk["wqkv"], pk["bqkv"], pk.get("rms"))
    qkv = ops.gemm(x, w, b)
    rinv = T.rmsnorm_heads_train_(qkv[:, :rms.numel()], rms, attn.eps) if rms is not None else None
    return qkv, rinv


def small_mlp_fwd(m: TimestepEmbedding, x: torch.Tensor):
    u = lin_fwd(x, m.linear_1)
    a = T.act_fwd(u, ACT_SILU)
    return lin_fwd(a, m.linear_2), (x, u, a)


def small_mlp_bwd(G: Grads, m: TimestepEmbedding, saved, dy: torch.Tensor, need_dx: bool = False):
    x, u, a = saved
    da = lin_bwd(G, m.linear_2, a, dy)
    du = T.act_bwd(u, da, ACT_SILU)
    return lin_bwd(G, m.linear_1, x, du, need_dx=need_dx)


def _params(mod: torch.nn.Module) -> List[torch.nn.Parameter]:
    return [p for p in mod.parameters()]


def _grads_for(G: Grads, params, needs) -> tuple:
    return tuple((G.take(p) if need else None) for p, need in zip(params, needs))


# ------------------------------------------------------------------------------------------ joint block
def joint_block_backward(blk: JointTransformerBlock, h: torch.Tensor, c: torch.Tensor, st: torch.Tensor, n_img: int,
                         dh: torch.Tensor, dc: Optional[torch.Tensor], G: Grads):
    """Recompute + backward of JointTransformerBlock.run (diffusers JointTransformerBlock.forward).
    h [I*N, D], c [I*Lc, D], st = silu(temb) [I, D]: the block INPUTS; dh / dc: gradients of its outputs.
    Returns (dh_in, dc_in, dst)."""
    D = blk.dim
    N, Lc = h.shape[0] // n_img, c.shape[0] // n_img
    dual, pre_only = blk.use_dual_attention, blk.context_pre_only
    sl = lambda m, i: m[:, i * D:(i + 1) * D]
    id_map = ops.rowmap_identity(n_img, N)

    # ---------------- recompute (un-fused where a pre-activation is needed)
    mod = lin_fwd(st, blk.norm1.linear)
    cmod = lin_fwd(st, blk.norm1_context.linear)
    nh2 = torch.empty_like(h) if dual else None
    nh = ops.layernorm(h, eps=1e-6, scale=sl(mod, 1), shift=sl(mod, 0), rows_per_mod=N,
                       scale2=sl(mod, 7) if dual else None, shift2=sl(mod, 6) if dual else None, out2=nh2)
    cs, cb = (0, 1) if pre_only else (1, 0)           # AdaLayerNormContinuous: scale first
    nc = ops.layernorm(c, eps=1e-6, scale=sl(cmod, cs), shift=sl(cmod, cb), rows_per_mod=Lc)
    qkv, rinv = project_qkv_train(blk.attn, nh)
    cqkv, crinv = project_qkv_train(blk.attn, nc, added=True)
    ao, cao = torch.empty_like(h), torch.empty_like(c)
    lse = torch.empty(n_img * blk.heads * (N + Lc), dtype=torch.float32, device=h.device)
    ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], ao, id_map, blk.heads,
                  q1=cqkv[:, :D], k1=cqkv[:, D:2 * D], v1=cqkv[:, 2 * D:], out1=cao, lse=lse)
    t_att = lin_fwd(ao, blk.attn.to_out[0])
    h1 = T.rowcombine(t_att, gate_a=sl(mod, 2), rows_per_gate_a=N, b=h)
    if dual:
        qkv2, rinv2 = project_qkv_train(blk.attn2, nh2)
        ao2 = torch.empty_like(h)
        lse2 = torch.empty(n_img * blk.heads * N, dtype=torch.float32, device=h.device)
        ops.attention(qkv2[:, :D], qkv2[:, D:2 * D], qkv2[:, 2 * D:], ao2, id_map, blk.heads, lse=lse2)
        t_att2 = lin_fwd(ao2, blk.attn2.to_out[0])
        h1 = T.rowcombine(t_att2, gate_a=sl(mod, 8), rows_per_gate_a=N, b=h1)
    nh3 = ops.layernorm(h1, eps=1e-6, scale=sl(mod, 4), shift=sl(mod, 3), rows_per_mod=N)
    f1, f2 = blk.ff.net[0].proj, blk.ff.net[2]
    u = lin_fwd(nh3, f1)
    ffh = T.act_fwd(u, ACT_GELU_TANH)
    t_ff = lin_fwd(ffh, f2)

    dmod = torch.zeros(mod.shape, dtype=torch.float32, device=h.device)
    dcmod = torch.zeros(cmod.shape, dtype=torch.float32, device=h.device)

    # ---------------- context stream tail (needed first: the joint attention backward wants d(cao))
    if not pre_only:
        t_catt = lin_fwd(cao, blk.attn.to_add_out)
        c1 = T.rowcombine(t_catt, gate_a=sl(cmod, 2), rows_per_gate_a=Lc, b=c)
        nc3 = ops.layernorm(c1, eps=1e-6, scale=sl(cmod, 4), shift=sl(cmod, 3), rows_per_mod=Lc)
        c1f, c2f = blk.ff_context.net[0].proj, blk.ff_context.net[2]
        cu = lin_fwd(nc3, c1f)
        cffh = T.act_fwd(cu, ACT_GELU_TANH)
        t_cff = lin_fwd(cffh, c2f)
        # c2 = c1 + gate_mlp * t_cff
        T.segsum(t_cff, dc, rows_per_group=Lc, out=sl(dcmod, 5))
        dct = T.rowcombine(dc, gate_a=sl(cmod, 5), rows_per_gate_a=Lc)
        dcffh = lin_bwd(G, c2f, cffh, dct)
        dcu = T.act_bwd(cu, dcffh, ACT_GELU_TANH)
        dnc3 = lin_bwd(G, c1f, nc3, dcu)
        dc1 = dc.clone()
        T.layernorm_bwd(c1, dnc3, eps=1e-6, dx=dc1, accumulate=True, scale=sl(cmod, 4), rows_per_mod=Lc,
                        dgamma=sl(dcmod, 4), dbeta=sl(dcmod, 3), grad_per_group=True)
        # c1 = c + gate_msa * t_catt
        T.segsum(t_catt, dc1, rows_per_group=Lc, out=sl(dcmod, 2))
        dct1 = T.rowcombine(dc1, gate_a=sl(cmod, 2), rows_per_gate_a=Lc)
        dcao = lin_bwd(G, blk.attn.to_add_out, cao, dct1)
    else:
        dc1 = torch.zeros_like(c)
        dcao = torch.zeros_like(c)

    # ---------------- sample stream: h2 = h1 + gate_mlp * t_ff
    T.segsum(t_ff, dh, rows_per_group=N, out=sl(dmod, 5))
    dt = T.rowcombine(dh, gate_a=sl(mod, 5), rows_per_gate_a=N)
    dffh = lin_bwd(G, f2, ffh, dt)
    du = T.act_bwd(u, dffh, ACT_GELU_TANH)
    dnh3 = lin_bwd(G, f1, nh3, du)
    dh1 = dh.clone()
    T.layernorm_bwd(h1, dnh3, eps=1e-6, dx=dh1, accumulate=True, scale=sl(mod, 4), rows_per_mod=N,
                    dgamma=sl(dmod, 4), dbeta=sl(dmod, 3), grad_per_group=True)
    dnh2 = None
    if dual:       # h1 = h1a + gate2 * t_att2
        T.segsum(t_att2, dh1, rows_per_group=N, out=sl(dmod, 8))
        dt2 = T.rowcombine(dh1, gate_a=sl(mod, 8), rows_per_gate_a=N)
        dao2 = lin_bwd(G, blk.attn2.to_out[0], ao2, dt2)
        dqkv2 = torch.empty_like(qkv2)
        ops.attention_bwd(qkv2[:, :D], qkv2[:, D:2 * D], qkv2[:, 2 * D:], ao2, dao2,
                          dqkv2[:, :D], dqkv2[:, D:2 * D], dqkv2[:, 2 * D:], id_map, blk.heads, lse2)
        qk_norm_bwd(G, blk.attn2, qkv2, rinv2, dqkv2)
        dnh2 = qkv_bwd(G, blk.attn2, nh2, dqkv2)
    # h1a = h + gate_msa * t_att
    T.segsum(t_att, dh1, rows_per_group=N, out=sl(dmod, 2))
    dt1 = T.rowcombine(dh1, gate_a=sl(mod, 2), rows_per_gate_a=N)
    dao = lin_bwd(G, blk.attn.to_out[0], ao, dt1)

    # ---------------- joint attention
    dqkv, dcqkv = torch.empty_like(qkv), torch.empty_like(cqkv)
    ops.attention_bwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], ao, dao, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:],
                      id_map, blk.heads, lse, q1=cqkv[:, :D], k1=cqkv[:, D:2 * D], v1=cqkv[:, 2 * D:], out1=cao, dout1=dcao,
                      dq1=dcqkv[:, :D], dk1=dcqkv[:, D:2 * D], dv1=dcqkv[:, 2 * D:])
    qk_norm_bwd(G, blk.attn, qkv, rinv, dqkv)
    qk_norm_bwd(G, blk.attn, cqkv, crinv, dcqkv, added=True)
    dnh = qkv_bwd(G, blk.attn, nh, dqkv)
    dnc = qkv_bwd(G, blk.attn, nc, dcqkv, added=True)

    # ---------------- the two AdaLN layers at the block input
    T.layernorm_bwd(h, dnh, eps=1e-6, dx=dh1, accumulate=True, scale=sl(mod, 1), scale2=sl(mod, 7) if dual else None,
                    rows_per_mod=N, dy2=dnh2, dgamma=sl(dmod, 1), dbeta=sl(dmod, 0),
                    dgamma2=sl(dmod, 7) if dual else None, dbeta2=sl(dmod, 6) if dual else None, grad_per_group=True)
    T.layernorm_bwd(c, dnc, eps=1e-6, dx=dc1, accumulate=True, scale=sl(cmod, cs), rows_per_mod=Lc,
                    dgamma=sl(dcmod, cs), dbeta=sl(dcmod, cb), grad_per_group=True)

    # ---------------- modulation linears (M = images)
    dst = lin_bwd(G, blk.norm1.linear, st, dmod.to(bf16))
    dst2 = lin_bwd(G, blk.norm1_context.linear, st, dcmod.to(bf16))
    ops.add_(dst, dst2)
    return dh1, dc1, dst


@ops.carries_gemm_scope
class JointBlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, blk, n_img, h, c, st, *params):
        ctx.blk, ctx.n_img = blk, n_img
        ctx.save_for_backward(h, c, st)
        ctx.np = len(params)
        with torch.no_grad():
            c_out, h_out = blk.run(h.clone(), c.clone(), st, n_img)
        if c_out is None:
            c_out = c.new_zeros(())           # context_pre_only: the context stream ends here
        return h_out, c_out

    @staticmethod
    def backward(ctx, dh, dc):
        h, c, st = ctx.saved_tensors
        blk = ctx.blk
        G = Grads()
        if blk.context_pre_only:
            dc = None
        dh_in, dc_in, dst = joint_block_backward(blk, h, c, st, ctx.n_img, dh.contiguous(),
                                                 None if dc is None else dc.contiguous(), G)
        ps = _params(blk)
        return (None, None, dh_in, dc_in, dst) + _grads_for(G, ps, ctx.needs_input_grad[5:])


def joint_block_train(blk: JointTransformerBlock, h, c, st, n_img: int):
    return JointBlockFn.apply(blk, n_img, h, c, st, *_params(blk))


# ------------------------------------------------------------------------------------------ VT block + mixer
def vt_block_backward(blk: VTSelfAttentionBlock, h: torch.Tensor, emb: torch.Tensor, rows_per_emb: int, rowmap,
                      group_mask, dense_mask, alpha: torch.Tensor, rows_per_alpha: int, dy: torch.Tensor, G: Grads):
    """Recompute + backward of VTSelfAttentionBlock.run followed by the AlphaBlender
    (crossview_temporal.py:562-582, 68-72; crossview_temporal_dit.py:320-327,363-370).
    Returns (dh, demb fp32 [G, D], dalpha fp32 [B])."""
    D = blk.dim
    ln = lambda x, n, **kw: ops.layernorm(x, eps=1e-5, weight=_bf(n.weight), bias=_bf(n.bias), **kw)
    # ---------------- recompute
    xs0 = torch.empty_like(h)
    y = ln(h, blk.norm_in, addvec=emb, rows_per_add=rows_per_emb, xsum=xs0)          # xs0 = h + emb
    pi, l2i = blk.ff_in.net[0].proj, blk.ff_in.net[2]
    u1 = lin_fwd(y, pi)
    g1 = T.geglu_fwd(u1)
    xs1 = lin_fwd(g1, l2i, epilogue=ops.EPI_RESID, res=xs0)
    y1 = ln(xs1, blk.norm1)
    qkv, rinv = project_qkv_train(blk.attn1, y1)
    ao = torch.empty_like(h)
    lse = torch.empty(rowmap.n_problems * blk.heads * rowmap.L0, dtype=torch.float32, device=h.device)
    ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], ao, rowmap, blk.heads, group_mask=group_mask,
                  dense_mask=dense_mask, lse=lse)
    to_out = blk.attn1.to_out[0]
    xs2 = lin_fwd(ao, to_out, epilogue=ops.EPI_RESID, res=xs1)
    y3 = ln(xs2, blk.norm3)
    p3, l23 = blk.ff.net[0].proj, blk.ff.net[2]
    u3 = lin_fwd(y3, p3)
    g3 = T.geglu_fwd(u3)
    out = lin_fwd(g3, l23, epilogue=ops.EPI_RESID, res=xs2)

    # ---------------- mixer: blended = alpha * h + (1 - alpha) * out
    one_minus = (1.0 - alpha).contiguous()
    # d(alpha) = <dy, h - out>: one pass, fp32 difference before the product, fp32 column sums, summed in fp64 - the value
    # is a heavily cancelling sum (sum|terms| / |sum| = 200-3000 on the test configurations), so no partial sum is rounded
    dalpha = T.segsum_diff(dy, h, out, rows_per_group=rows_per_alpha).double().sum(-1).float()
    dout = T.rowcombine(dy, coef_a=one_minus, rows_per_coef_a=rows_per_alpha)
    del out

    def ln_bwd(x, n, dyy, dx, **kw):
        dg = torch.zeros(1, D, dtype=torch.float32, device=h.device)
        db = torch.zeros(1, D, dtype=torch.float32, device=h.device)
        T.layernorm_bwd(x, dyy, eps=1e-5, dx=dx, accumulate=True, weight=_bf(n.weight), dgamma=dg, dbeta=db, **kw)
        G.add(n.weight, dg[0])
        G.add(n.bias, db[0])

    # out = xs2 + ff(norm3(xs2))
    dxs2 = dout
    dg3 = lin_bwd(G, l23, g3, dout)
    du3 = T.geglu_bwd(u3, dg3)
    dy3 = lin_bwd(G, p3, y3, du3)
    ln_bwd(xs2, blk.norm3, dy3, dxs2)
    # xs2 = xs1 + to_out(attn(norm1(xs1)))
    dao = lin_bwd(G, to_out, ao, dxs2)
    dqkv = torch.empty_like(qkv)
    ops.attention_bwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], ao, dao, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:],
                      rowmap, blk.heads, lse, group_mask=group_mask, dense_mask=dense_mask)
    qk_norm_bwd(G, blk.attn1, qkv, rinv, dqkv)
    dy1 = qkv_bwd(G, blk.attn1, y1, dqkv)
    dxs1 = dxs2
    ln_bwd(xs1, blk.norm1, dy1, dxs1)
    # xs1 = xs0 + ff_in(norm_in(xs0))
    dg1 = lin_bwd(G, l2i, g1, dxs1)
    du1 = T.geglu_bwd(u1, dg1)
    dyin = lin_bwd(G, pi, y, du1)
    dxs0 = dxs1
    ln_bwd(h, blk.norm_in, dyin, dxs0, addvec=emb, rows_per_add=rows_per_emb)
    # xs0 = h + emb[row // rows_per_emb]   (one embedding row per token - explicit perspective modelling - needs no sum)
    demb = dxs0 if rows_per_emb == 1 else T.segsum(dxs0, rows_per_group=rows_per_emb)
    dh = T.rowcombine(dy, coef_a=alpha, rows_per_coef_a=rows_per_alpha, b=dxs0)
    return dh, demb, dalpha


@ops.carries_gemm_scope
class VTBlockFn(torch.autograd.Function):
    """h_out = AlphaBlender(h, VTSelfAttentionBlock(h + emb)); inputs that carry gradients: h, emb, alpha."""

    @staticmethod
    def forward(ctx, blk, rowmap, rows_per_emb, rows_per_alpha, group_mask, dense_mask, h, emb, alpha, *params):
        ctx.blk, ctx.rowmap, ctx.rpe, ctx.rpa = blk, rowmap, rows_per_emb, rows_per_alpha
        ctx.gm, ctx.dm = group_mask, dense_mask
        ctx.save_for_backward(h, emb, alpha)
        with torch.no_grad():
            out = h.clone()
            blk.run(out, rowmap, emb=emb, rows_per_emb=rows_per_emb, group_mask=group_mask, dense_mask=dense_mask,
                    blend_alpha=alpha, rows_per_alpha=rows_per_alpha, blend_into=out)
        return out

    @staticmethod
    def backward(ctx, dy):
        h, emb, alpha = ctx.saved_tensors
        G = Grads()
        dh, demb, dalpha = vt_block_backward(ctx.blk, h, emb, ctx.rpe, ctx.rowmap, ctx.gm, ctx.dm, alpha, ctx.rpa,
                                             dy.contiguous(), G)
        ps = _params(ctx.blk)
        return (None, None, None, None, None, None, dh, demb.to(emb.dtype), dalpha.to(alpha.dtype)) + \
            _grads_for(G, ps, ctx.needs_input_grad[9:])


def vt_block_train(blk, h, rowmap, emb, rows_per_emb, alpha, rows_per_alpha, group_mask=None, dense_mask=None):
    return VTBlockFn.apply(blk, rowmap, rows_per_emb, rows_per_alpha, group_mask, dense_mask, h, emb, alpha, *_params(blk))


def alpha_train(mixer: AlphaBlender, image_only_indicator: Optional[torch.Tensor], batch: int) -> torch.Tensor:
    """AlphaBlender.get_alpha with autograd on mix_factor (one scalar: plain torch)."""
    mf = mixer.mix_factor.float()
    if mixer.merge_strategy == "fixed":
        return mf.expand(batch).contiguous()
    if mixer.merge_strategy == "learned":
        return torch.sigmoid(mf).expand(batch).contiguous()
    flag = image_only_indicator.reshape(batch).to(device=mf.device, dtype=torch.bool)
    return torch.where(flag, torch.ones((), device=mf.device), torch.sigmoid(mf).expand(batch)).contiguous()


# ------------------------------------------------------------------------------------------ small modules
@ops.carries_gemm_scope
class MlpFn(torch.autograd.Function):
    """TimestepEmbedding: linear_2(silu(linear_1(x))) [+ res]; x carries no gradient (sinusoids / pooled text)."""

    @staticmethod
    def forward(ctx, m, x, res, *params):
        ctx.m = m
        with torch.no_grad():
            y, saved = small_mlp_fwd(m, x)
            if res is not None:
                y = T.rowcombine(y, b=res)
        ctx.saved = saved
        ctx.has_res = res is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        G = Grads()
        dy = dy.contiguous()
        small_mlp_bwd(G, ctx.m, ctx.saved, dy)
        ps = _params(ctx.m)
        return (None, None, dy if ctx.has_res else None) + _grads_for(G, ps, ctx.needs_input_grad[3:])


def mlp_train(m: TimestepEmbedding, x: torch.Tensor, res: Optional[torch.Tensor] = None) -> torch.Tensor:
    return MlpFn.apply(m, x, res, *_params(m))


@ops.carries_gemm_scope
class LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, lin, x, *params):
        ctx.lin = lin
        ctx.save_for_backward(x)
        with torch.no_grad():
            return lin_fwd(x, lin)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        G = Grads()
        dx = lin_bwd(G, ctx.lin, x, dy.contiguous(), need_dx=ctx.needs_input_grad[1])
        return (None, dx) + _grads_for(G, _params(ctx.lin), ctx.needs_input_grad[2:])


def linear_train(lin: torch.nn.Linear, x: torch.Tensor) -> torch.Tensor:
    return LinearFn.apply(lin, x, *_params(lin))


class SiluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        with torch.no_grad():
            return T.act_fwd(x, ACT_SILU)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return T.act_bwd(x, dy.contiguous(), ACT_SILU)


class AddFn(torch.autograd.Function):
    """a + b on bf16 matrices (HIP)"""

    @staticmethod
    def forward(ctx, a, b):
        with torch.no_grad():
            return T.rowcombine(a, b=b)

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


@ops.carries_gemm_scope
class PatchEmbedFn(torch.autograd.Function):
    """SD3 PatchEmbed (strided conv as patchify + GEMM, + cropped pos embed); the latents carry no gradient."""

    @staticmethod
    def forward(ctx, pe, x, *params):
        ctx.pe = pe
        ctx.save_for_backward(x)
        with torch.no_grad():
            return pe.run(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        pe = ctx.pe
        G = Grads()
        if pe.proj.weight.requires_grad:
            wp = pe.packed_weight()
            cols = ops.patchify(x, pe.patch_size, wp.shape[1])
            dw, db = T.linear_wgrad(dy.contiguous(), cols)
            k = pe.proj.weight[0].numel()
            G.add(pe.proj.weight, dw[:, :k].contiguous())
            G.add(pe.proj.bias, db)
        return (None, None) + _grads_for(G, _params(pe), ctx.needs_input_grad[2:])


@ops.carries_gemm_scope
class OutFn(torch.autograd.Function):
    """norm_out (AdaLayerNormContinuous) + proj_out + unpatchify"""

    @staticmethod
    def forward(ctx, model, geom, h, st, *params):
        ctx.model, ctx.geom = model, geom
        ctx.save_for_backward(h, st)
        I, C, hh, ww, p, N, D = geom
        with torch.no_grad():
            mod = lin_fwd(st, model.norm_out.linear)
            nh = ops.layernorm(h, eps=1e-6, scale=mod[:, :D], shift=mod[:, D:], rows_per_mod=N)
            y = lin_fwd(nh, model.proj_out)
            return ops.unpatchify(y, I, C, hh, ww, p)

    @staticmethod
    def backward(ctx, dout):
        h, st = ctx.saved_tensors
        model = ctx.model
        I, C, hh, ww, p, N, D = ctx.geom
        G = Grads()
        mod = lin_fwd(st, model.norm_out.linear)
        nh = ops.layernorm(h, eps=1e-6, scale=mod[:, :D], shift=mod[:, D:], rows_per_mod=N)
        # unpatchify is a permutation: its transpose is the patch gather with (py, px, c) column order
        do = dout.to(bf16).contiguous().view(I, C, hh, p, ww, p).permute(0, 2, 4, 3, 5, 1).reshape(I * hh * ww, p * p * C).contiguous()
        dnh = lin_bwd(G, model.proj_out, nh, do)
        dmod = torch.zeros(mod.shape, dtype=torch.float32, device=h.device)
        dh = T.layernorm_bwd(h, dnh, eps=1e-6, scale=mod[:, :D], rows_per_mod=N, dgamma=dmod[:, :D], dbeta=dmod[:, D:],
                             grad_per_group=True)
        dst = lin_bwd(G, model.norm_out.linear, st, dmod.to(bf16))
        ps = _params(model.norm_out) + _params(model.proj_out)
        return (None, None, dh, dst) + _grads_for(G, ps, ctx.needs_input_grad[4:])


# ------------------------------------------------------------------------------------------ layout ImageAdapter
def _conv3_wgrad(dh: torch.Tensor, x_pad: torch.Tensor, grid, idx: torch.Tensor) -> torch.Tensor:
    """dW [N, 9*C] (tap-major, bf16) of h = conv3x3(x): dW[n, t, c] = sum_pixels dh[pixel, n] * x_pad[row(pixel) + shift_t, c].
    (train_ops.conv_wgrad: dh scattered onto the padded grid, all nine taps one dwm_gemm_tn launch)"""
    return T.conv_wgrad(dh, x_pad, idx, grid.tap_shifts())


def _conv3_flip(w: torch.Tensor) -> torch.Tensor:
    """weight of the input-gradient convolution: [N, C, 3, 3] -> tap-major [C, 9*N] with the taps mirrored"""
    return STORE.derived(w, "c3flip", lambda: _bf(w).flip(2, 3).permute(1, 2, 3, 0).reshape(w.shape[1], -1).contiguous())


@ops.carries_gemm_scope
class AdapterFn(torch.autograd.Function):
    """dwm.models.adapters.ImageAdapter (src/dwm/models/adapters.py:40-60) with a hand-written backward.  Forward =
    the inference path (opendwm_amd.adapters.ImageAdapter.run) keeping only the condition images (the reference wraps
    the adapter body in gradient checkpointing, adapters.py:44-52); backward recomputes the body level by level, keeping
    each resnet's input grid and ReLU output, then walks it in reverse: 1x1 convolutions as GEMMs, the 3x3 input gradient
    as the implicit GEMM with mirrored taps, the 3x3 weight gradient as nine tap-shifted GEMMs."""

    @staticmethod
    def forward(ctx, adapter, x, *params):
        ctx.adapter = adapter
        ctx.save_for_backward(x)
        with torch.no_grad():
            return tuple(adapter.run(x))

    @staticmethod
    def backward(ctx, *dfeats):
        from .ops import ACT_RELU, EPI_RESID, PaddedGrid
        (x,) = ctx.saved_tensors
        ad = ctx.adapter
        G = Grads()
        x = x.flatten(0, -4).contiguous()
        if x.dtype not in (torch.float32, bf16):
            x = x.to(bf16)
        I, _, H, W = x.shape
        r = ad.downscale_factor
        h, w = H // r, W // r
        # ---- recompute, keeping what the backward needs.  A level = one AdapterBlock; its grid changes where the block
        # starts with AvgPool2d (the MMDiT layout adapters pool in the first block only, the SD 2.1 UNet's in blocks 1-3)
        cur = ops.unshuffle_tokens(x, r)            # compact tokens; None while the running tensor lives on a padded grid
        levels = []
        xp, grid, idx = None, None, None
        for bi, (blk, zc) in enumerate(zip(ad.body, ad.zero_convs)):
            lv = {"blk": blk, "zc": zc, "res": [], "pooled_from": None}
            if blk.downsample is not None:
                if cur is None:
                    cur = xp[idx]                                                 # compact rows of the previous level's output
                if h % 2 or w % 2:
                    raise NotImplementedError("AvgPool2d(ceil_mode) on odd sizes")
                cur = ops.avgpool2_tokens(cur, I, h, w)
                lv["pooled_from"] = (h, w)
                h, w = h // 2, w // 2
                xp = None
            if xp is None:
                grid = PaddedGrid(I, h, w)
                idx = grid.interior_index().to(x.device)
                if blk.in_conv is not None:
                    wi = _bf(blk.in_conv.weight).reshape(blk.in_conv.weight.shape[0], -1)
                    if wi.shape[1] != cur.shape[1]:                               # first block: K padded to 64 by unshuffle_tokens
                        wpad = torch.zeros((wi.shape[0], cur.shape[1]), dtype=bf16, device=wi.device)
                        wpad[:, :wi.shape[1]] = wi
                        wi = wpad
                    xp = ops.gemm(cur, wi.contiguous(), _bf(blk.in_conv.bias), c_grid=grid)
                    lv["in"] = ("compact", cur)
                else:
                    xp = ops.pad_tokens(cur, grid)
                    lv["in"] = ("pad", None)
                cur = None
            elif blk.in_conv is not None:
                wi = _bf(blk.in_conv.weight).reshape(blk.in_conv.weight.shape[0], -1).contiguous()
                lv["in"] = ("grid", xp)
                xp = ops.gemm(xp, wi, _bf(blk.in_conv.bias), a_grid=grid, c_grid=grid)
            else:
                lv["in"] = ("same", None)
            lv["grid"], lv["idx"], lv["first"] = grid, idx, bi == 0
            for res in blk.resnets:
                pk = res.packed()
                h1 = ops.gemm(xp, pk["w3"], _bf(res.block1.bias), act=ACT_RELU, a_grid=grid, conv3x3=True)
                xn = ops.gemm(h1, pk["w1"], _bf(res.block2.bias), epilogue=EPI_RESID, res=xp, c_grid=grid)
                lv["res"].append((res, xp, h1))
                xp = xn
            lv["out"] = xp
            levels.append(lv)
        # ---- backward, last level first; dx: gradient w.r.t. the level's output grid (padded rows, zero border)
        dx = None
        quarter = torch.full((1,), 0.25, dtype=torch.float32, device=x.device)
        for lv, df in zip(reversed(levels), reversed(dfeats)):
            grid, idx = lv["grid"], lv["idx"]
            if df is not None:
                df = df.to(bf16).contiguous()
                zc = lv["zc"]
                if zc is not None:
                    xc = lv["out"][idx]                                           # compact rows of the level output
                    dwz, dbz = T.linear_wgrad(df, xc, want_bias=True)
                    G.add(zc.weight, dwz)
                    G.add(zc.bias, dbz)
                    wz_t = w_t(zc.weight)
                    dx = ops.gemm(df, wz_t, None, c_grid=grid) if dx is None else \
                        ops.gemm(df, wz_t, None, epilogue=EPI_RESID, res=dx, out=dx, c_grid=grid)
                else:
                    dpad = ops.pad_tokens(df, grid)
                    dx = dpad if dx is None else T.rowcombine(dx, b=dpad)
            if dx is None:
                continue
            for res, xin, h1 in reversed(lv["res"]):
                dyc = dx[idx]                                                     # compact gradient of the block output
                dw1, db1 = T.linear_wgrad(dyc, h1, want_bias=True)
                G.add(res.block2.weight, dw1)
                G.add(res.block2.bias, db1)
                dh1 = T.linear_dgrad(dyc, w_t(res.block2.weight))
                T.act_bwd(h1, dh1, ACT_RELU, out=dh1)
                G.add(res.block1.bias, T.segsum(dh1)[0])
                G.add(res.block1.weight, _conv3_wgrad(dh1, xin, grid, idx).view(dh1.shape[1], 3, 3, -1).permute(0, 3, 1, 2))
                dhp = ops.pad_tokens(dh1, grid)
                ops.gemm(dhp, _conv3_flip(res.block1.weight), None, a_grid=grid, conv3x3=True, epilogue=EPI_RESID, res=dx,
                         out=dx, c_grid=grid)                                     # dx += conv3x3^T(dh1), in place
            kind, saved = lv["in"]
            blk = lv["blk"]
            dcur = None                           # gradient w.r.t. the compact tensor that entered this level (if it had one)
            if kind == "grid":                    # in_conv on the previous level's grid (same resolution)
                dxc = dx[idx]
                dwi, dbi = T.linear_wgrad(dxc, saved[idx], want_bias=True)
                G.add(blk.in_conv.weight, dwi)
                G.add(blk.in_conv.bias, dbi)
                dx = ops.gemm(dxc, w_t(blk.in_conv.weight), None, c_grid=grid)
            elif kind == "compact":
                dxc = dx[idx]
                dwi, dbi = T.linear_wgrad(dxc, saved, want_bias=True)
                ci = blk.in_conv.weight.shape[1]
                G.add(blk.in_conv.weight, dwi[:, :ci].contiguous())
                G.add(blk.in_conv.bias, dbi)
                if not lv["first"]:
                    dcur = T.linear_dgrad(dxc, w_t(blk.in_conv.weight))
            elif kind == "pad":
                if not lv["first"]:
                    dcur = dx[idx]
            # kind == "same": the previous level's grid IS this level's input: dx carries on unchanged
            if kind in ("compact", "pad"):
                if lv["first"] or dcur is None:
                    dx = None                                                     # the condition images carry no gradient
                else:
                    if lv["pooled_from"] is None:
                        raise RuntimeError("adapter backward: a level that re-grids without pooling")
                    # AvgPool2d(2) backward: every child pixel gets a quarter of the pooled gradient - nearest 2x upsample
                    # straight onto the previous level's padded grid
                    hp, wp_ = lv["pooled_from"]
                    up = ops.upsample2_padded(dcur, I, hp // 2, wp_ // 2)
                    dx = T.rowcombine(up, coef_a=quarter, rows_per_coef_a=up.shape[0])
        ps = _params(ad)
        return (None, None) + _grads_for(G, ps, ctx.needs_input_grad[2:])


@ops.carries_gemm_scope
class RayEmbFn(torch.autograd.Function):
    """Explicit perspective modelling (crossview_temporal_dit.py:440-458, 528-568): per-token embedding of a cross-view /
    temporal block = per-image index embedding [I, D] + RayEncoder.proj(ray features [I*N, 72]).  One K = 128 GEMM with the
    per-image row as the epilogue residual; the ray features carry no gradient, `proj.weight` and the index embedding do."""

    @staticmethod
    def forward(ctx, renc, n_tok, ray_feat, emb_img, *params):
        ctx.renc, ctx.n_tok = renc, n_tok
        ctx.save_for_backward(ray_feat)
        with torch.no_grad():
            return ops.gemm(ray_feat, renc.packed(), None, epilogue=ops.EPI_RESID, res=emb_img.contiguous(), res_mod=-n_tok)

    @staticmethod
    def backward(ctx, dout):
        (ray_feat,) = ctx.saved_tensors
        renc = ctx.renc
        G = Grads()
        dout = dout.contiguous()
        if renc.proj.weight.requires_grad:
            dw, _ = T.linear_wgrad(dout, ray_feat, want_bias=False)                  # [D, 128]: K was zero-padded from 72
            G.add(renc.proj.weight, dw[:, :renc.proj.weight.shape[1]].contiguous())
        demb = ops.cast_bf16(T.segsum(dout, rows_per_group=ctx.n_tok))
        return (None, None, None, demb) + _grads_for(G, _params(renc), ctx.needs_input_grad[4:])


# ------------------------------------------------------------------------------------------ model forward
def forward_train(model, sample, timestep, encoder_hidden_states, pooled_projections, disable_crossview=None,
                  disable_temporal=None, crossview_attention_mask=None, added_time_ids=None, condition_image_tensor=None,
                  camera_intrinsics_norm=None, camera2referego=None):
    """Autograd-enabled forward of DiTCrossviewTemporalConditionModel (text-conditioned configuration;
    crossview_temporal_dit.py:372-630).  Returns the prediction [B, T, V, C, H, W] (bf16) with a grad_fn."""
    STORE.set_precision(bf16)               # training runs in bf16 compute over fp32 masters
    B, Tn, V, _, H, W = sample.shape
    p = model._cfg.patch_size
    height, width = H // p, W // p
    N, I, D = height * width, B * Tn * V, model.inner_dim
    dev = sample.device

    def as_bf16(t):
        return t if t.dtype == bf16 else (ops.cast_bf16(t.contiguous()) if t.dtype == torch.float32 else t.to(bf16))

    x = sample.flatten(0, 2).contiguous()
    h = PatchEmbedFn.apply(model.pos_embed, x, *_params(model.pos_embed))
    ehs = as_bf16(encoder_hidden_states.flatten(0, 2))
    Lc = ehs.shape[1]
    c = linear_train(model.context_embedder, ehs.reshape(I * Lc, -1))
    pooled = as_bf16(pooled_projections.flatten(0, 2)).contiguous()
    tte = model.time_text_embed
    t_emb = mlp_train(tte.timestep_embedder, ops.timestep_sinusoid(timestep.flatten(), 256))
    temb = mlp_train(tte.text_embedder, pooled, res=t_emb)
    st = SiluFn.apply(temb)

    view_cam_emb = None
    ray_feat = None
    if model.perspective_modeling_type == "explicit":                                              # :440-458
        if camera_intrinsics_norm is None or camera2referego is None:
            raise RuntimeError("perspective_modeling_type='explicit' needs camera_intrinsics_norm and camera2referego")
        with torch.no_grad():
            ray_feat = model.rayencoder.features(camera_intrinsics_norm, camera2referego, height, width)

    def with_rays(emb_img):
        """(embedding for the VT block, rows per embedding row): the per-image row alone, or row + ray projection per token"""
        if ray_feat is None:
            return emb_img, N
        return RayEmbFn.apply(model.rayencoder, N, ray_feat, emb_img, *_params(model.rayencoder)), 1
    if model.perspective_modeling_type == "implicit":
        ve = ops.timestep_sinusoid(added_time_ids.flatten(), 256).view(I, -1)
        view_cam_emb = mlp_train(model.view_embedding, ve)
    if model.enable_crossview and disable_crossview is None:
        disable_crossview = torch.zeros(B, dtype=torch.bool, device=dev)
    if model.enable_temporal and disable_temporal is None:
        disable_temporal = torch.zeros(B, dtype=torch.bool, device=dev)

    residuals: List[torch.Tensor] = []
    if model.condition_image_adapter is not None and condition_image_tensor is not None:            # :459-462
        ad = model.condition_image_adapter
        residuals = list(AdapterFn.apply(ad, condition_image_tensor, *_params(ad)))
        for f in residuals:
            if f.shape != h.shape:
                raise RuntimeError(f"condition residual {tuple(f.shape)} does not match hidden states {tuple(h.shape)}")

    for i, block in enumerate(model.transformer_blocks):
        if residuals:
            h = AddFn.apply(h, residuals.pop(0))                                                   # :491-494
        h, c = joint_block_train(block, h, c, st, I)
        if model.enable_temporal and i in model.temporal_block_layers:
            k = model.temporal_block_layers.index(i)
            idx = torch.arange(Tn, device=dev).view(1, Tn, 1).expand(B, Tn, V)
            seq = ops.timestep_sinusoid(idx, D)
            use_cam = model.enable_crossview and not model.disable_view_emb_on_temporal_module and view_cam_emb is not None
            seq_emb = mlp_train(model.time_pos_embeds[k], seq, res=view_cam_emb if use_cam else None)
            rpe = N
            if model.enable_crossview and not model.disable_view_emb_on_temporal_module:           # :559-566
                seq_emb, rpe = with_rays(seq_emb)
            tt = model.temporal_attention_type
            mk = ops.rowmap_temporal_full if tt == "full" else \
                ops.rowmap_temporal_rowwise if tt == "rowwise" else ops.rowmap_temporal_pointwise
            alpha = alpha_train(model.time_mixers[k], disable_temporal, B)
            h = vt_block_train(model.temporal_transformer_blocks[k], h, mk(B, Tn, V, height, width), seq_emb, rpe,
                               alpha, Tn * V * N)
        if model.enable_crossview and i in model.crossview_block_layers:
            k = model.crossview_block_layers.index(i)
            idx = torch.arange(V, device=dev).view(1, 1, V).expand(B, Tn, V)
            vemb = mlp_train(model.view_pos_embeds[k], ops.timestep_sinusoid(idx, D), res=view_cam_emb)
            vemb, rpe = with_rays(vemb)                                                             # :528-537
            ct = model.crossview_attention_type
            gmask = dmask = None
            if ct == "rowwise":
                rm = ops.rowmap_crossview_rowwise(B, Tn, V, height, width)
                gmask = crossview_attention_mask
            elif ct == "full":
                rm = ops.rowmap_crossview_full(B, Tn, V, height, width)
                dmask = crossview_attention_mask
            else:
                raise NotImplementedError(f"Not support {ct}")
            alpha = alpha_train(model.view_mixers[k], disable_crossview, B)
            h = vt_block_train(model.crossview_transformer_blocks[k], h, rm, vemb, rpe, alpha, Tn * V * N,
                               group_mask=gmask, dense_mask=dmask)

    geom = (I, model.out_channels, height, width, p, N, D)
    out = OutFn.apply(model, geom, h, st, *(_params(model.norm_out) + _params(model.proj_out)))
    return out.view(B, Tn, V, model.out_channels, height * p, width * p)


# ------------------------------------------------------------------------------------------ optimizer
class AdamW(torch.optim.Optimizer):
    """torch.optim.AdamW semantics with the update done by the HIP kernel (`dwm_adamw`), which also refreshes the bf16
    compute shadows of blocks.STORE.

    It IS a torch.optim.Optimizer: parameter groups, the (sparse) per-parameter state with its own step count, and the
    state-dict format are torch's bookkeeping, so `optimizer/<step>.pth` files interchange with the reference's
    torch.optim.AdamW (src/dwm/distributed.py:7-70) - including the warm-up configs whose `freezing_pattern` leaves frozen
    parameters in the list (ctsd.py:1014-1022, 1089-1092: the optimizer is built from ALL `model_wrapper.parameters()`) -
    and torch LR schedulers attach to it (ctsd.py:1098-1100, 1434-1435)."""

    def __init__(self, params, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False,
                        foreach=None, capturable=False, differentiable=False, fused=None)
        super().__init__(params, defaults)

    # group-0 hyper-parameters (the trainer has one group) and the step count, for callers and tests
    lr = property(lambda self: self.param_groups[0]["lr"])
    betas = property(lambda self: tuple(self.param_groups[0]["betas"]))
    eps = property(lambda self: self.param_groups[0]["eps"])
    weight_decay = property(lambda self: self.param_groups[0]["weight_decay"])

    @property
    def t(self) -> int:
        """largest per-parameter step count"""
        return max((int(float(st["step"])) for st in self.state.values() if "step" in st), default=0)

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            if group.get("amsgrad") or group.get("maximize"):
                raise NotImplementedError("AdamW: amsgrad / maximize")
            b1, b2 = group["betas"]
            batches: dict = {}                  # step count -> lists for ONE dwm_adamw_multi launch (normally a single batch)
            for p in group["params"]:
                if p.grad is None:              # frozen, or unused in this step: no state, no step (as torch)
                    continue
                if p.dtype != torch.float32:
                    raise RuntimeError("AdamW: fp32 master parameters expected")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                g = p.grad if p.grad.dtype == torch.float32 else p.grad.float()
                shadow = STORE.bf(p) if (p.is_cuda and p.numel() % 4 == 0 and p.is_contiguous()) else None
                if shadow is None:
                    STORE._shadow.pop(id(p), None)      # re-cast on next use
                step = int(st["step"].item())
                if p.is_contiguous() and st["exp_avg"].is_contiguous() and st["exp_avg_sq"].is_contiguous():
                    b = batches.setdefault(step, ([], [], [], [], []))
                    for lst, t in zip(b, (p.data, g.contiguous(), st["exp_avg"], st["exp_avg_sq"], shadow)):
                        lst.append(t)
                else:
                    T.adamw_(p.data, g.contiguous(), st["exp_avg"], st["exp_avg_sq"], shadow, lr=float(group["lr"]), beta1=b1, beta2=b2,
                             eps=group["eps"], weight_decay=group["weight_decay"], step=step, grad_scale=grad_scale)
            for step, (ps, gs, ms, vs, shs) in batches.items():
                T.adamw_multi_(ps, gs, ms, vs, shs, lr=float(group["lr"]), beta1=b1, beta2=b2, eps=group["eps"],
                               weight_decay=group["weight_decay"], step=step, grad_scale=grad_scale)
        STORE.bump(keep_shadows=True)
        return loss
