"""Module shells of the CTSD MMDiT.  They own the parameters under exactly the names the
reference module tree gives them (so released checkpoints load) and run
their arithmetic through libdwm_hip.so (opendwm_amd.ops).  Class names mirror the
reference / diffusers classes they stand in for:

  JointTransformerBlock   diffusers.models.attention.JointTransformerBlock (0.31.0)
  VTSelfAttentionBlock    dwm.models.crossview_temporal.VTSelfAttentionBlock
                          (src/dwm/models/crossview_temporal.py:536-582)
  AlphaBlender            dwm.models.crossview_temporal.AlphaBlender (:9-72)
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from . import ops
from .ops import ACT_GELU_TANH, ACT_SILU, EPI_GEGLU, EPI_RESID, EPI_RMSHEAD

bf16 = torch.bfloat16


class ParamStore:
    """bf16 compute copies of (fp32 master) parameters, keyed by parameter identity.  A copy is
    refreshed when the parameter's storage / autograd version changes or after `bump()` (the
    optimizer step of opendwm_amd.train writes the new bf16 values straight into the shadows and
    then calls `bump(keep_shadows=True)` so only derived tensors - fused / transposed weights - are
    rebuilt)."""

    def __init__(self):
        self.step = 0
        self._shadow = {}       # id(param) -> (key, bf16 tensor)
        self._derived = {}      # (id(param), tag) -> (step, tensor)
        self.precision = bf16   # compute dtype of the forward in flight: bf16, or torch.float32 (the accuracy path)

    def set_precision(self, dtype: torch.dtype) -> None:
        """bf16 (default) or fp32 compute copies; packed / derived tensors are per precision, so a switch rebuilds them"""
        if dtype not in (bf16, torch.float32):
            raise ValueError("compute dtype must be torch.bfloat16 or torch.float32")
        self.precision = dtype

    def cached(self, owner, make):
        """packed tensors of a module (`make()` -> dict), one entry per compute precision, rebuilt after an optimizer step /
        state-dict load (`bump`) or when `owner._pk` was reset to None"""
        slot = getattr(owner, "_pk", None)
        if not isinstance(slot, dict) or "_by_precision" not in slot:
            slot = {"_by_precision": {}}
            owner._pk = slot
        hit = slot["_by_precision"].get(self.precision)
        if hit is None or hit[0] != self.step:
            hit = (self.step, make())
            slot["_by_precision"][self.precision] = hit
        return hit[1]

    def bf(self, t: torch.Tensor) -> torch.Tensor:
        """the compute copy of a parameter: bf16, or fp32 while the fp32 accuracy path runs"""
        if self.precision == torch.float32:
            d = t.detach()
            if d.dtype == torch.float32:
                return d if d.is_contiguous() else d.contiguous()
            key = (t.data_ptr(), t._version, tuple(t.shape), "f32")
            hit = self._shadow.get(id(t))
            if hit is None or hit[0] != key:
                hit = (key, d.float().contiguous())
                self._shadow[id(t)] = hit
            return hit[1]
        if t.dtype == bf16:
            d = t.detach()
            return d if d.is_contiguous() else d.contiguous()
        key = (t.data_ptr(), t._version, tuple(t.shape))
        hit = self._shadow.get(id(t))
        if hit is None or hit[0] != key:
            d = t.detach()
            if d.is_cuda and d.dtype == torch.float32 and d.is_contiguous() and d.numel() % 4 == 0:
                sh = ops.cast_bf16(d)
            else:
                sh = d.to(bf16).contiguous()
            hit = (key, sh)
            self._shadow[id(t)] = hit
        return hit[1]

    def derived(self, t: torch.Tensor, tag: str, make):
        """cache of a tensor derived from parameter(s) (transpose, fused qkv, ...) valid for one optimizer step"""
        k = (id(t), tag, self.precision)
        hit = self._derived.get(k)
        if hit is None or hit[0] != (self.step, t.data_ptr(), t._version):
            hit = ((self.step, t.data_ptr(), t._version), make())
            self._derived[k] = hit
        return hit[1]

    def bump(self, keep_shadows: bool = False) -> None:
        self.step += 1
        self._derived.clear()
        ops.clear_split_weights()           # operand planes of the fp32 path: keyed on tensors that are rebuilt now
        if not keep_shadows:
            self._shadow.clear()


STORE = ParamStore()
# inference: fold the softmax scale and log2(e) into the q RMSNorm weights (Attention.packed "rms_ps") and tell the attention kernels so
PRESCALE_Q = True
# dwm_attn_args.variant bits OR-ed into the inference blocks' attention calls (kernel selection; 0 = the library's choice)
ATTN_VARIANT = 0


def _bf(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    return None if t is None else STORE.bf(t)


def stream32(h: torch.Tensor) -> bool:
    """is `h` the fp32 residual stream of a bf16 forward?  (In the fp32 accuracy path every tensor is fp32 and the ops
    dispatch on the dtype; here only the hidden / context streams are: the residual adds - ~130 per forward, each a rounding
    of the whole stream when it is kept in bf16 - accumulate in fp32, everything a GEMM reads stays bf16.)"""
    return h.dtype == torch.float32 and STORE.precision == bf16


def _act_like(h: torch.Tensor) -> torch.Tensor:
    """an activation buffer shaped like the stream h, in the compute dtype"""
    return torch.empty(h.shape, dtype=bf16 if stream32(h) else h.dtype, device=h.device)


def _resid_into(stream: torch.Tensor, a: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], **kw) -> torch.Tensor:
    """stream <- stream + gate * (a @ w^T + b): the RESID GEMM in place on a residual stream (bf16, or the fp32 stream)"""
    if stream32(stream):
        return ops.gemm(a, w, b, epilogue=EPI_RESID, res=stream, out32=stream, mirror=False, **kw)
    return ops.gemm(a, w, b, epilogue=EPI_RESID, res=stream, out=stream, **kw)


def geglu_pack(w: torch.Tensor) -> torch.Tensor:
    """Reorder the rows of a GEGLU projection ([value rows ; gate rows]) into 64-row groups
    [32 value rows | 32 gate rows] — the layout DWM_EPI_GEGLU expects."""
    n2 = w.shape[0] // 2
    if n2 % 32 != 0:
        raise RuntimeError("GEGLU inner dim must be a multiple of 32")
    tail = w.shape[1:]
    v = w[:n2].reshape(n2 // 32, 32, *tail)
    g = w[n2:].reshape(n2 // 32, 32, *tail)
    return torch.stack([v, g], dim=1).reshape(w.shape).contiguous()


class RMSNorm(nn.Module):
    """Parameter holder for diffusers RMSNorm(dim, eps, elementwise_affine=True); applied
    inside the q/k projection GEMM epilogue."""

    def __init__(self, dim: int, eps: float):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))


class _ActProj(nn.Module):
    """diffusers GELU / GEGLU activation module: holds `.proj`."""

    def __init__(self, dim_in: int, dim_out: int):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out)


class FeedForward(nn.Module):
    """diffusers FeedForward(dim, dim_out, mult=4, activation_fn): keys net.0.proj.*, net.2.*"""

    def __init__(self, dim: int, dim_out: Optional[int] = None, activation_fn: str = "geglu"):
        super().__init__()
        inner = dim * 4
        self.activation_fn = activation_fn
        proj_out = inner * 2 if activation_fn == "geglu" else inner
        self.net = nn.ModuleList([_ActProj(dim, proj_out), nn.Identity(), nn.Linear(inner, dim_out or dim)])


class TimestepEmbedding(nn.Module):
    """diffusers TimestepEmbedding(in_channels, time_embed_dim, out_dim): linear_2(silu(linear_1(x)))."""

    def __init__(self, in_channels: int, time_embed_dim: int, out_dim: Optional[int] = None):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.linear_2 = nn.Linear(time_embed_dim, out_dim or time_embed_dim)

    def run(self, x: torch.Tensor, res: Optional[torch.Tensor] = None) -> torch.Tensor:
        h = ops.gemm(x, _bf(self.linear_1.weight), _bf(self.linear_1.bias), act=ACT_SILU)
        if res is None:
            return ops.gemm(h, _bf(self.linear_2.weight), _bf(self.linear_2.bias))
        return ops.gemm(h, _bf(self.linear_2.weight), _bf(self.linear_2.bias), epilogue=EPI_RESID, res=res)


class CombinedTimestepTextProjEmbeddings(nn.Module):
    """diffusers CombinedTimestepTextProjEmbeddings(embedding_dim, pooled_projection_dim)."""

    def __init__(self, embedding_dim: int, pooled_projection_dim: int):
        super().__init__()
        self.timestep_embedder = TimestepEmbedding(256, embedding_dim)
        self.text_embedder = TimestepEmbedding(pooled_projection_dim, embedding_dim)   # PixArtAlphaTextProjection: same keys

    def run(self, timestep: torch.Tensor, pooled: torch.Tensor) -> torch.Tensor:
        t_emb = self.timestep_embedder.run(ops.timestep_sinusoid(timestep, 256, dtype=STORE.precision))
        return self.text_embedder.run(pooled, res=t_emb)


class _AdaNorm(nn.Module):
    """AdaLayerNormZero / SD35AdaLayerNormZeroX / AdaLayerNormContinuous: holds `.linear`."""

    def __init__(self, dim: int, chunks: int):
        super().__init__()
        self.linear = nn.Linear(dim, chunks * dim)


class Attention(nn.Module):
    """Parameter holder for diffusers Attention (q/k/v/out projections, optional added kv
    projections and qk RMSNorms).  `packed()` returns the fused projection weights."""

    def __init__(self, dim: int, heads: int, dim_head: int, bias: bool, out_bias: bool = True,
                 added_kv: bool = False, context_pre_only: Optional[bool] = None,
                 qk_norm: Optional[str] = None, eps: float = 1e-5):
        super().__init__()
        self.heads, self.dim_head, self.eps = heads, dim_head, eps
        inner = heads * dim_head
        self.to_q = nn.Linear(dim, inner, bias=bias)
        self.to_k = nn.Linear(dim, inner, bias=bias)
        self.to_v = nn.Linear(dim, inner, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(inner, dim, bias=out_bias), nn.Identity()])
        self.has_qk_norm = qk_norm == "rms_norm"
        if self.has_qk_norm:
            self.norm_q = RMSNorm(dim_head, eps)
            self.norm_k = RMSNorm(dim_head, eps)
        self.added_kv = added_kv
        if added_kv:
            self.add_q_proj = nn.Linear(dim, inner, bias=True)
            self.add_k_proj = nn.Linear(dim, inner, bias=True)
            self.add_v_proj = nn.Linear(dim, inner, bias=True)
            if not context_pre_only:
                self.to_add_out = nn.Linear(inner, dim, bias=True)
            if self.has_qk_norm:
                self.norm_added_q = RMSNorm(dim_head, eps)
                self.norm_added_k = RMSNorm(dim_head, eps)
        self._pk = None
        self._pk_step = -1

    def packed(self) -> dict:
        def make():
            def fuse(q, k, v):
                w = torch.cat([_bf(q.weight), _bf(k.weight), _bf(v.weight)], 0).contiguous()
                b = None if q.bias is None else torch.cat([_bf(q.bias), _bf(k.bias), _bf(v.bias)]).contiguous()
                return w, b
            pk = {}
            pk["wqkv"], pk["bqkv"] = fuse(self.to_q, self.to_k, self.to_v)
            # "rms": the q / k RMSNorm weights per column of the fused projection; "rms_ps" (inference): the same with the softmax scale
            # and log2(e) folded into the q columns, so that the attention kernels take the scores as log2-domain without rescaling
            # their Q fragments (dwm_attn_args.variant bit 15; one rounding of q instead of two).  Training keeps "rms".
            def rms_cols(nq, nk, fold):
                wq = _bf(nq.weight)
                if fold:        # a temporary: converted here, NOT through STORE.bf (its shadow cache is keyed on the tensor's identity)
                    wq = (nq.weight.detach().float() * (self.dim_head ** -0.5 * 1.4426950408889634)).to(wq.dtype)
                return torch.cat([wq.repeat(self.heads), _bf(nk.weight).repeat(self.heads)]).contiguous()
            if self.has_qk_norm:
                pk["rms"] = rms_cols(self.norm_q, self.norm_k, False)
                pk["rms_ps"] = rms_cols(self.norm_q, self.norm_k, True)
            if self.added_kv:
                pk["wadd"], pk["badd"] = fuse(self.add_q_proj, self.add_k_proj, self.add_v_proj)
                if self.has_qk_norm:
                    pk["rms_add"] = rms_cols(self.norm_added_q, self.norm_added_k, False)
                    pk["rms_add_ps"] = rms_cols(self.norm_added_q, self.norm_added_k, True)
            return pk
        return STORE.cached(self, make)

    @property
    def attn_variant(self) -> int:
        """dwm_attn_args.variant bits that go with what `project_qkv` produces: bit 15 when q leaves it pre-scaled"""
        return ops.ATTN_Q_PRESCALED if (self.has_qk_norm and PRESCALE_Q) else 0

    def project_qkv(self, x: torch.Tensor, added: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x [rows, dim] -> fused [rows, 3*inner] with q,k RMS-normalised per head (q scaled by head_dim^-1/2 log2(e) as well when
        `attn_variant` says so: pass that to ops.attention).  `out`: the [rows, 3*inner] buffer to fill."""
        pk = self.packed()
        ps = "_ps" if self.attn_variant else ""
        w, b, rms = (pk["wadd"], pk["badd"], pk.get("rms_add" + ps)) if added else (pk["wqkv"], pk["bqkv"], pk.get("rms" + ps))
        if rms is None:
            return ops.gemm(x, w, b, out=out)
        if self.dim_head != 64:
            raise RuntimeError("qk RMSNorm is implemented for head_dim 64 only")
        return ops.gemm(x, w, b, epilogue=EPI_RMSHEAD, rms_w=rms, rms_ncols=rms.numel(), rms_eps=self.eps, out=out)


class JointTransformerBlock(nn.Module):
    """diffusers JointTransformerBlock(dim, heads, head_dim, context_pre_only, qk_norm,
    use_dual_attention); forward restated in SURVEY.md Appendix A.3."""

    def __init__(self, dim: int, num_attention_heads: int, attention_head_dim: int,
                 context_pre_only: bool = False, qk_norm: Optional[str] = None,
                 use_dual_attention: bool = False):
        super().__init__()
        self.dim, self.heads = dim, num_attention_heads
        self.context_pre_only = context_pre_only
        self.use_dual_attention = use_dual_attention
        self.norm1 = _AdaNorm(dim, 9 if use_dual_attention else 6)
        self.norm1_context = _AdaNorm(dim, 2 if context_pre_only else 6)
        self.attn = Attention(dim, num_attention_heads, attention_head_dim, bias=True, added_kv=True,
                              context_pre_only=context_pre_only, qk_norm=qk_norm, eps=1e-6)
        if use_dual_attention:
            self.attn2 = Attention(dim, num_attention_heads, attention_head_dim, bias=True,
                                   qk_norm=qk_norm, eps=1e-6)
        self.ff = FeedForward(dim, dim, activation_fn="gelu-approximate")
        if not context_pre_only:
            self.ff_context = FeedForward(dim, dim, activation_fn="gelu-approximate")

    def run(self, h: torch.Tensor, c: torch.Tensor, silu_temb: torch.Tensor, n_img: int, mod=None, cmod=None):
        """h [I*N, D], c [I*Lc, D] (updated in place): bf16, or the fp32 residual streams of a bf16 forward (`stream32`);
        silu_temb [I, D].  mod / cmod: this block's AdaLN modulation rows [I, 6 D | 9 D] / [I, 6 D | 2 D] when the caller has
        computed them already (column slices of one stacked launch, dit.stack_modulation).  Returns (c, h)."""
        D = self.dim
        N, Lc = h.shape[0] // n_img, c.shape[0] // n_img
        x32 = stream32(h)
        if mod is None:
            mod = ops.gemm(silu_temb, _bf(self.norm1.linear.weight), _bf(self.norm1.linear.bias))
        if cmod is None:
            cmod = ops.gemm(silu_temb, _bf(self.norm1_context.linear.weight), _bf(self.norm1_context.linear.bias))
        sl = lambda m, i: m[:, i * D:(i + 1) * D]
        # AdaLayerNormZero(X) chunk order: shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp[, shift2, scale2, gate2]
        nh2 = _act_like(h) if self.use_dual_attention else None
        nh = ops.layernorm(h, eps=1e-6, scale=sl(mod, 1), shift=sl(mod, 0), rows_per_mod=N,
                           scale2=sl(mod, 7) if nh2 is not None else None,
                           shift2=sl(mod, 6) if nh2 is not None else None, out2=nh2, x32=x32)
        if self.context_pre_only:      # AdaLayerNormContinuous: scale first
            nc = ops.layernorm(c, eps=1e-6, scale=sl(cmod, 0), shift=sl(cmod, 1), rows_per_mod=Lc, x32=x32)
        else:
            nc = ops.layernorm(c, eps=1e-6, scale=sl(cmod, 1), shift=sl(cmod, 0), rows_per_mod=Lc, x32=x32)

        # the two segments of q / k / v and of the attention output in ONE allocation each: the streaming attention kernel folds the
        # distance between the segments into 32-bit row offsets while it fits (+-16 GiB) - its fastest form; pairs further apart,
        # which a caching allocator does hand out, take the form that adds the displacement per row (same bits, a few more
        # instructions per key step)
        rows_h, rows_c = nh.shape[0], nc.shape[0]
        both = torch.empty(rows_h + rows_c, 3 * D, dtype=nh.dtype, device=nh.device)
        qkv = self.attn.project_qkv(nh, out=both[:rows_h])
        cqkv = self.attn.project_qkv(nc, added=True, out=both[rows_h:])
        ao_both = torch.empty(rows_h + rows_c, D, dtype=nh.dtype, device=nh.device)
        ao, cao = ao_both[:rows_h], ao_both[rows_h:]
        ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], ao, ops.rowmap_identity(n_img, N), self.heads,
                      q1=cqkv[:, :D], k1=cqkv[:, D:2 * D], v1=cqkv[:, 2 * D:], out1=cao, variant=self.attn.attn_variant | ATTN_VARIANT)
        to_out = self.attn.to_out[0]
        _resid_into(h, ao, _bf(to_out.weight), _bf(to_out.bias), gate=sl(mod, 2), rows_per_gate=N)
        if self.use_dual_attention:
            qkv2 = self.attn2.project_qkv(nh2)
            ops.attention(qkv2[:, :D], qkv2[:, D:2 * D], qkv2[:, 2 * D:], ao, ops.rowmap_identity(n_img, N), self.heads,
                          variant=self.attn2.attn_variant | ATTN_VARIANT)
            to_out2 = self.attn2.to_out[0]
            _resid_into(h, ao, _bf(to_out2.weight), _bf(to_out2.bias), gate=sl(mod, 8), rows_per_gate=N)
        nh = ops.layernorm(h, eps=1e-6, scale=sl(mod, 4), shift=sl(mod, 3), rows_per_mod=N, out=nh, x32=x32)
        f1, f2 = self.ff.net[0].proj, self.ff.net[2]
        ffh = ops.gemm(nh, _bf(f1.weight), _bf(f1.bias), act=ACT_GELU_TANH)
        _resid_into(h, ffh, _bf(f2.weight), _bf(f2.bias), gate=sl(mod, 5), rows_per_gate=N)

        if self.context_pre_only:
            return None, h
        ta = self.attn.to_add_out
        _resid_into(c, cao, _bf(ta.weight), _bf(ta.bias), gate=sl(cmod, 2), rows_per_gate=Lc)
        nc = ops.layernorm(c, eps=1e-6, scale=sl(cmod, 4), shift=sl(cmod, 3), rows_per_mod=Lc, out=nc, x32=x32)
        c1, c2 = self.ff_context.net[0].proj, self.ff_context.net[2]
        cff = ops.gemm(nc, _bf(c1.weight), _bf(c1.bias), act=ACT_GELU_TANH)
        _resid_into(c, cff, _bf(c2.weight), _bf(c2.bias), gate=sl(cmod, 5), rows_per_gate=Lc)
        return c, h


class AlphaBlender(nn.Module):
    """crossview_temporal.py:9-72.  The blend itself is fused into the last GEMM of the
    VT block; this module owns `mix_factor` and computes alpha[b]."""

    strategies = ["fixed", "learned", "learned_with_images"]

    def __init__(self, alpha: float, merge_strategy: str = "learned_with_images"):
        super().__init__()
        if merge_strategy not in AlphaBlender.strategies:
            raise ValueError("merge_strategy needs to be in {}".format(AlphaBlender.strategies))
        self.merge_strategy = merge_strategy
        if merge_strategy == "fixed":
            self.register_buffer("mix_factor", torch.tensor([float(alpha)]))
        else:
            self.register_parameter("mix_factor", nn.Parameter(torch.tensor([float(alpha)])))

    def get_alpha(self, image_only_indicator: Optional[torch.Tensor], batch: int) -> torch.Tensor:
        """fp32 alpha[batch] (crossview_temporal.py:33-51).  Read-only for the caller: the tensor is kept and handed out again while
        neither the mix factor (its storage, its version, the optimizer step: the HIP AdamW writes through the raw pointer) nor
        the indicator (the same tensor object at the same version; held here, so its address cannot be reused) changed - the
        UNet asks 54 times per denoise step, five tiny launches each.  A write through `.data` (EMA copy, weight surgery, an
        optimizer that is not this package's) changes none of those: call `blocks.STORE.bump()` (or the model's
        `_invalidate_packed()`, which `load_state_dict` / `.to()` do) after it, as for every packed weight copy."""
        mfp, ind = self.mix_factor, image_only_indicator
        if mfp.is_cuda and torch.cuda.is_current_stream_capturing():
            return self._alpha(ind, batch)      # a captured graph derives alpha from the indicator on every replay, as before
        key = (mfp.data_ptr(), mfp._version, STORE.step, batch, None if ind is None else ind._version)
        c = getattr(self, "_alpha_cache", None)
        if c is not None and c[0] == key and c[1] is ind:
            return c[2]
        a = self._alpha(ind, batch)
        self._alpha_cache = (key, ind, a)
        return a

    def _alpha(self, image_only_indicator: Optional[torch.Tensor], batch: int) -> torch.Tensor:
        mf = self.mix_factor.detach().float()
        if self.merge_strategy == "fixed":
            return mf.expand(batch).contiguous()
        if self.merge_strategy == "learned":
            return torch.sigmoid(mf).expand(batch).contiguous()
        if image_only_indicator is None:
            raise ValueError("Please provide image_only_indicator to use learned_with_images merge strategy")
        flag = image_only_indicator.reshape(batch).to(device=mf.device, dtype=torch.bool)
        return torch.where(flag, torch.ones((), device=mf.device), torch.sigmoid(mf)).contiguous()


class VTSelfAttentionBlock(nn.Module):
    """crossview_temporal.py:536-582: norm_in -> GEGLU ff_in (+res) -> norm1 -> attn1 (+res)
    -> norm3 -> GEGLU ff (+res).  All row-wise ops run in the caller's token order; only
    the attention kernel sees the rearranged (problem, token) view through a RowMap."""

    def __init__(self, dim: int, time_mix_inner_dim: int, num_attention_heads: int,
                 attention_head_dim: int, qk_norm=None):
        super().__init__()
        if dim != time_mix_inner_dim:
            raise NotImplementedError("VTSelfAttentionBlock: dim != time_mix_inner_dim is never used by the reference")
        self.dim, self.heads = dim, num_attention_heads
        self.is_res = True
        self.norm_in = nn.LayerNorm(dim)
        self.ff_in = FeedForward(dim, dim_out=time_mix_inner_dim, activation_fn="geglu")
        self.norm1 = nn.LayerNorm(time_mix_inner_dim)
        self.attn1 = Attention(time_mix_inner_dim, num_attention_heads, attention_head_dim, bias=False,
                               qk_norm=qk_norm, eps=1e-5)
        self.norm3 = nn.LayerNorm(time_mix_inner_dim)
        self.ff = FeedForward(time_mix_inner_dim, activation_fn="geglu")
        self._pk = None
        self._pk_step = -1

    def packed(self) -> dict:
        def make():
            pk = {}
            for name, ff in (("ff_in", self.ff_in), ("ff", self.ff)):
                pk[name + "_w"] = geglu_pack(_bf(ff.net[0].proj.weight))
                pk[name + "_b"] = geglu_pack(_bf(ff.net[0].proj.bias))
            return pk
        return STORE.cached(self, make)

    def run(self, h: torch.Tensor, rowmap: ops.RowMap, *, emb: Optional[torch.Tensor] = None,
            rows_per_emb: int = 1, group_mask: Optional[torch.Tensor] = None,
            dense_mask: Optional[torch.Tensor] = None,
            blend_alpha: Optional[torch.Tensor] = None, rows_per_alpha: int = 1,
            blend_into: Optional[torch.Tensor] = None) -> torch.Tensor:
        """h [rows, D] bf16 (or the fp32 residual stream of a bf16 forward, `stream32`: the block's own stream x is then fp32
        as well).  x = h + emb[row // rows_per_emb]; y = block(x); if blend_alpha is
        given the result alpha*blend_into + (1-alpha)*y is written into blend_into (the mixer of
        crossview_temporal_dit.py:320-327 / :363-370)."""
        D = self.dim
        pk = self.packed()
        x32 = stream32(h)
        ln = lambda x, n, **kw: ops.layernorm(x, eps=1e-5, weight=_bf(n.weight), bias=_bf(n.bias), x32=stream32(x), **kw)
        if emb is not None:
            xs = torch.empty_like(h)
            y = ln(h, self.norm_in, addvec=emb, rows_per_add=rows_per_emb, xsum=xs)
        else:
            xs = h.clone()
            y = ln(h, self.norm_in)
        g = ops.gemm(y, pk["ff_in_w"], pk["ff_in_b"], epilogue=EPI_GEGLU)
        l2 = self.ff_in.net[2]
        _resid_into(xs, g, _bf(l2.weight), _bf(l2.bias))

        y = ln(xs, self.norm1, out=y)
        qkv = self.attn1.project_qkv(y)
        ao = y     # norm1 output is dead once qkv exists
        ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], ao, rowmap, self.heads,
                      group_mask=group_mask, dense_mask=dense_mask, variant=self.attn1.attn_variant | ATTN_VARIANT)
        to_out = self.attn1.to_out[0]
        _resid_into(xs, ao, _bf(to_out.weight), _bf(to_out.bias))

        y = ln(xs, self.norm3, out=y)
        g = ops.gemm(y, pk["ff_w"], pk["ff_b"], epilogue=EPI_GEGLU, out=g)
        l2 = self.ff.net[2]
        if blend_alpha is None:
            return _resid_into(xs, g, _bf(l2.weight), _bf(l2.bias))
        if x32:
            if not stream32(blend_into):
                raise RuntimeError("VTSelfAttentionBlock.run: an fp32 stream blends into an fp32 stream")
            ops.gemm(g, _bf(l2.weight), _bf(l2.bias), epilogue=EPI_RESID, res=xs, blend=blend_into,
                     alpha=blend_alpha, rows_per_alpha=rows_per_alpha, out32=blend_into, mirror=False)
            return blend_into
        ops.gemm(g, _bf(l2.weight), _bf(l2.bias), epilogue=EPI_RESID, res=xs, blend=blend_into,
                 alpha=blend_alpha, rows_per_alpha=rows_per_alpha, out=blend_into)
        return blend_into

    def forward(self, hidden_states: torch.Tensor, self_attention_mask: Optional[torch.Tensor] = None):
        """Reference signature: hidden_states [Bp, L, C], self_attention_mask bool [Bp, L, L]."""
        with torch.no_grad():
            bp, L, c = hidden_states.shape
            h = ops.cast_bf16(hidden_states.reshape(bp * L, c).contiguous()) \
                if hidden_states.dtype != bf16 else hidden_states.reshape(bp * L, c).contiguous()
            out = self.run(h, ops.rowmap_identity(bp, L), dense_mask=self_attention_mask)
            return out.view(bp, L, c)
