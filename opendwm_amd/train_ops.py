"""Thin wrappers over the training entry points of libdwm_hip.so (include/dwm_hip.h, "Training"
section) plus the two GEMM-shaped composites every linear layer's backward needs.  Same
conventions as opendwm_amd.ops: bf16 CUDA tensors, current stream, no fallback."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence, Tuple

import torch

from . import _lib, ops
from .ops import _p, _stream, ACT_GELU_TANH, ACT_SILU  # noqa: F401

bf16 = torch.bfloat16
# weight gradients by dwm_gemm_tn (operands as they are); "0": the transposes + NT GEMM path (A/B measurements)
WGRAD_TN = os.environ.get("DWM_WGRAD_TN", "1") != "0"


def _rows2d(t: torch.Tensor, name: str, dtype=bf16) -> None:
    if t.dtype != dtype or t.dim() != 2 or t.stride(1) != 1 or not t.is_cuda:
        raise RuntimeError(f"{name}: expected a {dtype} CUDA matrix with unit column stride, got {t.dtype} {tuple(t.shape)} {t.stride()}")


def transpose(x: torch.Tensor, rows_pad: Optional[int] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x [rows, cols] -> out [cols, rows_pad] (zero-filled beyond `rows`; default: rows rounded up to 64)."""
    _rows2d(x, "x")
    rows, cols = x.shape
    if rows_pad is None:
        rows_pad = (rows + 63) // 64 * 64
    if out is None:
        out = torch.empty((cols, rows_pad), dtype=bf16, device=x.device)
    _lib.check(_lib.load().dwm_transpose_bf16(_p(x), x.stride(0), rows, cols, _p(out), out.stride(0), rows_pad, _stream()),
               "dwm_transpose_bf16")
    return out


def segsum(a: torch.Tensor, b: Optional[torch.Tensor] = None, rows_per_group: Optional[int] = None,
           out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp32 [groups, ncols]: per-group column sums of a (* b); `out` (if given) is accumulated into."""
    _rows2d(a, "a")
    rows, ncols = a.shape
    rpg = rows if rows_per_group is None else rows_per_group
    groups = (rows + rpg - 1) // rpg
    if out is None:
        out = torch.zeros((groups, ncols), dtype=torch.float32, device=a.device)
    if b is not None:
        _rows2d(b, "b")
    _lib.check(_lib.load().dwm_segsum(_p(a), a.stride(0), _p(b), 0 if b is None else b.stride(0), rows, ncols, rpg,
                                      _p(out), out.stride(0), _stream()), "dwm_segsum")
    return out


def segsum_diff(a: torch.Tensor, b: torch.Tensor, b2: torch.Tensor, rows_per_group: Optional[int] = None) -> torch.Tensor:
    """fp32 [groups, ncols]: per-group column sums of a * (b - b2), the difference taken in fp32 before the product."""
    for t, n in ((a, "a"), (b, "b"), (b2, "b2")):
        _rows2d(t, n)
    if b.shape != a.shape or b2.shape != a.shape:
        raise RuntimeError("segsum_diff: shape mismatch")
    rows, ncols = a.shape
    rpg = rows if rows_per_group is None else rows_per_group
    out = torch.zeros(((rows + rpg - 1) // rpg, ncols), dtype=torch.float32, device=a.device)
    _lib.check(_lib.load().dwm_segsum_diff(_p(a), a.stride(0), _p(b), b.stride(0), _p(b2), b2.stride(0), rows, ncols, rpg,
                                           _p(out), out.stride(0), _stream()), "dwm_segsum_diff")
    return out


def act_fwd(x: torch.Tensor, act: int) -> torch.Tensor:
    y = torch.empty_like(x)
    _lib.check(_lib.load().dwm_act_fwd(_p(x), _p(y), x.numel(), act, _stream()), "dwm_act_fwd")
    return y


def act_bwd(x: torch.Tensor, dy: torch.Tensor, act: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    dx = torch.empty_like(x) if out is None else out
    _lib.check(_lib.load().dwm_act_bwd(_p(x), _p(dy), _p(dx), x.numel(), act, _stream()), "dwm_act_bwd")
    return dx


def geglu_fwd(u: torch.Tensor) -> torch.Tensor:
    _rows2d(u, "u")
    rows, two = u.shape
    g = torch.empty((rows, two // 2), dtype=bf16, device=u.device)
    _lib.check(_lib.load().dwm_geglu_fwd(_p(u), u.stride(0), rows, two // 2, _p(g), g.stride(0), _stream()), "dwm_geglu_fwd")
    return g


def geglu_bwd(u: torch.Tensor, dg: torch.Tensor) -> torch.Tensor:
    _rows2d(u, "u"); _rows2d(dg, "dg")
    rows, two = u.shape
    du = torch.empty_like(u)
    _lib.check(_lib.load().dwm_geglu_bwd(_p(u), u.stride(0), _p(dg), dg.stride(0), rows, two // 2, _p(du), du.stride(0),
                                         _stream()), "dwm_geglu_bwd")
    return du


def rowcombine(a: torch.Tensor, *, gate_a: Optional[torch.Tensor] = None, rows_per_gate_a: int = 1,
               coef_a: Optional[torch.Tensor] = None, rows_per_coef_a: int = 1,
               b: Optional[torch.Tensor] = None, coef_b: Optional[torch.Tensor] = None, rows_per_coef_b: int = 1,
               out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = a * gate_a[row // rpg] * coef_a[row // rpc] + b * coef_b[row // rpc] (see dwm_rowcombine)."""
    _rows2d(a, "a")
    if out is None:
        out = torch.empty((a.shape[0], a.shape[1]), dtype=bf16, device=a.device)
    r = _lib.RowCombineArgs()
    r.a, r.lda = _p(a), a.stride(0)
    if gate_a is not None:
        _rows2d(gate_a, "gate_a")
        r.gate_a, r.ld_gate_a, r.rows_per_gate_a = _p(gate_a), gate_a.stride(0), rows_per_gate_a
    if coef_a is not None:
        r.coef_a, r.rows_per_coef_a = _p(coef_a), rows_per_coef_a
    if b is not None:
        _rows2d(b, "b")
        r.b, r.ldb = _p(b), b.stride(0)
        if coef_b is not None:
            r.coef_b, r.rows_per_coef_b = _p(coef_b), rows_per_coef_b
    r.out, r.ldo, r.rows, r.ncols = _p(out), out.stride(0), a.shape[0], a.shape[1]
    _lib.check(_lib.load().dwm_rowcombine(C.byref(r), _stream()), "dwm_rowcombine")
    return out


def layernorm_bwd(x: torch.Tensor, dy: torch.Tensor, *, eps: float, dx: Optional[torch.Tensor] = None,
                  accumulate: bool = False, addvec: Optional[torch.Tensor] = None, rows_per_add: int = 1,
                  weight: Optional[torch.Tensor] = None, scale: Optional[torch.Tensor] = None,
                  scale2: Optional[torch.Tensor] = None, rows_per_mod: int = 0, dy2: Optional[torch.Tensor] = None,
                  dgamma: Optional[torch.Tensor] = None, dbeta: Optional[torch.Tensor] = None,
                  dgamma2: Optional[torch.Tensor] = None, dbeta2: Optional[torch.Tensor] = None,
                  grad_per_group: bool = False) -> torch.Tensor:
    """Backward of ops.layernorm; dgamma/dbeta[/2] are fp32 [G or 1, D] accumulators (see dwm_layernorm_bwd)."""
    _rows2d(x, "x"); _rows2d(dy, "dy")
    rows, D = x.shape
    if dx is None:
        dx = torch.empty((rows, D), dtype=bf16, device=x.device)
        accumulate = False
    a = _lib.LayerNormBwdArgs()
    a.x, a.ldx, a.dy, a.lddy, a.dx, a.lddx = _p(x), x.stride(0), _p(dy), dy.stride(0), _p(dx), dx.stride(0)
    a.accumulate, a.rows, a.D, a.eps = int(accumulate), rows, D, eps
    if addvec is not None:
        a.addvec, a.ld_add, a.rows_per_add = _p(addvec), addvec.stride(0), rows_per_add
    if dy2 is not None:
        a.dy2, a.lddy2 = _p(dy2), dy2.stride(0)
    a.weight = _p(weight)
    mod = scale if scale is not None else scale2
    if mod is not None:
        a.scale, a.scale2, a.ld_mod, a.rows_per_mod = _p(scale), _p(scale2), mod.stride(0), rows_per_mod
    elif grad_per_group:
        a.rows_per_mod = rows_per_mod
    g0 = next((g for g in (dgamma, dbeta, dgamma2, dbeta2) if g is not None), None)
    if g0 is not None:
        a.dgamma, a.dbeta, a.dgamma2, a.dbeta2 = _p(dgamma), _p(dbeta), _p(dgamma2), _p(dbeta2)
        a.ld_grad, a.grad_per_group = g0.stride(0), int(grad_per_group)
    _lib.check(_lib.load().dwm_layernorm_bwd(C.byref(a), _stream()), "dwm_layernorm_bwd")
    return dx


def rmsnorm_heads_train_(x: torch.Tensor, w_expanded: torch.Tensor, eps: float) -> torch.Tensor:
    """In-place per-head RMSNorm of x [rows, ncols]; returns rinv fp32 [rows, ncols // 64]."""
    _rows2d(x, "x")
    rows, ncols = x.shape
    rinv = torch.empty((rows, ncols // 64), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().dwm_rmsnorm_heads_train(_p(x), x.stride(0), rows, ncols, _p(w_expanded), eps, _p(rinv), _stream()),
               "dwm_rmsnorm_heads_train")
    return rinv


def rmsnorm_heads_bwd_(y: torch.Tensor, rinv: torch.Tensor, w_expanded: torch.Tensor, dy: torch.Tensor,
                       dw: torch.Tensor) -> torch.Tensor:
    """dy -> dx in place; dw fp32 [ncols] accumulated."""
    _rows2d(y, "y"); _rows2d(dy, "dy")
    rows, ncols = y.shape
    _lib.check(_lib.load().dwm_rmsnorm_heads_bwd(_p(y), y.stride(0), _p(rinv), _p(w_expanded), _p(dy), dy.stride(0), rows, ncols,
                                                 _p(dw), _stream()), "dwm_rmsnorm_heads_bwd")
    return dy


def adamw_(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, p_bf16: Optional[torch.Tensor], *,
           lr: float, beta1: float, beta2: float, eps: float, weight_decay: float, step: int, grad_scale: float = 1.0) -> None:
    for t in (p, g, m, v):
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise RuntimeError("adamw_: fp32 contiguous tensors expected")
    _lib.check(_lib.load().dwm_adamw(_p(p), _p(g), _p(m), _p(v), _p(p_bf16), p.numel(), lr, beta1, beta2, eps, weight_decay,
                                     1.0 - beta1 ** step, 1.0 - beta2 ** step, grad_scale, _stream()), "dwm_adamw")


ADAMW_CHUNK = 1 << 16            # elements per workgroup of the multi-tensor AdamW
_ADAMW_BLOCKS: dict = {}         # (device, tuple of numels) -> (block_item, block_start) on the device: static per parameter list


def adamw_multi_(ps, gs, ms, vs, shadows, *, lr: float, beta1: float, beta2: float, eps: float, weight_decay: float, step: int,
                 grad_scale: float = 1.0) -> None:
    """dwm_adamw_multi: one launch for the whole list (fp32 contiguous p / g / m / v of equal numel per entry; shadows: bf16 copy or
    None).  All entries share the hyper-parameters and `step`."""
    if not ps:
        return
    dev = ps[0].device
    rows = []
    for p, g, m, v, sh in zip(ps, gs, ms, vs, shadows):
        for t in (p, g, m, v):
            if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != p.numel() or t.device != dev:
                raise RuntimeError("adamw_multi_: fp32 contiguous tensors of one shape on one device expected")
        rows.append((p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), 0 if sh is None else sh.data_ptr(), p.numel()))
    items = torch.tensor(rows, dtype=torch.int64).to(dev, non_blocking=True)          # [n, 6] = dwm_adamw_item[n]
    key = (dev.index, tuple(r[5] for r in rows))
    tab = _ADAMW_BLOCKS.get(key)
    if tab is None:
        bi, bs = [], []
        for i, r in enumerate(rows):
            nb = (r[5] + ADAMW_CHUNK - 1) // ADAMW_CHUNK
            bi.append(torch.full((nb,), i, dtype=torch.int32))
            bs.append(torch.arange(nb, dtype=torch.int64) * ADAMW_CHUNK)
        _ADAMW_BLOCKS.clear()                                                           # one parameter list at a time
        tab = _ADAMW_BLOCKS[key] = (torch.cat(bi).to(dev), torch.cat(bs).to(dev))
    _lib.check(_lib.load().dwm_adamw_multi(_p(items), _p(tab[0]), _p(tab[1]), tab[0].numel(), ADAMW_CHUNK, lr, beta1, beta2, eps,
                                           weight_decay, 1.0 - beta1 ** step, 1.0 - beta2 ** step, grad_scale, _stream()),
               "dwm_adamw_multi")


def cast_f32(x: torch.Tensor, out: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    """fp32 (+)= bf16 matrix / vector."""
    x2 = x if x.dim() == 2 else x.reshape(1, -1)
    _rows2d(x2, "x")
    if out is None:
        out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        accumulate = False
    o2 = out if out.dim() == 2 else out.reshape(1, -1)
    _lib.check(_lib.load().dwm_cast_bf16_to_f32(_p(x2), x2.stride(0), _p(o2), o2.stride(0), x2.shape[0], x2.shape[1],
                                                int(accumulate), _stream()), "dwm_cast_bf16_to_f32")
    return out


# ------------------------------------------------------------------------------------------ composites
def linear_dgrad(dy: torch.Tensor, w_t: torch.Tensor, out: Optional[torch.Tensor] = None, **epi) -> torch.Tensor:
    """dX [M, K] = dY [M, N] @ W [N, K], with W^T [K, N] given (the GEMM contracts over the columns of
    both operands).  Extra keyword arguments are GEMM epilogue options (e.g. a fused residual add)."""
    return ops.gemm(dy, w_t, None, out=out, **epi)


def gemm_tn(dy: torch.Tensor, x: torch.Tensor, tap_shifts: Optional[Sequence[int]] = None, split_k: int = 0,
            out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out [N, taps*C] (bf16) = sum over rows m of dy[m, n] * x[clamp(m + shift_t, 0, rows(x) - 1), c] (dwm_gemm_tn): the weight
    gradient of a linear layer (no taps) or - dy and x on the same padded token grid, dy zero on its border rows - of a
    convolution with those taps, from the row-major operands as they are (no transposes, no per-tap gathers).
    dy [M, N], M % 64 == 0; x [rows, C]."""
    _rows2d(dy, "dy")
    _rows2d(x, "x")
    M, N = dy.shape
    Cc = x.shape[1]
    ntaps = len(tap_shifts) if tap_shifts is not None else 0
    if ntaps > 27:
        raise RuntimeError("gemm_tn: at most 27 taps")
    if tap_shifts is None and x.shape[0] < M:
        raise RuntimeError("gemm_tn: x has fewer rows than dy")
    cols = max(ntaps, 1) * Cc
    if out is None:
        out = torch.empty((N, cols), dtype=bf16, device=dy.device)
    _rows2d(out, "out")
    if out.shape != (N, cols):
        raise RuntimeError(f"gemm_tn: out must be [{N}, {cols}]")
    ws = ops._gemm_workspace(dy.device)
    g = _lib.GemmTnArgs()
    g.A, g.lda, g.B, g.ldb, g.b_rows = _p(dy), dy.stride(0), _p(x), x.stride(0), x.shape[0]
    g.out, g.ldo, g.M, g.N, g.C = _p(out), out.stride(0), M, N, Cc
    g.ntaps, g.split_k = ntaps, split_k
    for t in range(ntaps):
        g.tap_shift[t] = int(tap_shifts[t])
    g.workspace, g.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    _lib.check(_lib.load().dwm_gemm_tn(C.byref(g), _stream()), "dwm_gemm_tn")
    return out


def _tn_fits(n: int, cols: int) -> bool:
    """one fp32 partial tile set [n, cols] of dwm_gemm_tn must fit the shared GEMM workspace"""
    return n * cols * 4 <= ops.GEMM_WORKSPACE_BYTES


def linear_wgrad(dy: torch.Tensor, x: torch.Tensor, want_bias: bool = True) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """dW [N, K] (bf16) = dY^T X and db [N] (fp32) = column sums of dY; dY [M, N], X [M, K].
    Both operands are transposed so the contraction (over the M tokens) runs along rows."""
    if WGRAD_TN and dy.shape[0] % 64 == 0 and dy.shape[1] % 8 == 0 and x.shape[1] % 8 == 0 and _tn_fits(dy.shape[1], x.shape[1]):
        dw = gemm_tn(dy, x)                # both operands as they are (gemm_tn.hip)
    else:
        dyt = transpose(dy)                # [N, Mp]
        xt = transpose(x)                  # [K, Mp]
        dw = ops.gemm(dyt, xt, None)       # [N, K]
    db = segsum(dy)[0] if want_bias else None
    return dw, db


def conv_wgrad(dy: torch.Tensor, x_pad: torch.Tensor, idx: torch.Tensor, shifts) -> torch.Tensor:
    """dW [N, taps*C] (tap-major, bf16): dW[n, t, c] = sum_pixels dy[pixel, n] * x_pad[idx[pixel] + shift_t, c].
    dy is scattered onto x_pad's row space (zeros elsewhere) and ALL taps are one dwm_gemm_tn launch over those rows; where
    that does not pay (an output grid much sparser than the input grid: stride-2 convolutions) or does not apply, one
    weight-gradient GEMM per tap on the transposed operands (gather of the tap-shifted rows + transpose, per tap)."""
    N, Cc = dy.shape[1], x_pad.shape[1]
    rows = x_pad.shape[0]
    if WGRAD_TN and N % 8 == 0 and Cc % 8 == 0 and 2 * dy.shape[0] >= rows and _tn_fits(N, len(shifts) * Cc):
        dyp = torch.zeros(((rows + 63) // 64 * 64, N), dtype=bf16, device=dy.device)
        dyp.index_copy_(0, idx, dy)
        return gemm_tn(dyp, x_pad, tap_shifts=[int(s) for s in shifts])
    dw = torch.empty((N, len(shifts) * Cc), dtype=bf16, device=dy.device)
    dyt = transpose(dy)
    for t, sh in enumerate(shifts):
        xt = transpose(x_pad[idx + sh])
        ops.gemm(dyt, xt, None, out=dw[:, t * Cc:(t + 1) * Cc])
    return dw


def groupnorm_bwd(x: torch.Tensor, dz: torch.Tensor, I: int, P: int, gamma: torch.Tensor, beta: torch.Tensor, groups: int,
                  eps: float, dgamma: torch.Tensor, dbeta: torch.Tensor, *, silu: bool = True, dx: Optional[torch.Tensor] = None,
                  accumulate: bool = False, dz_grid=None, img_map: Optional[tuple] = None) -> torch.Tensor:
    """Backward of ops.groupnorm_silu: x [I*P, C] = the forward input, dz = gradient of the forward output - compact
    rows, or (dz_grid) the padded grid the forward wrote into; returns dx [I*P, C] bf16 (accumulate: added to the given dx);
    dgamma / dbeta: fp32 [C] accumulators (see dwm_groupnorm_bwd)."""
    _rows2d(x, "x")
    _rows2d(dz, "dz")
    Cc = x.shape[1]
    rows = dz_grid.rows if dz_grid is not None else I * P
    if not x.is_contiguous() or x.shape[0] != I * P or not dz.is_contiguous() or dz.shape != (rows, Cc):
        raise RuntimeError("groupnorm_bwd: x must be contiguous [I*P, C], dz [rows, C] (rows of the padded grid if dz_grid)")
    for name, t in (("dgamma", dgamma), ("dbeta", dbeta)):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != Cc or not t.is_cuda:
            raise RuntimeError(f"groupnorm_bwd: {name} must be a contiguous fp32 CUDA vector of C elements")
    if dx is None:
        if accumulate:
            raise RuntimeError("groupnorm_bwd: accumulate needs dx")
        dx = torch.empty_like(x)
    _rows2d(dx, "dx")
    if dx.shape != x.shape or not dx.is_contiguous():
        raise RuntimeError("groupnorm_bwd: bad dx")
    lib = _lib.load()
    stats = torch.empty(2 * lib.dwm_groupnorm_stats_floats(I, P, groups), dtype=torch.float32, device=x.device)
    m = _lib.RowMap2D()
    if dz_grid is not None:
        dz_grid.fill(m)
    im = _lib.GnImgMap()
    if img_map is not None:
        im.iv, im.pn, im.s_ihi, im.s_ilo, im.s_phi = img_map
    _lib.check(lib.dwm_groupnorm_bwd(_p(x), _p(dz), _p(dx), I, P, Cc, groups, eps, _p(gamma), _p(beta), int(silu), int(accumulate),
                                     _p(stats), _p(dgamma), _p(dbeta), C.byref(m), C.byref(im), _stream()), "dwm_groupnorm_bwd")
    return dx
