"""Tensor-level wrappers around the C ABI (include/dwm_hip.h).  PyTorch supplies device
memory and the current HIP stream; all arithmetic happens in libdwm_hip.so.  Every
function raises if its tensors are not bf16 CUDA(HIP) tensors — there is no fallback."""
from __future__ import annotations

import contextlib
import ctypes as C
import os
import threading
from dataclasses import dataclass
from typing import Optional, Sequence

import torch

from . import _lib
from ._lib import (ACT_GELU_TANH, ACT_NONE, ACT_RELU, ACT_SILU, EPI_GEGLU, EPI_PLAIN, EPI_RESID,
                   EPI_RMSHEAD)

BIG = 1 << 30
bf16 = torch.bfloat16


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _chk2d(t: torch.Tensor, name: str, dtype=bf16):
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a device tensor (the HIP path has no CPU fallback)")
    if t.dtype != dtype:
        raise RuntimeError(f"{name}: expected {dtype}, got {t.dtype}")
    if t.dim() != 2 or t.stride(1) != 1:
        raise RuntimeError(f"{name}: expected a 2-D tensor with contiguous rows, got {tuple(t.shape)} strides {t.stride()}")


def _chkvec(t: Optional[torch.Tensor], name: str, dtype=bf16):
    if t is None:
        return
    if not t.is_cuda or t.dtype != dtype or not t.is_contiguous():
        raise RuntimeError(f"{name}: expected a contiguous {dtype} device tensor")


# --------------------------------------------------------------------------------- GEMM
@dataclass
class PaddedGrid:
    """Zero-padded token grid [I, h+2, w+2] used by the implicit-GEMM 3x3 convolutions: compact pixel
    index m = (i*h + y)*w + x lives at row i*(h+2)(w+2) + (y+1)(w+2) + (x+1) of the padded matrix."""
    I: int
    h: int
    w: int

    @property
    def rows(self) -> int:
        return self.I * (self.h + 2) * (self.w + 2)

    @property
    def pixels(self) -> int:
        return self.I * self.h * self.w

    def fill(self, m: "_lib.RowMap2D") -> None:
        m.rw, m.rh = self.w, self.h
        m.rpitch, m.ipitch = self.w + 2, (self.h + 2) * (self.w + 2)
        m.origin = self.w + 2 + 1

    def tap_shifts(self):
        """row shift of tap (dy, dx), dy/dx in {-1,0,1}, in the order of a [N, 3, 3, C] weight"""
        return [(dy - 1) * (self.w + 2) + (dx - 1) for dy in range(3) for dx in range(3)]

    def fill_stride2(self, m: "_lib.RowMap2D") -> None:
        """map of the (h/2 x w/2) OUTPUT pixels of a stride-2 3x3 conv with diffusers' (0,1,0,1) padding onto
        this grid: output (y, x) reads input rows (2y + dy, 2x + dx), dy, dx in 0..2, i.e. padded rows
        (2y + 1 + dy, 2x + 1 + dx): origin = first interior row, taps shift by dy*(w+2) + dx."""
        m.rw, m.rh = self.w // 2, self.h // 2
        m.rpitch, m.ipitch = 2 * (self.w + 2), (self.h + 2) * (self.w + 2)
        m.origin = self.w + 2 + 1
        m.xstep = 2

    def tap_shifts_stride2(self):
        return [dy * (self.w + 2) + dx for dy in range(3) for dx in range(3)]

    def fill_stride2_sym(self, m: "_lib.RowMap2D") -> None:
        """stride-2 3x3 conv with symmetric padding 1 (diffusers Downsample2D(padding=1), SD 2.1 UNet): output (y, x)
        reads input rows (2y - 1 + dy, 2x - 1 + dx) = padded rows (2y + dy, 2x + dx): origin 0, same tap shifts."""
        self.fill_stride2(m)
        m.origin = 0

    def interior_index(self) -> torch.Tensor:
        """[pixels] int64 padded-row index of every compact pixel (host reference of the kernel's map)"""
        i = torch.arange(self.I)[:, None, None]
        y = torch.arange(self.h)[None, :, None]
        x = torch.arange(self.w)[None, None, :]
        return (i * (self.h + 2) * (self.w + 2) + (y + 1) * (self.w + 2) + x + 1).reshape(-1)


@dataclass
class TimeGrid:
    """Token rows [(b t v), (h w)] with one zero frame before and after every sequence:
    [B, T + 2, V*N, C] - the input layout of a Conv3d with kernel (3, 1, 1) / padding (1, 0, 0) run as a
    3-tap implicit GEMM (diffusers TemporalResnetBlock).  `vn` = V * h * w rows per frame."""
    B: int
    T: int
    vn: int

    @property
    def rows(self) -> int:
        return self.B * (self.T + 2) * self.vn

    @property
    def pixels(self) -> int:
        return self.B * self.T * self.vn

    def fill(self, m: "_lib.RowMap2D") -> None:
        m.rw, m.rh = self.vn, self.T
        m.rpitch, m.ipitch = self.vn, (self.T + 2) * self.vn
        m.origin = self.vn

    def tap_shifts(self):
        return [-self.vn, 0, self.vn]


@dataclass
class Grid3D:
    """Input layout of a causal 3x3x3 convolution (CogVideoXCausalConv3d) run as a 27-tap implicit GEMM: token rows
    ordered (frame t, video b, y, x) - a frame of all videos is one contiguous slab - with two context frames in
    front of the clip and a zero spatial border: [T + 2, B, h + 2, w + 2, C].  Compact pixel m = ((t*B + b)*h + y)*w + x
    lives at row ((t + 2)*B + b)*(h+2)(w+2) + (y+1)(w+2) + x + 1.  The context frames hold the previous chunk's last
    two input frames (or the first frame twice); there is no trailing padding (causal)."""
    T: int
    B: int
    h: int
    w: int

    @property
    def frame_rows(self) -> int:          # rows of one padded frame slab (all videos)
        return self.B * (self.h + 2) * (self.w + 2)

    @property
    def rows(self) -> int:
        return (self.T + 2) * self.frame_rows

    @property
    def pixels(self) -> int:
        return self.T * self.B * self.h * self.w

    def fill(self, m: "_lib.RowMap2D") -> None:
        m.rw, m.rh = self.w, self.h
        m.rpitch, m.ipitch = self.w + 2, (self.h + 2) * (self.w + 2)
        m.origin = 2 * self.frame_rows + self.w + 2 + 1

    def tap_shifts(self):
        """row shift of tap (dt, dy, dx) in the order of a [N, 3, 3, 3, C] weight; dt = 2 is the current frame"""
        return [(dt - 2) * self.frame_rows + (dy - 1) * (self.w + 2) + (dx - 1)
                for dt in range(3) for dy in range(3) for dx in range(3)]


def _overlaps(a: torch.Tensor, b: torch.Tensor) -> bool:
    """do the byte ranges spanned by two 2-D row-strided tensors intersect?"""
    def span(t):
        lo = t.data_ptr()
        return lo, lo + ((t.shape[0] - 1) * t.stride(0) + t.shape[1]) * t.element_size() if t.numel() else lo
    (a0, a1), (b0, b1) = span(a), span(b)
    return a0 < b1 and b0 < a1


def gemm(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, *,
         out: Optional[torch.Tensor] = None, epilogue: int = EPI_PLAIN, act: int = ACT_NONE,
         gate: Optional[torch.Tensor] = None, rows_per_gate: int = 1,
         res: Optional[torch.Tensor] = None, res_mod: int = 0,
         blend: Optional[torch.Tensor] = None, alpha: Optional[torch.Tensor] = None,
         rows_per_alpha: int = 1,
         rms_w: Optional[torch.Tensor] = None, rms_ncols: int = 0, rms_eps: float = 1e-6,
         a_grid=None, conv3x3: bool = False, stride2=False, conv_taps: Optional[list] = None,
         c_grid=None, out32: Optional[torch.Tensor] = None, mirror: bool = True, w_is_activation: bool = False,
         rows: Optional[int] = None, split_k: int = 0, tile: int = 0, _debug: int = 0) -> torch.Tensor:
    """out[M, Nout] = epilogue(a[M, K] @ w[N, K]^T).  See dwm_gemm_bf16.
    tile: 0 = the kernel's choice, 1 = 256 x 256 tiles, 2 = 256 x 128 tiles (two workgroups per CU).
    a_grid: A is a padded token grid (PaddedGrid.rows x C); M = its pixel count; with conv3x3 the K
    axis is 9 taps x C (w is [N, 9*C], tap-major).  c_grid: out / res / blend are padded grids.
    split_k: 0 = the kernel's rule (small tile grid + long K -> K ranges, fp32 partials, ordered reduction),
    1 = never, n > 1 = exactly n ranges.  _debug: ablation knobs, honoured only by -DDWM_DEV_HOOKS builds of the library.
    out32 (RESID): fp32 residual stream - `res` and `blend` are then fp32 matrices shaped like the output, the result goes to
    out32 in fp32 (out32 may be `res` or `blend` itself) and, rounded, to the bf16 `out`; mirror=False: no bf16 copy at all
    (`out` must be None; the call returns out32) - the hidden / context streams of the bf16 MMDiT forward.
    w_is_activation: `w` is a per-call tensor, not a parameter (fp32 path: its operand planes are not cached)."""
    if a.dtype == torch.float32:            # the fp32 accuracy path (dwm_gemm_f32)
        return _gemm_f32(a, w, bias, out=out, epilogue=epilogue, act=act, gate=gate, rows_per_gate=rows_per_gate, res=res,
                         res_mod=res_mod, blend=blend, alpha=alpha, rows_per_alpha=rows_per_alpha, rms_w=rms_w,
                         rms_ncols=rms_ncols, rms_eps=rms_eps, rows=rows, a_grid=a_grid, conv3x3=conv3x3, c_grid=c_grid,
                         stride2=stride2, conv_taps=conv_taps, cache_w=not w_is_activation)
    _chk2d(a, "a")
    _chk2d(w, "w")
    if not w.is_contiguous():
        raise RuntimeError("w must be contiguous [N, K]")
    N, K = w.shape
    if a_grid is not None:
        ntap_ = 9 if conv3x3 else (len(conv_taps) if conv_taps is not None else 1)
        if a.shape[0] != a_grid.rows or a.shape[1] * ntap_ != K:
            raise RuntimeError(f"gemm: padded A {tuple(a.shape)} does not match grid / weight {tuple(w.shape)}")
        M = a_grid.pixels // 4 if stride2 else a_grid.pixels
    else:
        M = a.shape[0] if rows is None else rows
        if a.shape[1] != K:
            raise RuntimeError(f"gemm: K mismatch {a.shape} x {w.shape}")
    nout = N // 2 if epilogue == EPI_GEGLU else N
    orow = c_grid.rows if c_grid is not None else M
    if c_grid is not None and c_grid.pixels != M:
        raise RuntimeError("gemm: c_grid pixel count != M")
    if not mirror:
        if out32 is None or out is not None:
            raise RuntimeError("gemm: mirror=False needs out32 and no `out`")
    else:
        if out is None:
            out = (torch.zeros if c_grid is not None else torch.empty)((orow, nout), dtype=bf16, device=a.device)
        _chk2d(out, "out")
        if out.shape != (orow, nout):
            raise RuntimeError(f"gemm: out shape {tuple(out.shape)} != {(orow, nout)}")
        if _overlaps(a, out):
            raise RuntimeError("gemm: `out` overlaps the A operand (a tile's output would overwrite rows other tiles still read)")
    _chkvec(bias, "bias")
    g = _lib.GemmArgs()
    g.A, g.lda, g.W, g.bias = a.data_ptr(), a.stride(0), w.data_ptr(), _p(bias)
    if out is not None:
        g.C, g.ldc = out.data_ptr(), out.stride(0)
    g.M, g.N, g.K, g.epilogue, g.act = M, N, K, epilogue, act
    if gate is not None:
        _chk2d(gate, "gate")
        g.gate, g.ld_gate, g.rows_per_gate = gate.data_ptr(), gate.stride(0), rows_per_gate
    if out32 is not None:
        if epilogue != EPI_RESID or res_mod != 0:
            raise RuntimeError("gemm: out32 needs the RESID epilogue (res_mod = 0; `res`, if any, in fp32)")
        _chk2d(out32, "out32", torch.float32)
        if out32.shape != (orow, nout):
            raise RuntimeError("gemm: out32 must have the shape of the output")
        if res is not None:
            _chk2d(res, "res", torch.float32)
            if res.shape != out32.shape:
                raise RuntimeError("gemm: the fp32 res must have the shape of the output")
            g.res, g.ld_res, g.res_mod = res.data_ptr(), res.stride(0), 0
        g.C32, g.ldc32 = out32.data_ptr(), out32.stride(0)
        split_k = 1
    elif res is not None:
        _chk2d(res, "res")
        g.res, g.ld_res, g.res_mod = res.data_ptr(), res.stride(0), res_mod
    if blend is not None:
        _chk2d(blend, "blend", torch.float32 if out32 is not None else bf16)
        _chkvec(alpha, "alpha", torch.float32)
        g.blend, g.ld_blend, g.alpha, g.rows_per_alpha = blend.data_ptr(), blend.stride(0), alpha.data_ptr(), rows_per_alpha
    if rms_w is not None:
        _chkvec(rms_w, "rms_w")
        g.rms_w, g.rms_ncols, g.rms_eps = rms_w.data_ptr(), rms_ncols, rms_eps
    if a_grid is not None:
        if stride2:
            if not conv3x3 or a_grid.h % 2 or a_grid.w % 2:
                raise RuntimeError("gemm: stride2 needs conv3x3 on an even-sized grid")
            (a_grid.fill_stride2_sym if stride2 == "sym" else a_grid.fill_stride2)(g.a_map)
        else:
            a_grid.fill(g.a_map)
        if conv3x3:
            g.ntaps, g.k_per_tap = 9, a.shape[1]
            for t, sh in enumerate(a_grid.tap_shifts_stride2() if stride2 else a_grid.tap_shifts()):
                g.tap_shift[t] = sh
        elif conv_taps is not None:
            g.ntaps, g.k_per_tap = len(conv_taps), a.shape[1]
            for t, sh in enumerate(conv_taps):
                g.tap_shift[t] = sh
    if c_grid is not None:
        c_grid.fill(g.c_map)
        for name, t in (("res", res), ("blend", blend)):
            if t is not None and t.shape[0] != c_grid.rows:
                raise RuntimeError(f"gemm: {name} must be a padded grid when c_grid is given")
    g.reserved = _debug
    g.split_k = split_k
    g.tile = _gemm_tile_request(tile)
    # split-K scratch (fp32 partial tiles): only handed over when the kernel's own rule can take it
    if split_k != 1 and epilogue in (EPI_PLAIN, EPI_RESID) and ((M + 255) // 256) * ((N + 255) // 256) <= 128 and K >= 1024:
        ws = _gemm_workspace(a.device)
        g.workspace, g.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    _lib.check(_lib.load().dwm_gemm_bf16(C.byref(g), _stream()), "dwm_gemm_bf16")
    return out if mirror else out32


# 4-wave GEMM kernels (gemm_bf16_4w.hip) for the calling thread's launches: the MMDiT inference forward and the MMDiT train step open a
# `gemm_4wave_scope` around their own launches (`model.gemm_4wave`); inside it `ops.gemm(..., tile=0)` asks for dwm_gemm_args.tile = 3
# ("automatic, and the 4-wave kernels may serve the launch").  Everything else - UNet, VAEs, direct ops.gemm calls - keeps the 8-wave
# kernels.  The switch is per thread (no module global to race on between models / streams driven from different threads); autograd
# runs backward on its own thread, so the training Functions re-open the scope there (`scope_of` / `capture_scope`).
# Environment DWM_GEMM4W (read HERE, on the host side - the library takes its kernel selection from dwm_gemm_args only): 0 = never,
# 1 = every automatic-tile launch, f = as the scope asks but the fast form only (dwm_gemm_args.tile = 4).
_G4W = threading.local()
_G4W_ENV = os.environ.get("DWM_GEMM4W", "")


def block_permute(src: torch.Tensor, dst: torch.Tensor, dims, src_strides, block_elems: int) -> torch.Tensor:
    """dst (dense, blocks in the order dims = (n0, n1, n2, n3)) <- src block at sum(i_k * src_strides[k]) (strides in blocks); blocks of
    `block_elems` contiguous elements.  dwm_block_permute: the pack / unpack of the frame-shard all-to-all (sharding.py)."""
    if not (src.is_cuda and dst.is_cuda and src.is_contiguous() and dst.is_contiguous() and src.dtype == dst.dtype):
        raise RuntimeError("block_permute: contiguous device tensors of one dtype expected")
    a = _lib.BlockPermuteArgs()
    a.src, a.dst = src.data_ptr(), dst.data_ptr()
    a.block_bytes = block_elems * src.element_size()
    n = 1
    for i in range(4):
        a.n[i], a.sstride[i] = dims[i], src_strides[i]
        n *= dims[i]
    if n * block_elems != src.numel() or dst.numel() != src.numel():
        raise RuntimeError("block_permute: dims x block size do not cover the tensors")
    _lib.check(_lib.load().dwm_block_permute(C.byref(a), _stream()), "dwm_block_permute")
    return dst


ATTN_Q_PRESCALED = 1 << 15      # dwm_attn_args.variant: q arrives with scale * log2(e) folded in by its producer
ATTN_STREAM = 1 << 12           # ... the one-wave-per-SIMD streaming form of the resident kernel (attention_stream.hip): the library's default where it covers the launch
ATTN_RES12 = 1 << 13            # ... keep the 12-wave resident kernel there (A/B measurements)

# Environment DWM_ATTN_VARIANT (an integer, e.g. 0x1000): bits OR-ed into dwm_attn_args.variant of every ops.attention call (A/B
# measurements of the attention kernels; the library reads no environment).  DWM_ATTN_RES4=1 / 2 = bits 12 / 12 + 13 (round 5's name).
_ATTN_ENV_VARIANT = int(os.environ.get("DWM_ATTN_VARIANT", "0") or "0", 0) | \
    {"1": 1 << 12, "2": (1 << 12) | (1 << 13)}.get(os.environ.get("DWM_ATTN_RES4", "")[:1], 0)


def _gemm_tile_request(tile: int) -> int:
    """dwm_gemm_args.tile of one launch: the caller's explicit configuration (1 / 2), else 3 / 4 ("automatic, and the 4-wave kernels may
    serve the launch" / "... their fast form only") inside a gemm_4wave_scope or under DWM_GEMM4W=1"""
    if tile != 0:
        return tile
    env = _G4W_ENV[:1]
    on = getattr(_G4W, "on", False)
    if env == "0":
        return 0
    if env == "f":
        return 4 if on else 0
    if env not in ("", "0"):
        return 3
    return 3 if on else 0


def gemm_4wave_enabled() -> bool:
    return bool(getattr(_G4W, "on", False))


@contextlib.contextmanager
def gemm_4wave_scope(on: bool):
    prev = gemm_4wave_enabled()
    _G4W.on = bool(on)
    try:
        yield
    finally:
        _G4W.on = prev


def carries_gemm_scope(cls):
    """class decorator for torch.autograd.Function subclasses: the backward (autograd's own thread) runs inside the
    gemm_4wave_scope its forward ran in"""
    fwd, bwd = cls.forward, cls.backward

    def forward(ctx, *args, **kwargs):
        ctx._gemm_4wave = gemm_4wave_enabled()
        return fwd(ctx, *args, **kwargs)

    def backward(ctx, *grads):
        with gemm_4wave_scope(getattr(ctx, "_gemm_4wave", False)):
            return bwd(ctx, *grads)

    cls.forward, cls.backward = staticmethod(forward), staticmethod(backward)
    return cls


_SPLIT_WEIGHTS: dict = {}
SPLIT_WEIGHT_CACHE_MAX = 2048        # > the ~1300 weight matrices of the largest model on the path


def split_weight(w: torch.Tensor, taps: int = 1, cache: bool = True) -> torch.Tensor:
    """fp32 [N, K] -> the pre-split bf16 operand [N, 3K] of dwm_gemm_f32 (hi = bf16(w), lo = bf16(w - hi)): [hi | lo | hi], or -
    `taps` > 1, K = taps x C tap-major (implicit convolution) - [hi_t | lo_t | hi_t] per tap t; more than 9 taps (the 27 of a causal
    3x3x3 convolution) are laid out as groups of 9, group g = [N, 9 * 3C] contiguous, one after the other (dwm_gemm_f32 walks a
    group per main-loop launch).
    Cached on the tensor's storage / version (weights and packed weights are long-lived), at most SPLIT_WEIGHT_CACHE_MAX entries
    (least recently used first out: per-call temporaries must not pile up); `clear_split_weights()` drops everything."""
    key = (w.data_ptr(), w._version, tuple(w.shape), taps)
    hit = _SPLIT_WEIGHTS.pop(key, None) if cache else None
    if hit is None:
        hi = w.to(bf16)
        lo = (w - hi.float()).to(bf16)
        if taps > 1:
            N, K = w.shape
            h3, l3 = hi.view(N, taps, K // taps), lo.view(N, taps, K // taps)
            ws = torch.stack([h3, l3, h3], 2)                                  # [N, taps, 3, C]
            if taps > 9:
                ws = torch.cat([ws[:, t0:t0 + 9].reshape(-1) for t0 in range(0, taps, 9)]).view(N, 3 * K)
            else:
                ws = ws.reshape(N, 3 * K).contiguous()
        else:
            ws = torch.cat([hi, lo, hi], 1).contiguous()
        hit = (ws, w)                                               # keeps `w` alive: its address is the key
    if not cache:
        return hit[0]
    _SPLIT_WEIGHTS[key] = hit                                       # (re-)inserted last: dict order = recency
    while len(_SPLIT_WEIGHTS) > SPLIT_WEIGHT_CACHE_MAX:
        _SPLIT_WEIGHTS.pop(next(iter(_SPLIT_WEIGHTS)))
    return hit[0]


def clear_split_weights() -> None:
    _SPLIT_WEIGHTS.clear()


def _gemm_f32(a, w, bias, *, out, epilogue, act, gate, rows_per_gate, res, res_mod, blend, alpha, rows_per_alpha, rms_w, rms_ncols,
              rms_eps, rows, a_grid=None, conv3x3=False, c_grid=None, stride2=False, conv_taps=None, cache_w=True):
    f32 = torch.float32
    _chk2d(a, "a", f32)
    _chk2d(w, "w", f32)
    if not w.is_contiguous():
        raise RuntimeError("w must be contiguous [N, K]")
    N, K = w.shape
    taps = 9 if conv3x3 else len(conv_taps) if conv_taps is not None else 1
    if taps > 27:
        raise NotImplementedError("gemm (fp32): at most 27 taps (three groups of 9, each tap walked as three plane products)")
    if a_grid is not None:
        if a.shape[0] != a_grid.rows or a.shape[1] * taps != K or not a.is_contiguous():
            raise RuntimeError(f"gemm: padded A {tuple(a.shape)} does not match grid / weight {tuple(w.shape)}")
        if stride2 and (not conv3x3 or a_grid.h % 2 or a_grid.w % 2):
            raise RuntimeError("gemm: stride2 needs conv3x3 on an even-sized grid")
        M, a_rows = (a_grid.pixels // 4 if stride2 else a_grid.pixels), a_grid.rows
    else:
        if conv3x3 or conv_taps is not None or stride2:
            raise RuntimeError("gemm: convolution taps need a_grid")
        M = a.shape[0] if rows is None else rows
        a_rows = M
        if a.shape[1] != K:
            raise RuntimeError(f"gemm: K mismatch {a.shape} x {w.shape}")
    nout = N // 2 if epilogue == EPI_GEGLU else N
    orow = c_grid.rows if c_grid is not None else M
    if c_grid is not None and c_grid.pixels != M:
        raise RuntimeError("gemm: c_grid pixel count != M")
    if out is None:
        out = (torch.zeros if c_grid is not None else torch.empty)((orow, nout), dtype=f32, device=a.device)
    _chk2d(out, "out", f32)
    if out.shape != (orow, nout):
        raise RuntimeError(f"gemm: out shape {tuple(out.shape)} != {(orow, nout)}")
    _chkvec(bias, "bias", f32)
    ws = split_weight(w, taps, cache=cache_w)      # (an activation in the W position - attention scores of the VAE - is not kept)
    g = _lib.GemmArgs()
    g.A, g.lda, g.W, g.bias, g.C, g.ldc = a.data_ptr(), a.stride(0), ws.data_ptr(), _p(bias), out.data_ptr(), out.stride(0)
    g.M, g.N, g.K, g.epilogue, g.act = M, N, K, epilogue, act
    if a_grid is not None:
        if stride2:
            (a_grid.fill_stride2_sym if stride2 == "sym" else a_grid.fill_stride2)(g.a_map)
        else:
            a_grid.fill(g.a_map)
        if conv3x3:
            g.ntaps, g.k_per_tap = 9, a.shape[1]
            for t, sh in enumerate(a_grid.tap_shifts_stride2() if stride2 else a_grid.tap_shifts()):
                g.tap_shift[t] = sh
        elif conv_taps is not None:
            g.ntaps, g.k_per_tap = len(conv_taps), a.shape[1]
            for t, sh in enumerate(conv_taps):
                g.tap_shift[t] = sh
    if c_grid is not None:
        c_grid.fill(g.c_map)
        for name, t in (("res", res), ("blend", blend)):
            if t is not None and t.shape[0] != c_grid.rows:
                raise RuntimeError(f"gemm: {name} must be a padded grid when c_grid is given")
    if gate is not None:
        _chk2d(gate, "gate", f32)
        g.gate, g.ld_gate, g.rows_per_gate = gate.data_ptr(), gate.stride(0), rows_per_gate
    if res is not None:
        _chk2d(res, "res", f32)
        g.res, g.ld_res, g.res_mod = res.data_ptr(), res.stride(0), res_mod
    if blend is not None:
        _chk2d(blend, "blend", f32)
        _chkvec(alpha, "alpha", f32)
        g.blend, g.ld_blend, g.alpha, g.rows_per_alpha = blend.data_ptr(), blend.stride(0), alpha.data_ptr(), rows_per_alpha
    if rms_w is not None:
        _chkvec(rms_w, "rms_w", f32)
        g.rms_w, g.rms_ncols, g.rms_eps = rms_w.data_ptr(), rms_ncols, rms_eps
    groups = (taps + 8) // 9                                  # one set of fp32 partial slices per group of 9 taps
    need = 4 * a_rows * (K // taps) + 256 + 4 * M * N * groups
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    if tiles * groups <= 128:
        need += 4 * M * N * min(32, 256 // tiles)          # room for the kernel's own split-K rule
    wsb = _f32_workspace(a.device, need)
    g.workspace, g.workspace_bytes = wsb.data_ptr(), wsb.numel() * 4
    _lib.check(_lib.load().dwm_gemm_f32(C.byref(g), _stream()), "dwm_gemm_f32")
    return out


_F32_WORKSPACES: dict = {}


def _f32_workspace(device: torch.device, nbytes: int) -> torch.Tensor:
    """scratch of dwm_gemm_f32 (operand planes + fp32 partial sums), one per (device, stream), grown on demand"""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _F32_WORKSPACES.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        ws = _F32_WORKSPACES[key] = torch.empty((nbytes + (64 << 20)) // 4, dtype=torch.float32, device=device)
    return ws


_WORKSPACES: dict = {}
GEMM_WORKSPACE_BYTES = 1 << 30          # 1 GiB per (device, stream): up to ~8 K ranges of the largest weight gradient (12288 x 1536 fp32)


def _gemm_workspace(device: torch.device) -> torch.Tensor:
    """fp32 scratch of the split-K GEMM path, one per (device, stream): partial tiles live there only between the
    two kernels of one dwm_gemm_bf16 call, so calls on one stream share it."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _WORKSPACES.get(key)
    if ws is None:
        ws = _WORKSPACES[key] = torch.empty(GEMM_WORKSPACE_BYTES // 4, dtype=torch.float32, device=device)
    return ws


# ---------------------------------------------------------------------------- attention
@dataclass
class RowMap:
    """row(p, l) of include/dwm_hip.h; see the constructors below for the reference
    rearranges (crossview_temporal_dit.py:300-361) they encode."""
    L0: int
    n_problems: int
    pdiv: Sequence[int] = (1, 1, 1)
    pmod: Sequence[int] = (BIG, 1, 1)
    pstride: Sequence[int] = (0, 0, 0)
    ldiv: Sequence[int] = (BIG, BIG)
    lstride: Sequence[int] = (1, 0, 0)
    # group mask geometry (cross-view): mask index = p // p_per_mask, group(l) = (l // group_size) % G
    p_per_mask: int = 1
    group_size: int = 1

    def rows(self) -> torch.Tensor:
        """[n_problems, L0] int64 row indices (host reference of the kernel's addressing)."""
        p = torch.arange(self.n_problems)[:, None]
        l = torch.arange(self.L0)[None, :]
        r = torch.zeros(self.n_problems, self.L0, dtype=torch.int64)
        for d, m, s in zip(self.pdiv, self.pmod, self.pstride):
            r = r + ((p // d) % m) * s
        r = r + (l % self.ldiv[0]) * self.lstride[0] + ((l // self.ldiv[0]) % self.ldiv[1]) * self.lstride[1] \
            + (l // (self.ldiv[0] * self.ldiv[1])) * self.lstride[2]
        return r


def rowmap_identity(n_problems: int, L0: int) -> RowMap:
    """(b) n c: problem p owns rows p*L0 .. p*L0+L0-1 (joint / dual attention)."""
    return RowMap(L0=L0, n_problems=n_problems, pstride=(L0, 0, 0))


def rowmap_crossview_rowwise(B: int, T: int, V: int, h: int, w: int) -> RowMap:
    """(bt v) (h w) c -> (bt h) (v w) c   (crossview_temporal_dit.py:307-309)."""
    return RowMap(L0=V * w, n_problems=B * T * h,
                  pdiv=(h, 1, 1), pmod=(BIG, h, 1), pstride=(V * h * w, w, 0),
                  ldiv=(w, BIG), lstride=(1, h * w, 0), p_per_mask=T * h, group_size=w)


def rowmap_crossview_full(B: int, T: int, V: int, h: int, w: int) -> RowMap:
    """(bt v) (h w) c -> bt (h v w) c   (crossview_temporal_dit.py:290-292)."""
    return RowMap(L0=h * V * w, n_problems=B * T,
                  pstride=(V * h * w, 0, 0),
                  ldiv=(w, V), lstride=(1, h * w, w), p_per_mask=T, group_size=w)


def rowmap_crossview_pointwise(B: int, T: int, V: int, h: int, w: int) -> RowMap:
    """btv hw c -> (bt hw) v c: TemporalBasicTransformerBlock(num_frames=V) of the SD 2.1 UNet without row-wise
    cross-view attention (crossview_temporal.py:358-361, :228-229)."""
    hw = h * w
    return RowMap(L0=V, n_problems=B * T * hw,
                  pdiv=(hw, 1, 1), pmod=(BIG, hw, 1), pstride=(V * hw, 1, 0),
                  ldiv=(BIG, BIG), lstride=(hw, 0, 0), p_per_mask=T * hw, group_size=1)


def rowmap_temporal_rowwise(B: int, T: int, V: int, h: int, w: int) -> RowMap:
    """(b t v) (h w) c -> (b v h) (t w) c   (crossview_temporal_dit.py:345-347)."""
    return RowMap(L0=T * w, n_problems=B * V * h,
                  pdiv=(V * h, h, 1), pmod=(BIG, V, h), pstride=(T * V * h * w, h * w, w),
                  ldiv=(w, BIG), lstride=(1, V * h * w, 0))


def rowmap_temporal_full(B: int, T: int, V: int, h: int, w: int) -> RowMap:
    """(b t v) hw c -> (b v) (t hw) c   (crossview_temporal_dit.py:336-338)."""
    return RowMap(L0=T * h * w, n_problems=B * V,
                  pdiv=(V, 1, 1), pmod=(BIG, V, 1), pstride=(T * V * h * w, h * w, 0),
                  ldiv=(h * w, BIG), lstride=(1, V * h * w, 0))


def rowmap_temporal_pointwise(B: int, T: int, V: int, h: int, w: int) -> RowMap:
    """(b t v) hw c -> (b v hw) t c   (crossview_temporal_dit.py:354-356)."""
    hw = h * w
    return RowMap(L0=T, n_problems=B * V * hw,
                  pdiv=(V * hw, hw, 1), pmod=(BIG, V, hw), pstride=(T * V * hw, hw, 1),
                  ldiv=(BIG, BIG), lstride=(V * hw, 0, 0))


def _attn_args(a, q, k, v, out, rowmap, heads, q1, k1, v1, out1, scale, group_mask, dense_mask):
    """fill a dwm_attn_args; returns the uint8 mask tensor that must outlive the launch (or None)"""
    dt = q.dtype if q.dtype == torch.float32 else bf16             # fp32: the accuracy path (dwm_attention_f32)
    for name, t in (("q", q), ("k", k), ("v", v), ("out", out)):
        _chk2d(t, name, dt)
    if not (q.stride(0) == k.stride(0) == v.stride(0)):
        raise RuntimeError("q, k, v must share a row stride")
    a.q0, a.k0, a.v0, a.ld0 = q.data_ptr(), k.data_ptr(), v.data_ptr(), q.stride(0)
    a.o0, a.ldo0 = out.data_ptr(), out.stride(0)
    a.L0, a.L1, a.n_problems = rowmap.L0, 0, rowmap.n_problems
    if q1 is not None:
        for name, t in (("q1", q1), ("k1", k1), ("v1", v1), ("out1", out1)):
            _chk2d(t, name, dt)
        if q1.shape[0] % rowmap.n_problems != 0:
            raise RuntimeError("segment 1 rows must be n_problems * L1")
        a.q1, a.k1, a.v1, a.ld1 = q1.data_ptr(), k1.data_ptr(), v1.data_ptr(), q1.stride(0)
        a.o1, a.ldo1 = out1.data_ptr(), out1.stride(0)
        a.L1 = q1.shape[0] // rowmap.n_problems
    a.heads, a.head_dim = heads, 64
    if q.shape[1] != heads * 64:
        raise RuntimeError("attention: head_dim must be 64")
    a.scale = float(scale) if scale is not None else 64 ** -0.5
    for i in range(3):
        a.pdiv[i], a.pmod[i], a.pstride[i] = rowmap.pdiv[i], rowmap.pmod[i], rowmap.pstride[i]
        a.lstride[i] = rowmap.lstride[i]
    a.ldiv[0], a.ldiv[1] = rowmap.ldiv
    a.mask_mode = 0
    keep = None
    if group_mask is not None:
        keep = group_mask.to(torch.uint8).contiguous()
        a.mask_mode, a.mask = 1, keep.data_ptr()
        a.mask_G, a.group_size, a.p_per_mask = keep.shape[-1], rowmap.group_size, rowmap.p_per_mask
    elif dense_mask is not None:
        keep = dense_mask.to(torch.uint8).contiguous()
        a.mask_mode, a.mask = 2, keep.data_ptr()
    return keep


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, out: torch.Tensor,
              rowmap: RowMap, heads: int, *,
              q1: Optional[torch.Tensor] = None, k1: Optional[torch.Tensor] = None,
              v1: Optional[torch.Tensor] = None, out1: Optional[torch.Tensor] = None,
              scale: Optional[float] = None,
              group_mask: Optional[torch.Tensor] = None, dense_mask: Optional[torch.Tensor] = None,
              variant: int = 0, lse: Optional[torch.Tensor] = None) -> None:
    """softmax(QK^T * scale [+mask]) V per (problem, head); q/k/v/out are 2-D [rows, heads*64]
    views sharing a row stride (e.g. column slices of a fused qkv buffer).  Segment 1
    (q1/k1/v1/out1: [n_problems*L1, heads*64]) is appended to every problem's key/query
    sequence (text context of the joint attention).  group_mask: bool/uint8 [Bm, G, G];
    dense_mask: bool/uint8 [n_problems, L, L].  lse (optional, fp32 [n_problems, heads, L0+L1])
    receives the negative log2-domain log-sum-exp the backward needs."""
    a = _lib.AttnArgs()
    keep = _attn_args(a, q, k, v, out, rowmap, heads, q1, k1, v1, out1, scale, group_mask, dense_mask)
    a.variant = variant | _ATTN_ENV_VARIANT
    if lse is not None:
        if lse.dtype != torch.float32 or not lse.is_contiguous() or lse.numel() != a.n_problems * heads * (a.L0 + a.L1):
            raise RuntimeError("lse: fp32 contiguous [n_problems, heads, L0+L1] expected")
        a.lse = lse.data_ptr()
    if q.dtype == torch.float32:
        _lib.check(_lib.load().dwm_attention_f32(C.byref(a), _stream()), "dwm_attention_f32")
    else:
        _lib.check(_lib.load().dwm_attention_fwd(C.byref(a), _stream()), "dwm_attention_fwd")
    if keep is not None:
        keep.record_stream(torch.cuda.current_stream())


def attention_bwd(q, k, v, out, dout, dq, dk, dv, rowmap: RowMap, heads: int, lse: torch.Tensor, *,
                  q1=None, k1=None, v1=None, out1=None, dout1=None, dq1=None, dk1=None, dv1=None,
                  scale: Optional[float] = None, group_mask=None, dense_mask=None) -> None:
    """Backward of `attention` (same arguments plus the forward outputs, their gradients and lse);
    writes dq/dk/dv (and dq1/dk1/dv1), which share a row stride per segment."""
    b = _lib.AttnBwdArgs()
    keep = _attn_args(b.fwd, q, k, v, out, rowmap, heads, q1, k1, v1, out1, scale, group_mask, dense_mask)
    b.fwd.lse = lse.data_ptr()
    for name, t in (("dout", dout), ("dq", dq), ("dk", dk), ("dv", dv)):
        _chk2d(t, name)
    if dout.stride(0) != out.stride(0) or not (dq.stride(0) == dk.stride(0) == dv.stride(0)):
        raise RuntimeError("dout must be laid out like out; dq, dk, dv must share a row stride")
    b.do0, b.dq0, b.dk0, b.dv0, b.ld_d0 = dout.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), dq.stride(0)
    if q1 is not None:
        for name, t in (("dout1", dout1), ("dq1", dq1), ("dk1", dk1), ("dv1", dv1)):
            _chk2d(t, name)
        if dout1.stride(0) != out1.stride(0):
            raise RuntimeError("dout1 must be laid out like out1")
        b.do1, b.dq1, b.dk1, b.dv1, b.ld_d1 = dout1.data_ptr(), dq1.data_ptr(), dk1.data_ptr(), dv1.data_ptr(), dq1.stride(0)
    delta = torch.empty(lse.numel(), dtype=torch.float32, device=lse.device)
    b.delta = delta.data_ptr()
    _lib.check(_lib.load().dwm_attention_bwd(C.byref(b), _stream()), "dwm_attention_bwd")
    delta.record_stream(torch.cuda.current_stream())
    if keep is not None:
        keep.record_stream(torch.cuda.current_stream())


def _cross_args(a, q, k, v, out, n_problems, heads, scale):
    for name, t in (("q", q), ("k", k), ("v", v), ("out", out)):
        _chk2d(t, name, q.dtype if q.dtype == torch.float32 else bf16)
    if k.stride(0) != v.stride(0) or q.shape[0] % n_problems or k.shape[0] % n_problems:
        raise RuntimeError("cross_attention: k, v must share a row stride; rows must be n_problems * L")
    a.q0 = a.q1 = q.data_ptr()
    a.k0 = a.k1 = k.data_ptr()
    a.v0 = a.v1 = v.data_ptr()
    a.ld0, a.ld1 = q.stride(0), k.stride(0)
    a.o0, a.ldo0 = out.data_ptr(), out.stride(0)
    a.L0, a.L1, a.n_problems = q.shape[0] // n_problems, k.shape[0] // n_problems, n_problems
    a.heads, a.head_dim = heads, 64
    if q.shape[1] != heads * 64:
        raise RuntimeError("attention: head_dim must be 64")
    a.scale = float(scale) if scale is not None else 64 ** -0.5
    rm = rowmap_identity(n_problems, a.L0)
    for i in range(3):
        a.pdiv[i], a.pmod[i], a.pstride[i] = rm.pdiv[i], rm.pmod[i], rm.pstride[i]
        a.lstride[i] = rm.lstride[i]
    a.ldiv[0], a.ldiv[1] = rm.ldiv
    a.cross = 1


def cross_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, out: torch.Tensor, n_problems: int, heads: int,
                    scale: Optional[float] = None, lse: Optional[torch.Tensor] = None) -> None:
    """Cross-attention (diffusers BasicTransformerBlock.attn2): q/out [n_problems*Lq, heads*64], k/v
    [n_problems*Lk, heads*64] (column slices of one buffer sharing a row stride); every query attends to all Lk keys of
    its problem.  Runs dwm_attention_fwd in `cross` mode: queries = segment 0, keys / values = segment 1.
    lse (optional, fp32 [n_problems, heads, Lq + Lk], the query entries are written): for cross_attention_bwd."""
    a = _lib.AttnArgs()
    _cross_args(a, q, k, v, out, n_problems, heads, scale)
    if lse is not None:
        if lse.dtype != torch.float32 or not lse.is_contiguous() or lse.numel() != n_problems * heads * (a.L0 + a.L1):
            raise RuntimeError("lse: fp32 contiguous [n_problems, heads, Lq+Lk] expected")
        a.lse = lse.data_ptr()
    if q.dtype == torch.float32:                 # the fp32 accuracy path (UNet text cross-attention)
        if lse is not None:
            raise NotImplementedError("cross_attention: no LSE output on the fp32 path (inference only)")
        _lib.check(_lib.load().dwm_attention_f32(C.byref(a), _stream()), "dwm_attention_f32")
        return
    _lib.check(_lib.load().dwm_attention_fwd(C.byref(a), _stream()), "dwm_attention_fwd")


def cross_attention_bwd(q, k, v, out, dout, dq, dk, dv, n_problems: int, heads: int, lse: torch.Tensor,
                        scale: Optional[float] = None) -> None:
    """Backward of `cross_attention`: dq like q (row stride of its own), dk / dv like k / v (sharing a row stride)."""
    b = _lib.AttnBwdArgs()
    _cross_args(b.fwd, q, k, v, out, n_problems, heads, scale)
    b.fwd.lse = lse.data_ptr()
    for name, t in (("dout", dout), ("dq", dq), ("dk", dk), ("dv", dv)):
        _chk2d(t, name)
    if dout.stride(0) != out.stride(0) or dk.stride(0) != dv.stride(0) or dq.shape != q.shape or dk.shape != k.shape:
        raise RuntimeError("cross_attention_bwd: dout like out, dq like q, dk / dv like k / v (one row stride)")
    # one gradient base per tensor for both segments (the kernels address segment 1 relative to segment 0, cf. the forward)
    b.do0 = b.do1 = dout.data_ptr()
    b.dq0 = b.dq1 = dq.data_ptr()
    b.dk0 = b.dk1 = dk.data_ptr()
    b.dv0 = b.dv1 = dv.data_ptr()
    b.ld_d0, b.ld_d1 = dq.stride(0), dk.stride(0)
    delta = torch.empty(lse.numel(), dtype=torch.float32, device=lse.device)
    b.delta = delta.data_ptr()
    _lib.check(_lib.load().dwm_attention_bwd(C.byref(b), _stream()), "dwm_attention_bwd")
    delta.record_stream(torch.cuda.current_stream())


# -------------------------------------------------------------------------------- norms
def layernorm(x: torch.Tensor, *, eps: float, out: Optional[torch.Tensor] = None,
              weight: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None,
              scale: Optional[torch.Tensor] = None, shift: Optional[torch.Tensor] = None,
              rows_per_mod: int = 0,
              scale2: Optional[torch.Tensor] = None, shift2: Optional[torch.Tensor] = None,
              out2: Optional[torch.Tensor] = None,
              addvec: Optional[torch.Tensor] = None, rows_per_add: int = 0,
              xsum: Optional[torch.Tensor] = None, x32: bool = False):
    """See dwm_layernorm.  scale/shift (and scale2/shift2) are 2-D views sharing one row
    stride (column slices of the AdaLN modulation matrix).  x32: x (and xsum) are the fp32 residual stream of the bf16
    forward, everything else bf16 (dwm_layernorm_x32)."""
    if x32:
        return _layernorm_x32(x, eps=eps, out=out, weight=weight, bias=bias, scale=scale, shift=shift, rows_per_mod=rows_per_mod,
                              scale2=scale2, shift2=shift2, out2=out2, addvec=addvec, rows_per_add=rows_per_add, xsum=xsum)
    dt = x.dtype if x.dtype == torch.float32 else bf16             # fp32: the accuracy path (dwm_layernorm_f32)
    _chk2d(x, "x", dt)
    rows, D = x.shape
    if out is None:
        out = torch.empty((rows, D), dtype=dt, device=x.device)
    _chk2d(out, "out", dt)
    a = _lib.LayerNormArgs()
    a.x, a.ldx, a.y, a.ldy = x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0)
    a.rows, a.D, a.eps = rows, D, eps
    _chkvec(weight, "weight", dt)
    _chkvec(bias, "bias", dt)
    a.weight, a.bias = _p(weight), _p(bias)
    if scale is not None:
        _chk2d(scale, "scale", dt)
        _chk2d(shift, "shift", dt)
        if scale.stride(0) != shift.stride(0):
            raise RuntimeError("scale/shift must share a row stride")
        a.scale, a.shift, a.ld_mod, a.rows_per_mod = scale.data_ptr(), shift.data_ptr(), scale.stride(0), rows_per_mod
    if out2 is not None:
        _chk2d(out2, "out2", dt)
        _chk2d(scale2, "scale2", dt)
        _chk2d(shift2, "shift2", dt)
        if scale2.stride(0) != a.ld_mod or shift2.stride(0) != a.ld_mod:
            raise RuntimeError("scale2/shift2 must share the row stride of scale/shift")
        a.y2, a.ldy2, a.scale2, a.shift2 = out2.data_ptr(), out2.stride(0), scale2.data_ptr(), shift2.data_ptr()
    if addvec is not None:
        _chk2d(addvec, "addvec", dt)
        a.addvec, a.ld_add, a.rows_per_add = addvec.data_ptr(), addvec.stride(0), rows_per_add
        if xsum is not None:
            _chk2d(xsum, "xsum", dt)
            a.xsum, a.ldxsum = xsum.data_ptr(), xsum.stride(0)
    if dt == torch.float32:
        _lib.check(_lib.load().dwm_layernorm_f32(C.byref(a), _stream()), "dwm_layernorm_f32")
        return out
    _lib.check(_lib.load().dwm_layernorm(C.byref(a), _stream()), "dwm_layernorm")
    return out


def _layernorm_x32(x, *, eps, out, weight, bias, scale, shift, rows_per_mod, scale2, shift2, out2, addvec, rows_per_add, xsum):
    _chk2d(x, "x", torch.float32)
    rows, D = x.shape
    if out is None:
        out = torch.empty((rows, D), dtype=bf16, device=x.device)
    _chk2d(out, "out")
    a = _lib.LayerNormArgs()
    a.x, a.ldx, a.y, a.ldy = x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0)
    a.rows, a.D, a.eps = rows, D, eps
    _chkvec(weight, "weight")
    _chkvec(bias, "bias")
    a.weight, a.bias = _p(weight), _p(bias)
    if scale is not None:
        _chk2d(scale, "scale")
        _chk2d(shift, "shift")
        if scale.stride(0) != shift.stride(0):
            raise RuntimeError("scale/shift must share a row stride")
        a.scale, a.shift, a.ld_mod, a.rows_per_mod = scale.data_ptr(), shift.data_ptr(), scale.stride(0), rows_per_mod
    if out2 is not None:
        _chk2d(out2, "out2")
        _chk2d(scale2, "scale2")
        _chk2d(shift2, "shift2")
        if scale2.stride(0) != a.ld_mod or shift2.stride(0) != a.ld_mod:
            raise RuntimeError("scale2/shift2 must share the row stride of scale/shift")
        a.y2, a.ldy2, a.scale2, a.shift2 = out2.data_ptr(), out2.stride(0), scale2.data_ptr(), shift2.data_ptr()
    if addvec is not None:
        _chk2d(addvec, "addvec")
        a.addvec, a.ld_add, a.rows_per_add = addvec.data_ptr(), addvec.stride(0), rows_per_add
        if xsum is not None:
            _chk2d(xsum, "xsum", torch.float32)
            a.xsum, a.ldxsum = xsum.data_ptr(), xsum.stride(0)
    _lib.check(_lib.load().dwm_layernorm_x32(C.byref(a), _stream()), "dwm_layernorm_x32")
    return out


def rmsnorm_heads_(x: torch.Tensor, w_expanded: torch.Tensor, eps: float) -> torch.Tensor:
    """In-place per-64-wide-head RMSNorm of x[rows, ncols]; w_expanded [ncols]."""
    _chk2d(x, "x")
    _chkvec(w_expanded, "w")
    _lib.check(_lib.load().dwm_rmsnorm_heads(x.data_ptr(), x.stride(0), x.shape[0], x.shape[1],
                                             w_expanded.data_ptr(), eps, _stream()), "dwm_rmsnorm_heads")
    return x


# -------------------------------------------------------------------------- elementwise
def silu(x: torch.Tensor) -> torch.Tensor:
    if x.dtype == torch.float32:
        _chkvec(x, "x", torch.float32)
        y = torch.empty_like(x)
        _lib.check(_lib.load().dwm_silu_f32(x.data_ptr(), y.data_ptr(), x.numel(), _stream()), "dwm_silu_f32")
        return y
    _chkvec(x, "x")
    y = torch.empty_like(x)
    _lib.check(_lib.load().dwm_silu(x.data_ptr(), y.data_ptr(), x.numel(), _stream()), "dwm_silu")
    return y


def timestep_sinusoid(t: torch.Tensor, channels: int, dtype: torch.dtype = bf16) -> torch.Tensor:
    """diffusers Timesteps(channels, flip_sin_to_cos=True, downscale_freq_shift=0) -> [n, channels] bf16 (or fp32: the accuracy path)."""
    t = t.reshape(-1).to(torch.float32).contiguous()
    if not t.is_cuda:
        raise RuntimeError("timestep_sinusoid: expected a device tensor")
    if dtype == torch.float32:
        out = torch.empty((t.numel(), channels), dtype=torch.float32, device=t.device)
        _lib.check(_lib.load().dwm_timestep_sinusoid_f32(t.data_ptr(), t.numel(), channels, out.data_ptr(), _stream()),
                   "dwm_timestep_sinusoid_f32")
        return out
    out = torch.empty((t.numel(), channels), dtype=bf16, device=t.device)
    _lib.check(_lib.load().dwm_timestep_sinusoid(t.data_ptr(), t.numel(), channels, out.data_ptr(), _stream()),
               "dwm_timestep_sinusoid")
    return out


def patchify(x: torch.Tensor, p: int, ldo: Optional[int] = None, dtype: torch.dtype = bf16) -> torch.Tensor:
    """[I, C, H, W] (fp32 / bf16) -> [I*(H/p)*(W/p), ldo] im2col rows (zero padded to ldo), bf16 (or fp32: the accuracy path)."""
    if not x.is_cuda or x.dim() != 4 or not x.is_contiguous() or x.dtype not in (torch.float32, bf16):
        raise RuntimeError("patchify: expected a contiguous fp32/bf16 [I,C,H,W] device tensor")
    I, Cc, H, W = x.shape
    cols = Cc * p * p
    ldo = ldo or (cols + 63) // 64 * 64
    if dtype == torch.float32:
        if x.dtype != torch.float32:
            raise RuntimeError("patchify: the fp32 path takes fp32 images")
        out = torch.empty((I * (H // p) * (W // p), ldo), dtype=torch.float32, device=x.device)
        _lib.check(_lib.load().dwm_patchify_f32(x.data_ptr(), I, Cc, H, W, p, out.data_ptr(), ldo, _stream()), "dwm_patchify_f32")
        return out
    out = torch.empty((I * (H // p) * (W // p), ldo), dtype=bf16, device=x.device)
    _lib.check(_lib.load().dwm_patchify(x.data_ptr(), int(x.dtype == torch.float32), I, Cc, H, W, p,
                                        out.data_ptr(), ldo, _stream()), "dwm_patchify")
    return out


def unpatchify(x: torch.Tensor, I: int, Cc: int, h: int, w: int, p: int) -> torch.Tensor:
    if x.dtype == torch.float32:
        _chk2d(x, "x", torch.float32)
        out = torch.empty((I, Cc, h * p, w * p), dtype=torch.float32, device=x.device)
        _lib.check(_lib.load().dwm_unpatchify_f32(x.data_ptr(), x.stride(0), I, Cc, h, w, p, out.data_ptr(), _stream()),
                   "dwm_unpatchify_f32")
        return out
    _chk2d(x, "x")
    out = torch.empty((I, Cc, h * p, w * p), dtype=bf16, device=x.device)
    _lib.check(_lib.load().dwm_unpatchify(x.data_ptr(), x.stride(0), I, Cc, h, w, p, out.data_ptr(), _stream()),
               "dwm_unpatchify")
    return out


def cfg_euler_step(pred: torch.Tensor, latents: torch.Tensor, guidance: float, dsigma,
                   model_in: Optional[torch.Tensor] = None, group_elems: int = 0) -> None:
    """latents(fp32, in place) += dsigma * (u + g (c - u)) with pred = [uncond; cond] bf16 (fp32 pred + fp32 model_in: the
    accuracy path)."""
    n = latents.numel()
    if pred.dtype == torch.float32:
        if pred.numel() != 2 * n or not pred.is_contiguous() or not pred.is_cuda or latents.dtype != torch.float32 or not latents.is_contiguous():
            raise RuntimeError("cfg_euler_step: fp32 pred must be contiguous with 2x the (fp32, contiguous) latent elements")
        if model_in is not None and (model_in.dtype != torch.float32 or model_in.numel() != 2 * n or not model_in.is_contiguous()):
            raise RuntimeError("cfg_euler_step: model_in of the fp32 path must be contiguous fp32 [2, n]")
        grp = dsigma if torch.is_tensor(dsigma) else None
        if grp is not None and (grp.dtype != torch.float32 or not grp.is_cuda or not grp.is_contiguous() or grp.numel() * group_elems != n):
            raise RuntimeError("cfg_euler_step: dsigma must be a contiguous fp32 device tensor with n / group_elems entries")
        _lib.check(_lib.load().dwm_cfg_euler_step_f32(pred.data_ptr(), latents.data_ptr(), _p(model_in), n, float(guidance),
                                                      0.0 if grp is not None else float(dsigma), _p(grp), group_elems, _stream()),
                   "dwm_cfg_euler_step_f32")
        return
    if pred.dtype != bf16 or pred.numel() != 2 * n or not pred.is_contiguous() or not pred.is_cuda:
        raise RuntimeError("cfg_euler_step: pred must be contiguous bf16 with 2x the latent elements")
    if latents.dtype != torch.float32 or not latents.is_contiguous() or not latents.is_cuda:
        raise RuntimeError("cfg_euler_step: latents must be contiguous fp32")
    if model_in is not None and (model_in.dtype != bf16 or model_in.numel() != 2 * n or not model_in.is_contiguous()):
        raise RuntimeError("cfg_euler_step: model_in must be contiguous bf16 [2, n]")
    if torch.is_tensor(dsigma):          # per-frame steps [n / group_elems] fp32 (diffusion forcing)
        if dsigma.dtype != torch.float32 or not dsigma.is_cuda or not dsigma.is_contiguous() or dsigma.numel() * group_elems != n:
            raise RuntimeError("cfg_euler_step: dsigma must be a contiguous fp32 device tensor with n / group_elems entries")
        _lib.check(_lib.load().dwm_cfg_euler_step_grouped(pred.data_ptr(), latents.data_ptr(), _p(model_in), n, float(guidance),
                                                          dsigma.data_ptr(), group_elems, _stream()), "dwm_cfg_euler_step_grouped")
        return
    _lib.check(_lib.load().dwm_cfg_euler_step(pred.data_ptr(), latents.data_ptr(), _p(model_in), n,
                                              float(guidance), float(dsigma), _stream()), "dwm_cfg_euler_step")


def cfg_multistep(pred: torch.Tensor, latents: torch.Tensor, x0_prev: torch.Tensor, guidance: float, kx: float, ko: float,
                  A: float, B: float, Cc: float, model_in: Optional[torch.Tensor] = None) -> None:
    """CFG combine + linear multistep scheduler update (see dwm_cfg_multistep); latents / x0_prev fp32 in place."""
    n = latents.numel()
    if pred.dtype not in (bf16, torch.float32) or pred.numel() != 2 * n or not pred.is_contiguous() or not pred.is_cuda:
        raise RuntimeError("cfg_multistep: pred must be contiguous bf16 (fp32: the accuracy path) with 2x the latent elements")
    for t in (latents, x0_prev):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != n:
            raise RuntimeError("cfg_multistep: latents / x0_prev must be contiguous fp32 of the same size")
    if model_in is not None and (model_in.dtype != pred.dtype or model_in.numel() != 2 * n or not model_in.is_contiguous()):
        raise RuntimeError("cfg_multistep: model_in must be contiguous, of pred's dtype and size")
    if pred.dtype == torch.float32:
        _lib.check(_lib.load().dwm_cfg_multistep_f32(pred.data_ptr(), latents.data_ptr(), x0_prev.data_ptr(), _p(model_in), n,
                                                     float(guidance), float(kx), float(ko), float(A), float(B), float(Cc), _stream()),
                   "dwm_cfg_multistep_f32")
        return
    _lib.check(_lib.load().dwm_cfg_multistep(pred.data_ptr(), latents.data_ptr(), x0_prev.data_ptr(), _p(model_in), n,
                                             float(guidance), float(kx), float(ko), float(A), float(B), float(Cc), _stream()),
               "dwm_cfg_multistep")


def ray_features(cam: torch.Tensor, h: int, w: int, ldo: int = 128, dtype: torch.dtype = bf16) -> torch.Tensor:
    """cam fp32 [I, 21] (see dwm_ray_features) -> bf16 (or, the fp32 accuracy path, fp32) [I*h*w, ldo]: RayEncoder's 72
    positional-encoding inputs per token."""
    if cam.dtype != torch.float32 or cam.dim() != 2 or cam.shape[1] != 21 or not cam.is_contiguous() or not cam.is_cuda:
        raise RuntimeError("ray_features: cam must be a contiguous fp32 device tensor [I, 21]")
    if dtype not in (bf16, torch.float32):
        raise RuntimeError("ray_features: bf16 or fp32 output")
    out = torch.empty((cam.shape[0] * h * w, ldo), dtype=dtype, device=cam.device)
    fn = "dwm_ray_features_f32" if dtype == torch.float32 else "dwm_ray_features"
    _lib.check(getattr(_lib.load(), fn)(cam.data_ptr(), cam.shape[0], h, w, out.data_ptr(), ldo, _stream()), fn)
    return out


def frame_affine(x: torch.Tensor, y: torch.Tensor, coef: torch.Tensor, group_elems: int, out: Optional[torch.Tensor] = None,
                 out_bf16: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = coef[g, 0] * x + coef[g, 1] * y (fp32), g = element // group_elems; see dwm_frame_affine."""
    n = x.numel()
    for t in (x, y):
        if t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda or t.numel() != n:
            raise RuntimeError("frame_affine: x / y must be contiguous fp32 device tensors of one size")
    if coef.dtype != torch.float32 or not coef.is_contiguous() or not coef.is_cuda or coef.numel() * group_elems != 2 * n:
        raise RuntimeError("frame_affine: coef must be a contiguous fp32 device tensor [n / group_elems, 2]")
    if out is None and out_bf16 is None:
        out = torch.empty_like(x)
    _lib.check(_lib.load().dwm_frame_affine(x.data_ptr(), y.data_ptr(), coef.data_ptr(), _p(out), _p(out_bf16), n, group_elems,
                                            _stream()), "dwm_frame_affine")
    return out if out is not None else out_bf16


def cfg_ddim_step(pred: torch.Tensor, latents: torch.Tensor, coef: torch.Tensor, group_elems: int, prediction_type: int,
                  guidance: Optional[float] = None, clip_range: float = 0.0, use_clipped_model_output: bool = False,
                  noise: Optional[torch.Tensor] = None, x0_out: Optional[torch.Tensor] = None,
                  model_in: Optional[torch.Tensor] = None) -> None:
    """[CFG combine +] tensor-timestep DDIM update of `latents` (fp32, in place); see dwm_cfg_ddim_step.
    guidance None: pred holds n elements; otherwise pred = [uncond; cond] with 2n."""
    n = latents.numel()
    cfg = guidance is not None
    if pred.dtype not in (bf16, torch.float32) or pred.numel() != (2 * n if cfg else n) or not pred.is_contiguous() or not pred.is_cuda:
        raise RuntimeError("cfg_ddim_step: pred must be a contiguous bf16 / fp32 device tensor with n (2n with guidance) elements")
    for t in (latents, noise, x0_out):
        if t is not None and (t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != n or not t.is_cuda):
            raise RuntimeError("cfg_ddim_step: latents / noise / x0_out must be contiguous fp32 device tensors of one size")
    if coef.dtype != torch.float32 or not coef.is_contiguous() or not coef.is_cuda or coef.numel() * group_elems != 6 * n:
        raise RuntimeError("cfg_ddim_step: coef must be a contiguous fp32 device tensor [n / group_elems, 6]")
    if model_in is not None and (model_in.dtype != bf16 or model_in.numel() != (2 * n if cfg else n) or not model_in.is_contiguous()):
        raise RuntimeError("cfg_ddim_step: model_in must be contiguous bf16")
    _lib.check(_lib.load().dwm_cfg_ddim_step(pred.data_ptr(), int(pred.dtype == torch.float32), int(cfg), latents.data_ptr(), _p(model_in),
                                             _p(x0_out), _p(noise), coef.data_ptr(), n, group_elems, float(guidance or 0.0),
                                             int(prediction_type), float(clip_range), int(bool(use_clipped_model_output)), _stream()),
               "dwm_cfg_ddim_step")


def unshuffle_tokens(x: torch.Tensor, r: int, ldo: Optional[int] = None, dtype: torch.dtype = bf16) -> torch.Tensor:
    """PixelUnshuffle(r): [I, C, H, W] (fp32 / bf16) -> token-major [I*(H/r)*(W/r), ldo], bf16 or (dtype=torch.float32, fp32
    input: the accuracy path) fp32; columns past C r r are zero."""
    if not x.is_cuda or x.dim() != 4 or not x.is_contiguous() or x.dtype not in (torch.float32, bf16):
        raise RuntimeError("unshuffle_tokens: expected a contiguous fp32/bf16 [I,C,H,W] device tensor")
    I, Cc, H, W = x.shape
    cols = Cc * r * r
    ldo = ldo or (cols + 63) // 64 * 64
    if dtype == torch.float32:
        if x.dtype != torch.float32:
            raise RuntimeError("unshuffle_tokens: the fp32 form takes fp32 images")
        out = torch.empty((I * (H // r) * (W // r), ldo), dtype=torch.float32, device=x.device)
        _lib.check(_lib.load().dwm_unshuffle_tokens_f32(x.data_ptr(), I, Cc, H, W, r, out.data_ptr(), ldo, _stream()), "dwm_unshuffle_tokens_f32")
        return out
    out = torch.empty((I * (H // r) * (W // r), ldo), dtype=bf16, device=x.device)
    _lib.check(_lib.load().dwm_unshuffle_tokens(x.data_ptr(), int(x.dtype == torch.float32), I, Cc, H, W, r,
                                                out.data_ptr(), ldo, _stream()), "dwm_unshuffle_tokens")
    return out


def avgpool2_tokens(x: torch.Tensor, I: int, h: int, w: int) -> torch.Tensor:
    """AvgPool2d(2) on token-major [I*h*w, C] -> [I*(h/2)*(w/2), C]."""
    dt = torch.float32 if x.dtype == torch.float32 else bf16
    _chk2d(x, "x", dt)
    if not x.is_contiguous() or x.shape[0] != I * h * w:
        raise RuntimeError("avgpool2_tokens: x must be contiguous [I*h*w, C]")
    out = torch.empty((I * (h // 2) * (w // 2), x.shape[1]), dtype=dt, device=x.device)
    fn = _lib.load().dwm_avgpool2_tokens_f32 if dt == torch.float32 else _lib.load().dwm_avgpool2_tokens
    _lib.check(fn(x.data_ptr(), I, h, w, x.shape[1], out.data_ptr(), _stream()), "dwm_avgpool2_tokens")
    return out


def add_(y: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """y += x (contiguous device tensors of one shape).  y bf16: x bf16, or fp32 (the fp32 sum is rounded once).  y fp32 (the
    fp32 residual stream): x fp32 or bf16."""
    if y.shape != x.shape or y.dtype not in (bf16, torch.float32) or x.dtype not in (bf16, torch.float32) \
            or not y.is_contiguous() or not x.is_contiguous() or not y.is_cuda:
        raise RuntimeError("add_: expected contiguous device tensors of one shape (bf16 / fp32)")
    if y.dtype == torch.float32:
        if x.dtype == torch.float32:
            _lib.check(_lib.load().dwm_add_f32_f32_inplace(y.data_ptr(), x.data_ptr(), y.numel(), _stream()), "dwm_add_f32_f32_inplace")
        else:
            cast_f32(x, out=y, accumulate=True)
    elif x.dtype == torch.float32:
        _lib.check(_lib.load().dwm_add_f32_inplace(y.data_ptr(), x.data_ptr(), y.numel(), _stream()), "dwm_add_f32_inplace")
    else:
        _lib.check(_lib.load().dwm_add_inplace(y.data_ptr(), x.data_ptr(), y.numel(), _stream()), "dwm_add_inplace")
    return y


def groupnorm_silu(x: torch.Tensor, I: int, P: int, gamma: torch.Tensor, beta: torch.Tensor, groups: int,
                   eps: float, silu: bool = True, out: Optional[torch.Tensor] = None,
                   out_grid=None, img_map: Optional[tuple] = None, zmap: Optional[dict] = None) -> torch.Tensor:
    """GroupNorm(groups) [+ SiLU] of token-major x [I*P, C]; with out_grid the result lands in the
    interior of a padded grid `out` [out_grid.rows, C] whose border must already be zero.  img_map =
    (iv, pn, s_ihi, s_ilo, s_phi): image i / pixel p -> token row (see dwm_groupnorm_silu_mapped).
    zmap = dict(mod [z rows, 2C], frames, videos, h, w, shift, zt): CogVideoXSpatialNorm3D (dwm_groupnorm_spatial)."""
    dt = torch.float32 if x.dtype == torch.float32 else bf16           # fp32: the accuracy path (dwm_groupnorm_silu_f32)
    _chk2d(x, "x", dt)
    if not x.is_contiguous() or x.shape[0] != I * P:
        raise RuntimeError("groupnorm_silu: x must be contiguous [I*P, C]")
    Cc = x.shape[1]
    _chkvec(gamma, "gamma", dt)
    _chkvec(beta, "beta", dt)
    rows = out_grid.rows if out_grid is not None else I * P
    if out is None:
        out = (torch.zeros if out_grid is not None else torch.empty)((rows, Cc), dtype=dt, device=x.device)
    if out.shape != (rows, Cc) or not out.is_contiguous() or out.dtype != dt:
        raise RuntimeError("groupnorm_silu: bad out")
    stats = torch.empty(_lib.load().dwm_groupnorm_stats_floats(I, P, groups), dtype=torch.float32, device=x.device)
    m = _lib.RowMap2D()
    if out_grid is not None:
        out_grid.fill(m)
    im = _lib.GnImgMap()
    if img_map is not None:
        im.iv, im.pn, im.s_ihi, im.s_ilo, im.s_phi = img_map
    if dt == torch.float32 and zmap is None:
        _lib.check(_lib.load().dwm_groupnorm_silu_f32(x.data_ptr(), out.data_ptr(), I, P, Cc, groups, eps, gamma.data_ptr(),
                                                      beta.data_ptr(), int(silu), stats.data_ptr(), C.byref(m), C.byref(im),
                                                      _stream()), "dwm_groupnorm_silu_f32")
        return out
    if zmap is not None:
        mod = zmap["mod"]
        _chk2d(mod, "zmap.mod", dt)
        zm = _lib.GnZMap()
        zm.mod, zm.ld_mod = mod.data_ptr(), mod.stride(0)
        zm.frames, zm.videos, zm.h, zm.w, zm.shift = zmap["frames"], zmap["videos"], zmap["h"], zmap["w"], zmap["shift"]
        for t, z in enumerate(zmap["zt"]):
            zm.zt[t] = z
        need = ((max(zmap["zt"]) + 1) * zmap["videos"]) * (zmap["h"] >> zmap["shift"]) * (zmap["w"] >> zmap["shift"])
        if mod.shape[0] < need or mod.shape[1] < 2 * Cc:
            raise RuntimeError("groupnorm_silu: zmap.mod is smaller than the latent grid it is indexed with")
        fn = "dwm_groupnorm_spatial_f32" if dt == torch.float32 else "dwm_groupnorm_spatial"
        _lib.check(getattr(_lib.load(), fn)(x.data_ptr(), out.data_ptr(), I, P, Cc, groups, eps, gamma.data_ptr(), beta.data_ptr(),
                                            int(silu), stats.data_ptr(), C.byref(m), C.byref(im), C.byref(zm), _stream()), fn)
        return out
    _lib.check(_lib.load().dwm_groupnorm_silu_mapped(x.data_ptr(), out.data_ptr(), I, P, Cc, groups, eps, gamma.data_ptr(),
                                                     beta.data_ptr(), int(silu), stats.data_ptr(), C.byref(m), C.byref(im),
                                                     _stream()), "dwm_groupnorm_silu_mapped")
    return out


def frame_mix(x: torch.Tensor, frame_elems: int, f0, f1, w0, w1, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out frame j = w0[j] * x[f0[j]] + w1[j] * x[f1[j]] over frames of `frame_elems` contiguous bf16 elements of the
    flat tensor x (temporal average pooling / temporal nearest upsampling of the CogVideoX VAE)."""
    n = len(f0)
    dt = x.dtype
    if dt not in (bf16, torch.float32) or not x.is_cuda or not x.is_contiguous() or x.numel() % frame_elems != 0:
        raise RuntimeError("frame_mix: x must be a contiguous bf16 (or, the fp32 accuracy path, fp32) device tensor of whole frames")
    nin = x.numel() // frame_elems
    if not (len(f1) == len(w0) == len(w1) == n) or n == 0 or n > 64 or max(max(f0), max(f1)) >= nin:
        raise RuntimeError("frame_mix: bad frame tables")
    if out is None:
        out = torch.empty(n * frame_elems, dtype=dt, device=x.device)
    if out.numel() != n * frame_elems or out.dtype != dt or not out.is_contiguous():
        raise RuntimeError("frame_mix: bad out")
    fm = _lib.FrameMix()
    fm.n_out = n
    for j in range(n):
        fm.f0[j], fm.f1[j], fm.w0[j], fm.w1[j] = f0[j], f1[j], w0[j], w1[j]
    fn = "dwm_frame_mix_f32" if dt == torch.float32 else "dwm_frame_mix_bf16"
    _lib.check(getattr(_lib.load(), fn)(x.data_ptr(), out.data_ptr(), frame_elems, C.byref(fm), _stream()), fn)
    return out


def upsample2_padded(x: torch.Tensor, I: int, h: int, w: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """nearest 2x upsample of token-major [I*h*w, C] into the padded grid of the [I, 2h, 2w] image."""
    dt = torch.float32 if x.dtype == torch.float32 else bf16
    _chk2d(x, "x", dt)
    if not x.is_contiguous() or x.shape[0] != I * h * w:
        raise RuntimeError("upsample2_padded: x must be contiguous [I*h*w, C]")
    g = PaddedGrid(I, 2 * h, 2 * w)
    if out is None:
        out = torch.zeros((g.rows, x.shape[1]), dtype=dt, device=x.device)
    _chk2d(out, "out", dt)
    fn = _lib.load().dwm_upsample2_padded_f32 if dt == torch.float32 else _lib.load().dwm_upsample2_padded
    _lib.check(fn(x.data_ptr(), out.data_ptr(), I, h, w, x.shape[1], _stream()), "dwm_upsample2_padded")
    return out


def pad_tokens(x: torch.Tensor, grid: PaddedGrid, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """compact token rows [grid.pixels, C] -> interior of a zero-bordered padded grid [grid.rows, C]."""
    dt = torch.float32 if x.dtype == torch.float32 else bf16
    _chk2d(x, "x", dt)
    if not x.is_contiguous() or x.shape[0] != grid.pixels:
        raise RuntimeError("pad_tokens: x must be contiguous [grid.pixels, C]")
    if out is None:
        out = torch.zeros((grid.rows, x.shape[1]), dtype=dt, device=x.device)
    _chk2d(out, "out", dt)
    m = _lib.RowMap2D()
    grid.fill(m)
    fn = _lib.load().dwm_pad_tokens_f32 if dt == torch.float32 else _lib.load().dwm_pad_tokens
    _lib.check(fn(x.data_ptr(), out.data_ptr(), x.shape[0], x.shape[1], C.byref(m), _stream()), "dwm_pad_tokens")
    return out


def softmax_rows(x: torch.Tensor, scale: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    dt = torch.float32 if x.dtype == torch.float32 else bf16
    _chk2d(x, "x", dt)
    out = torch.empty_like(x) if out is None else out
    _chk2d(out, "out", dt)
    fn = _lib.load().dwm_softmax_rows_f32 if dt == torch.float32 else _lib.load().dwm_softmax_rows
    _lib.check(fn(x.data_ptr(), out.data_ptr(), x.shape[0], x.shape[1], x.stride(0), float(scale), _stream()), "dwm_softmax_rows")
    return out


def cast_bf16(x: torch.Tensor) -> torch.Tensor:
    if x.dtype == bf16:
        return x
    if x.dtype != torch.float32 or not x.is_cuda:
        raise RuntimeError("cast_bf16: expected an fp32 device tensor")
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=bf16, device=x.device)
    _lib.check(_lib.load().dwm_cast_f32_to_bf16(x.data_ptr(), out.data_ptr(), x.numel(), _stream()),
               "dwm_cast_f32_to_bf16")
    return out


def cast_f32(x: torch.Tensor, out: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    """fp32 (+)= bf16, 2-D row-strided or contiguous (dwm_cast_bf16_to_f32): the entry of the fp32 residual stream"""
    if x.dtype != bf16 or not x.is_cuda:
        raise RuntimeError("cast_f32: expected a bf16 device tensor")
    x2 = x if x.dim() == 2 else x.reshape(1, -1)
    _chk2d(x2, "x")
    if out is None:
        out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        accumulate = False
    o2 = out if out.dim() == 2 else out.reshape(1, -1)
    _chk2d(o2, "out", torch.float32)
    if o2.shape != x2.shape:
        raise RuntimeError("cast_f32: shape mismatch")
    _lib.check(_lib.load().dwm_cast_bf16_to_f32(x2.data_ptr(), x2.stride(0), o2.data_ptr(), o2.stride(0), x2.shape[0], x2.shape[1],
                                                int(accumulate), _stream()), "dwm_cast_bf16_to_f32")
    return out


def tr_probe(offsets: torch.Tensor) -> torch.Tensor:
    """Diagnostic: semantics probe of ds_read_b64_tr_b16 (tests only)."""
    offsets = offsets.to(torch.int32).contiguous()
    out = torch.empty((64, 4), dtype=torch.int16, device=offsets.device)
    _lib.check(_lib.load().dwm_debug_tr_probe(offsets.data_ptr(), out.data_ptr(), _stream()), "dwm_debug_tr_probe")
    return out
