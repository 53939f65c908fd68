"""Intra-sample sharding of ONE multi-view video across GPUs (not in the reference, which only
ever runs whole samples per GPU: src/dwm/train.py:116-122).

Shard axis = frames.  With the frames of a sample split over R ranks

  * the joint MMDiT blocks, the embeddings, the ImageAdapter, the CFG combine and the scheduler update are per image
    -> local;
  * the cross-view blocks attend over the views of ONE frame (crossview_temporal_dit.py:223-327) -> local;
  * the temporal blocks attend over all frames of one (view, token row / token) (:329-370) -> the hidden state is
    re-sharded from "my frames, all token rows" to "all frames, my token rows" with ONE all-to-all before the block and
    one after it (RCCL over xGMI: every rank sends 1/R of its shard to every peer, which is the full-mesh traffic
    pattern the point-to-point links are built for).  Everything inside the temporal block except its attention is
    row-wise, so the block runs unchanged on the re-sharded rows.

Per denoise step at BASELINE config 3 / 8 GPUs: 24 all-to-alls of 33 MB per rank (29 MB leave the GPU).
Supported temporal attention types: "rowwise" and "pointwise" (a token row never mixes with another one); "full"
temporal attention couples every token of a view and would have to shard by view instead."""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist


def all_to_all_chunks(out: torch.Tensor, inp: torch.Tensor, group) -> None:
    """out[j] <- rank j's inp[my rank]; both [R, ...] contiguous.  RCCL for device tensors; the gloo backend (CPU
    tests, and the two-ranks-on-one-GPU test) only moves host memory, so device tensors are staged through it there."""
    if dist.get_backend(group) == "gloo" and inp.is_cuda:
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(o, inp.cpu(), group=group)
        out.copy_(o)
        return
    dist.all_to_all_single(out, inp, group=group)


def _permute_blocks(src: torch.Tensor, dims, src_strides, block_elems: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """a dense copy of `src`'s blocks in the order `dims`, block (i0..i3) taken from block index sum(i_k * src_strides[k]) of `src`:
    ONE HIP launch (dwm_block_permute) for device tensors - the pack / unpack of the exchange is not torch glue; host tensors (the
    gloo CPU tests) go through the equivalent torch view"""
    out = torch.empty_like(src) if out is None else out
    if out.numel() != src.numel() or out.dtype != src.dtype or not out.is_contiguous():
        raise ValueError("_permute_blocks: `out` must be a contiguous tensor of the source's size and dtype")
    if src.is_cuda:
        from . import ops
        return ops.block_permute(src, out, dims, src_strides, block_elems)
    flat = src.reshape(-1, block_elems)
    idx = sum(torch.arange(dims[k]).view([-1 if i == k else 1 for i in range(4)]) * src_strides[k] for k in range(4)).reshape(-1)
    out.view(-1, block_elems).copy_(flat[idx])
    return out


class FrameShard:
    """The frame-axis shard of one rank: rank r of R holds frames [r*T/R, (r+1)*T/R)."""

    def __init__(self, group=None):
        self.group = group if group is not None else dist.group.WORLD
        self.size = dist.get_world_size(self.group)
        self.rank = dist.get_rank(self.group)

    def frame_range(self, total_frames: int):
        if total_frames % self.size:
            raise ValueError(f"{total_frames} frames do not split over {self.size} ranks")
        n = total_frames // self.size
        return self.rank * n, (self.rank + 1) * n

    def check(self, height: int, temporal_attention_type: str):
        if temporal_attention_type not in ("rowwise", "pointwise"):
            raise NotImplementedError(f"frame sharding supports rowwise / pointwise temporal attention, not {temporal_attention_type!r}")
        if height % self.size:
            raise ValueError(f"{height} token rows do not split over {self.size} ranks")

    # ---- [B, Tl, V, height, width] rows of my frames  <->  [B, T, V, height / R, width] rows of all frames
    def frames_to_rows(self, h: torch.Tensor, B: int, Tl: int, V: int, height: int, width: int,
                       out: Optional[torch.Tensor] = None) -> torch.Tensor:
        R, D = self.size, h.shape[-1]
        hl = height // R
        blk = hl * width * D                                                                       # one (image, row block): contiguous
        h = h.contiguous()
        # pack: [b, tl, v, j] blocks -> [dest j][b, tl, v]
        send = _permute_blocks(h, (R, B, Tl, V), (1, Tl * V * R, V * R, R), blk)
        recv = torch.empty_like(send)
        all_to_all_chunks(recv.view(R, -1), send.view(R, -1), self.group)                          # [src j = frame block][b, tl, v]
        # unpack: -> [b, j, tl, v] = all frames (j, tl) of my token rows
        return _permute_blocks(recv, (B, R, Tl, V), (Tl * V, B * Tl * V, V, 1), blk, out=out).view(B * R * Tl * V * hl * width, D)

    def rows_to_frames(self, hx: torch.Tensor, B: int, Tl: int, V: int, height: int, width: int,
                       out: Optional[torch.Tensor] = None) -> torch.Tensor:
        R, D = self.size, hx.shape[-1]
        hl = height // R
        blk = hl * width * D
        hx = hx.contiguous()
        # pack: [b, j, tl, v] blocks -> [dest j = frame block][b, tl, v]
        send = _permute_blocks(hx, (R, B, Tl, V), (Tl * V, R * Tl * V, V, 1), blk)
        recv = torch.empty_like(send)
        all_to_all_chunks(recv.view(R, -1), send.view(R, -1), self.group)                          # [src j = row block][b, tl, v]
        # unpack: -> [b, tl, v, j] = all token rows of my frames
        return _permute_blocks(recv, (B, Tl, V, R), (Tl * V, V, 1, B * Tl * V), blk, out=out).view(B * Tl * V * height * width, D)

    def gather_frames(self, x: torch.Tensor, frame_dim: int = 1) -> torch.Tensor:
        """all ranks' frame blocks concatenated along `frame_dim` (per-image vectors, final latents)."""
        x = x.contiguous()
        parts = [torch.empty_like(x) for _ in range(self.size)]
        dist.all_gather(parts, x, group=self.group)
        return torch.cat(parts, frame_dim)
