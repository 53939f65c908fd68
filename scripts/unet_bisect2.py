"""Diagnostic (GPU box): ONE TransformerModel of the full-width SD 2.1 UNet (down_blocks.<lvl>.attentions.0) on random input
at the BASELINE configs[1] geometry, HIP vs the fp32 oracle on the device, sub-block by sub-block (BasicTransformerBlock,
cross-view block + mixer, temporal block + mixer).   usage: python scripts/unet_bisect2.py [level] [T] [H] [W]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import unet_oracle as U          # noqa: E402
from opendwm_amd import blocks as BL, unet as UM      # noqa: E402
from tests.common import rel_err     # noqa: E402

bf16 = torch.bfloat16


def main():
    lvl = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    H = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    W = int(sys.argv[4]) if len(sys.argv) > 4 else 56
    B, V = 2, 6
    dev = torch.device("cuda:0")
    cfg = U.make_unet_config()
    sd = {k: v.to(bf16) for k, v in U.make_unet_state_dict(cfg, 0).items()}
    m = UM.UNetCrossviewTemporalConditionModel(**cfg)
    m.load_state_dict({k: v.float() for k, v in sd.items()}, strict=True)
    m = m.to(dev).to(bf16).eval()
    h, w = H >> lvl, W >> lvl
    Cc = cfg["block_out_channels"][lvl]
    heads = cfg["num_attention_heads"][lvl]
    tm = m.down_blocks[lvl].attentions[0]
    p = f"down_blocks.{lvl}.attentions.0"
    g = torch.Generator().manual_seed(5)
    x6 = torch.randn(B, T, V, Cc, h, w, generator=g).to(bf16).float()
    ehs = (torch.randn(B, T, V, 77, cfg["cross_attention_dim"], generator=g) * 0.5).to(bf16).float()
    mask = U.ring_crossview_mask(B, V)
    rec = []
    for cls, tag in ((UM.BasicTransformerBlock, "basic"), (BL.VTSelfAttentionBlock, "vt")):
        run0 = cls.run

        def run(self, *a, _run0=run0, _tag=tag, **kw):
            y = _run0(self, *a, **kw)
            rec.append((_tag, y.float().clone()))
            return y
        cls.run = run
    I, N = B * T * V, h * w
    x = x6.flatten(0, 2).flatten(2).transpose(1, 2).reshape(I * N, Cc).contiguous().to(dev).to(bf16)
    geo = UM._Geom(B, T, V, h, w, UM._Scratch())
    ctx = UM._TextContext(ehs.to(dev).to(bf16))
    zeros = torch.zeros(B, dtype=torch.bool, device=dev)
    with torch.no_grad():
        out = tm.run(x.clone(), ctx, geo, zeros, zeros, mask.to(dev)).float()
    sd_dev = {k: v.to(dev).float() for k, v in sd.items() if k.startswith(p)}
    idx = [0]

    def cmp(tag, want):
        t, mine = rec[idx[0]]
        e = rel_err(mine, want.reshape(mine.shape))
        print(f"{idx[0]:2d} ours={t:5s} oracle={tag:8s} rel={e:.4e}" + ("   <<<<" if e > 2e-2 else ""), flush=True)
        idx[0] += 1
    btb0, ab0 = U.basic_transformer_block, U.alpha_blender

    def btb(*a, **kw):
        y = btb0(*a, **kw)
        cmp("basic", y)
        return y

    def ab(*a, **kw):
        y = ab0(*a, **kw)
        cmp("mixer", y)
        return y
    U.basic_transformer_block, U.alpha_blender = btb, ab
    with torch.no_grad():
        ref = U.transformer_model(sd_dev, p, cfg, heads, x6.to(dev), ehs.to(dev), zeros, zeros, mask.to(dev),
                                  cfg["transformer_layers_per_block"] if isinstance(cfg["transformer_layers_per_block"], int)
                                  else cfg["transformer_layers_per_block"][lvl])
    want = ref.flatten(0, 2).flatten(2).transpose(1, 2).reshape(I * N, Cc)
    print("final", rel_err(out, want), "level", lvl, "C", Cc, "heads", heads, "T", T, "hxw", h, w)


if __name__ == "__main__":
    main()
