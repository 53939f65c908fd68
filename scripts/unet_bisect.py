"""Diagnostic (GPU box): the full-width SD 2.1 UNet forward, HIP vs the fp32 oracle on the device, block by block - the
output of every ResBlock / TransformerModel in call order - to locate the first block whose error jumps.
usage: python scripts/unet_bisect.py [T] [H] [W] [variant_override]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import unet_oracle as U          # noqa: E402
from opendwm_amd import ops, unet as UM      # noqa: E402
from tests.common import rel_err, to_dev     # noqa: E402

bf16 = torch.bfloat16


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    W = int(sys.argv[3]) if len(sys.argv) > 3 else 56
    override = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    dev = torch.device("cuda:0")
    if override:
        attn0 = ops.attention

        def attn(*a, **kw):
            kw["variant"] = kw.get("variant", 0) | override
            return attn0(*a, **kw)
        ops.attention = attn
    cfg = U.make_unet_config()
    sd = {k: v.to(bf16) for k, v in U.make_unet_state_dict(cfg, 0).items()}
    inp = U.make_unet_inputs(cfg, 2, T, 6, H, W, text_len=77)
    inp = {k: (v.to(bf16).float() if v.is_floating_point() and k not in ("timesteps", "added_time_ids") else v) for k, v in inp.items()}
    m = UM.UNetCrossviewTemporalConditionModel(**cfg)
    m.load_state_dict({k: v.float() for k, v in sd.items()}, strict=True)
    m = m.to(dev).to(bf16).eval()
    rec = []
    for cls, tag in ((UM.ResBlock, "res"), (UM.TransformerModel, "attn")):
        run0 = cls.run

        def run(self, x, *a, _run0=run0, _tag=tag, **kw):
            g = a[1]
            y = _run0(self, x, *a, **kw)
            rec.append((_tag, g.h, g.w, y.float().clone()))
            return y
        cls.run = run
    di = to_dev(inp, dev)
    with torch.no_grad():
        out = m(di.pop("sample"), di.pop("timesteps"), **di)[0][0].float()
    sd_dev = {k: v.to(dev).float() for k, v in sd.items()}
    idx = [0]

    def wrap(fn, tag):
        def f(*a, **kw):
            y = fn(*a, **kw)                                   # [B,T,V,C,h,w]
            t, h, w, mine = rec[idx[0]]
            want = y.flatten(0, 2).flatten(2).transpose(1, 2).reshape(-1, y.shape[3])
            e = rel_err(mine, want)
            print(f"{idx[0]:3d} {tag:5s} {a[1] if isinstance(a[1], str) else ''} level {h}x{w} C={y.shape[3]} rel={e:.4e}" +
                  ("   <<<<" if e > 3e-2 else ""), flush=True)
            idx[0] += 1
            return y
        return f
    U.res_block = wrap(U.res_block, "res")
    U.transformer_model = wrap(U.transformer_model, "attn")
    with torch.no_grad():
        ref = U.unet_forward(sd_dev, cfg, **to_dev(inp, dev))
    print("final", rel_err(out, ref), "T", T, "HxW", H, W, "override", override)


if __name__ == "__main__":
    main()
