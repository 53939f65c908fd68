"""Generates tests/golden/*.pt from the CPU oracle (oracle/ctsd_oracle.py).

The reference cannot be imported in the build container (diffusers==0.31.0 absent,
SURVEY.md §8c) and ships no golden tensors, so these fixtures pin the ORACLE (and
through it the HIP path) against regressions; they are not reference outputs —
"parity unpinned" in oracle/ctsd_oracle.py still applies.

    python tests/golden/make_golden.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import ctsd_oracle as O            # noqa: E402
from tests.common import small_config, small_inputs, GOLDEN   # noqa: E402


def unet_small_config():
    from oracle import unet_oracle as U
    return U.make_unet_config(block_out_channels=(128, 256, 512, 512), num_attention_heads=(2, 4, 8, 8), cross_attention_dim=128,
                              projection_class_embeddings_input_dim=11 * 256)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(4)
    cfg = small_config()
    sd = O.make_state_dict(cfg, seed=0)
    inp = small_inputs(cfg, seed=0)
    trace = {}
    y = O.dit_forward(sd, cfg, trace=trace, **inp)
    keep = {k: v.to(torch.float32) for k, v in trace.items() if k in ("hidden0", "joint0", "joint1", "crossview1", "temporal3")}
    torch.save({"output": y.float(), "trace": keep}, os.path.join(GOLDEN, "dit_small_forward.pt"))

    # two CFG denoise steps on fixed noise (ctsd.py:1496-1575)
    g = torch.Generator().manual_seed(7)
    lat = torch.randn(1, 3, 3, 16, 8, 12, generator=g)
    cond = {k: v for k, v in inp.items() if k not in ("sample", "timestep")}
    out = O.denoise(sd, cfg, lat, cond, steps=4, guidance_scale=4.0, stop=2)
    torch.save({"latents_in": lat, "latents_out": out.float()}, os.path.join(GOLDEN, "denoise_small_2steps.pt"))

    # pointwise-temporal / full-temporal variants of one forward
    for tt in ("pointwise", "full"):
        c2 = small_config(temporal_attention_type=tt)
        y2 = O.dit_forward(sd, c2, **inp)
        torch.save({"output": y2.float()}, os.path.join(GOLDEN, f"dit_small_forward_{tt}.pt"))
    # SD 2.1 UNet (oracle/unet_oracle.py): one forward + two DPM-Solver++ CFG steps at small width
    from oracle import unet_oracle as U
    ucfg = unet_small_config()
    usd = U.make_unet_state_dict(ucfg, 0)
    uinp = U.make_unet_inputs(ucfg, 2, 2, 3, 8, 16, text_len=10)
    uy = U.unet_forward(usd, ucfg, **uinp)
    ucond = {k: v for k, v in uinp.items() if k not in ("sample", "timesteps")}
    uout = U.unet_denoise(usd, ucfg, uinp["sample"][:1], ucond, steps=4, guidance_scale=3.0, stop=2)
    torch.save({"output": uy.float(), "denoise_2steps": uout.float()}, os.path.join(GOLDEN, "unet_small.pt"))
    print("written to", GOLDEN)


if __name__ == "__main__":
    main()
