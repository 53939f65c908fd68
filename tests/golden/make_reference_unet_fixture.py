"""Golden vector from EXECUTING the reference SD 2.1 UNet's own composition - runs only where /root/reference exists.

REAL reference code executed (classes imported behind an import-only `diffusers` stub, instances hand-built):
  * UNetCrossviewTemporalConditionModel.forward                                     crossview_temporal_unet.py:648-835
  * UNetMidBlock / DownBlock / CrossAttnDownBlock / UpBlock / CrossAttnUpBlock CrossviewTemporal .forward   :61-352
  * ResBlock.forward, TemporalBasicTransformerBlock.forward, TransformerModel.forward (+ its
    forward_crossview / forward_temporal_block_and_mix_result), AlphaBlender        crossview_temporal.py:9-514
Leaves (diffusers-built modules) are the oracle's restatements (oracle.unet_oracle) bound to one synthetic state dict:
ResnetBlock2D, TemporalResnetBlock, BasicTransformerBlock, FeedForward, Attention, Timesteps, TimestepEmbedding,
Downsample2D / Upsample2D; torch.nn.GroupNorm / LayerNorm / Linear / Conv2d / SiLU are the real thing.

usage: python tests/golden/make_reference_unet_fixture.py  ->  tests/golden/reference_unet_forward.pt
"""
import os
import sys

import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import ctsd_oracle as O                                      # noqa: E402
from oracle import unet_oracle as U                                      # noqa: E402
from tests.golden.make_golden import unet_small_config                   # noqa: E402
from tests.golden.make_reference_driver_fixtures import _Finder          # noqa: E402


def _mod(cls):
    m = object.__new__(cls)
    torch.nn.Module.__init__(m)
    m.gradient_checkpointing = False
    return m.eval()


def _load(layer, sd, p):
    layer.weight.data.copy_(sd[p + ".weight"])
    if layer.bias is not None:
        layer.bias.data.copy_(sd[p + ".bias"])
    return layer


def build(R, RU, cfg, sd):
    eps = cfg["norm_eps"]
    down, mid, up = U._block_plan(cfg)

    def mixer(p):
        mx = R.AlphaBlender(alpha=1.0, merge_strategy="learned_with_images")
        mx.mix_factor.data.copy_(sd[p + ".mix_factor"])
        return mx

    def res_block(p):
        b = _mod(R.ResBlock)
        b.spatial_res_block = lambda x, temb: U.resnet_block_2d(sd, p + ".spatial_res_block", x, temb, eps)
        if (p + ".temporal_res_block.conv1.weight") in sd:
            b.temporal_res_block = lambda x, temb: U.temporal_resnet_block(sd, p + ".temporal_res_block", x, temb, eps)
            b.time_mixer = mixer(p + ".time_mixer")
        else:
            b.temporal_res_block = None
        return b

    def tbt_block(p, C, heads):
        b = _mod(R.TemporalBasicTransformerBlock)
        b.is_res = True
        for nm in ("norm_in", "norm1", "norm3"):
            setattr(b, nm, _load(torch.nn.LayerNorm(C), sd, f"{p}.{nm}"))
        b.ff_in = lambda y: O.feed_forward(sd, p + ".ff_in", y, "geglu")
        b.ff = lambda y: O.feed_forward(sd, p + ".ff", y, "geglu")
        b.attn1 = lambda y, encoder_hidden_states=None, attention_mask=None: U._attention(sd, p + ".attn1", heads, y, mask=attention_mask)
        b.attn2 = None
        return b

    def transformer(p, C, heads, n_layers):
        t = _mod(R.TransformerModel)
        t.norm = _load(torch.nn.GroupNorm(32, C, eps=1e-6), sd, p + ".norm")
        t.proj_in = _load(torch.nn.Linear(C, C), sd, p + ".proj_in")
        t.proj_out = _load(torch.nn.Linear(C, C), sd, p + ".proj_out")
        t.time_proj = lambda idx: O.timesteps_sinusoid(idx, C)
        t.transformer_blocks = [
            (lambda h, encoder_hidden_states=None, q=f"{p}.transformer_blocks.{l}": U.basic_transformer_block(sd, q, heads, h, encoder_hidden_states))
            for l in range(n_layers)]
        has_cv = (p + ".view_pos_embed.linear_1.weight") in sd
        has_t = (p + ".time_pos_embed.linear_1.weight") in sd
        t.view_pos_embed = (lambda x: O.timestep_embedding_mlp(sd, p + ".view_pos_embed", x)) if has_cv else None
        t.time_pos_embed = (lambda x: O.timestep_embedding_mlp(sd, p + ".time_pos_embed", x)) if has_t else None
        t.crossview_transformer_blocks = [tbt_block(f"{p}.crossview_transformer_blocks.{l}", C, heads) if has_cv else None for l in range(n_layers)]
        t.temporal_transformer_blocks = [tbt_block(f"{p}.temporal_transformer_blocks.{l}", C, heads) if has_t else None for l in range(n_layers)]
        if has_cv:
            t.view_mixer = mixer(p + ".view_mixer")
        if has_t:
            t.time_mixer = mixer(p + ".time_mixer")
        t.enable_rowwise_crossview, t.enable_rowwise_temporal = cfg["enable_rowwise_crossview"], cfg["enable_rowwise_temporal"]
        return t

    m = _mod(RU.UNetCrossviewTemporalConditionModel)
    c0 = cfg["block_out_channels"][0]
    m.time_proj = lambda t: O.timesteps_sinusoid(t, c0)
    m.time_embedding = lambda x: O.timestep_embedding_mlp(sd, "time_embedding", x)
    m.add_time_proj = lambda t: O.timesteps_sinusoid(t, cfg["addition_time_embed_dim"])
    m.add_embedding = lambda x: O.timestep_embedding_mlp(sd, "add_embedding", x)
    m.condition_image_adapter = None
    m.depth_net = None
    m.conv_in = _load(torch.nn.Conv2d(cfg["in_channels"], c0, 3, padding=1), sd, "conv_in")
    m.conv_norm_out = _load(torch.nn.GroupNorm(32, c0, eps=1e-5), sd, "conv_norm_out")
    m.conv_act = torch.nn.SiLU()
    m.conv_out = _load(torch.nn.Conv2d(c0, cfg["out_channels"], 3, padding=1), sd, "conv_out")
    m.down_blocks = []
    for i, blk in enumerate(down):
        b = _mod(RU.CrossAttnDownBlockCrossviewTemporal if blk["attn"] else RU.DownBlockCrossviewTemporal)
        b.has_cross_attention = blk["attn"]
        b.resnets = [res_block(f"down_blocks.{i}.resnets.{j}") for j in range(len(blk["resnets"]))]
        if blk["attn"]:
            b.attentions = [transformer(f"down_blocks.{i}.attentions.{j}", blk["channels"], blk["heads"], blk["tlayers"]) for j in range(len(blk["resnets"]))]
        b.downsamplers = [lambda x, q=f"down_blocks.{i}.downsamplers.0.conv": U.conv2d(sd, q, x, stride=2)] if blk["downsample"] else None
        m.down_blocks.append(b)
    mb = _mod(RU.UNetMidBlockCrossviewTemporal)
    mb.has_cross_attention = True
    mb.resnets = [res_block("mid_block.resnets.0"), res_block("mid_block.resnets.1")]
    mb.attentions = [transformer("mid_block.attentions.0", mid["channels"], mid["heads"], mid["tlayers"])]
    m.mid_block = mb
    m.up_blocks = []
    for i, blk in enumerate(up):
        b = _mod(RU.CrossAttnUpBlockCrossviewTemporal if blk["attn"] else RU.UpBlockCrossviewTemporal)
        b.has_cross_attention = blk["attn"]
        b.resnets = [res_block(f"up_blocks.{i}.resnets.{j}") for j in range(len(blk["resnets"]))]
        if blk["attn"]:
            b.attentions = [transformer(f"up_blocks.{i}.attentions.{j}", blk["channels"], blk["heads"], blk["tlayers"]) for j in range(len(blk["resnets"]))]
        b.upsamplers = [lambda x, q=f"up_blocks.{i}.upsamplers.0.conv": U.conv2d(sd, q, F.interpolate(x, scale_factor=2.0, mode="nearest"))] \
            if blk["upsample"] else None
        m.up_blocks.append(b)
    return m


def main():
    sys.meta_path.insert(0, _Finder())
    sys.path.insert(0, "/root/reference/src")
    import dwm.models.crossview_temporal as R
    import dwm.models.crossview_temporal_unet as RU
    out = {}
    with torch.no_grad():
        acfg = dict(in_channels=3, channels=[128, 128, 256, 512, 512], is_downblocks=[False, True, True, True, False], num_res_blocks=2,
                    downscale_factor=8, use_zero_convs=True)
        for name, over in (("rowwise", {}), ("pointwise", dict(enable_rowwise_crossview=False, enable_rowwise_temporal=False)),
                           # layout branch: the REAL ImageAdapter.forward (adapters.py:40-60) feeding the residual insertion of the
                           # REAL UNet forward (crossview_temporal_unet.py:719-729, 753-754)
                           ("layout", dict(condition_image_adapter_config=acfg))):
            cfg = dict(unet_small_config(), **over)
            sd = U.make_unet_state_dict(cfg, 0)
            inp = U.make_unet_inputs(cfg, 2, 2, 3, 8, 16, text_len=10)
            inp["disable_temporal"] = torch.tensor([False, True])
            inp["disable_crossview"] = torch.tensor([True, False])
            if name == "pointwise":
                inp["crossview_attention_mask"] = None
            m = build(R, RU, cfg, sd)
            if name == "layout":
                from dwm.models.adapters import ImageAdapter as RefAdapter
                from tests.golden.make_reference_forward_fixture import build_adapter
                m.condition_image_adapter = build_adapter(RefAdapter, cfg, sd)
                inp["condition_image_tensor"] = torch.rand(2, 2, 3, 3, 64, 128, generator=torch.Generator().manual_seed(5))
            res = RU.UNetCrossviewTemporalConditionModel.forward(m, **inp)
            ref = res[0][0] if isinstance(res[0], tuple) else res[0]
            mine = U.unet_forward(sd, cfg, **inp)
            print(name, "reference UNet forward vs oracle: max abs diff", float((ref - mine).abs().max()), "| output std", float(ref.std()))
            out[name] = dict(output=ref.clone(), over=over, flags=(inp["disable_crossview"], inp["disable_temporal"]))
    torch.save(out, os.path.join(HERE, "reference_unet_forward.pt"))
    print("wrote reference_unet_forward.pt", {k: list(v["output"].shape) for k, v in out.items()})


if __name__ == "__main__":
    main()
