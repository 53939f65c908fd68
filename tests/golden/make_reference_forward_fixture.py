"""Golden vector from EXECUTING the reference model's own forward logic - runs only where /root/reference exists.

The REAL methods DiTCrossviewTemporalConditionModel.forward (src/dwm/models/crossview_temporal_dit.py:372-630, with its
forward_crossview / forward_temporal_block_and_mix_result :223-370), VTSelfAttentionBlock.forward
(crossview_temporal.py:562-582) and AlphaBlender.forward (:9-72) are run on hand-built instances (the classes are imported
behind an import-only `diffusers` stub, see make_reference_fixtures.py).  Their diffusers-built LEAF modules - PatchEmbed,
context_embedder, CombinedTimestepTextProjEmbeddings, the Timesteps / TimestepEmbedding index embeddings,
JointTransformerBlock, FeedForward, Attention, AdaLayerNormContinuous, proj_out - are the oracle's restatements
(oracle.ctsd_oracle) bound to one synthetic state dict; torch.nn.LayerNorm is the real thing.

So the vector pins the COMPOSITION the reference owns: 6-D flattening, which embeddings are added where, the order of
joint / temporal / cross-view blocks per layer, mask and disable-flag plumbing, residual order inside the VT block, the
final un-patchify einsum.  It does not pin the leaves (diffusers arithmetic).

usage: python tests/golden/make_reference_forward_fixture.py  ->  tests/golden/reference_forward.pt
"""
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import ctsd_oracle as O                                      # noqa: E402
from tests.common import small_config, small_inputs                      # noqa: E402
from tests.golden.make_reference_fixtures import install_stub            # noqa: E402


def build_adapter(RefAdapter, cfg, sd):
    """the REAL ImageAdapter.forward (adapters.py:40-60) over oracle AdapterBlock leaves; zero convs are real Conv2d"""
    ac = cfg["condition_image_adapter_config"]
    a = object.__new__(RefAdapter)
    torch.nn.Module.__init__(a)
    a.eval()
    a.gradient_checkpointing = False
    a.unshuffle = torch.nn.PixelUnshuffle(ac["downscale_factor"])
    a.body = [(lambda x, i=i: O.adapter_block(sd, cfg, i, x)) for i in range(len(ac["channels"]))]
    a.zero_convs = []
    for i, ch in enumerate(ac["channels"]):
        z = torch.nn.Conv2d(ch, ch, 1)
        z.weight.data.copy_(sd[f"condition_image_adapter.zero_convs.{i}.weight"])
        z.bias.data.copy_(sd[f"condition_image_adapter.zero_convs.{i}.bias"])
        a.zero_convs.append(z)
    a.zero_gates = None
    return a


def build(RefDiT, VT, AlphaBlender, cfg, sd):
    D = cfg["num_attention_heads"] * cfg["attention_head_dim"]
    heads = cfg["num_attention_heads"]
    m = object.__new__(RefDiT)
    torch.nn.Module.__init__(m)
    m.eval()
    m.config = types.SimpleNamespace(patch_size=cfg["patch_size"])
    m.out_channels = cfg["out_channels"]
    m.mask_module = None
    m.condition_image_adapter = None
    m.perspective_modeling_type = cfg["perspective_modeling_type"]
    m.pos_embed = lambda x: O.patch_embed(sd, cfg, x)
    m.context_embedder = lambda e: O.linear(sd, "context_embedder", e)
    m.time_text_embed = lambda t, pooled: (
        O.timestep_embedding_mlp(sd, "time_text_embed.timestep_embedder", O.timesteps_sinusoid(t, 256).to(pooled.dtype))
        + O.timestep_embedding_mlp(sd, "time_text_embed.text_embedder", pooled))
    m.view_cam_proj = lambda ids: O.timesteps_sinusoid(ids, 256)
    m.view_embedding = lambda v: O.timestep_embedding_mlp(sd, "view_embedding", v)
    m.index_proj = lambda idx: O.timesteps_sinusoid(idx, D)
    m.transformer_blocks = [
        (lambda h, c, temb, i=i: O.joint_transformer_block(sd, f"transformer_blocks.{i}", cfg, i, h, c, temb))
        for i in range(cfg["num_layers"])]
    m.enable_temporal, m.enable_crossview = cfg["enable_temporal"], cfg["enable_crossview"]
    m.temporal_block_layers, m.crossview_block_layers = list(cfg["temporal_block_layers"]), list(cfg["crossview_block_layers"])
    m.temporal_attention_type, m.crossview_attention_type = cfg["temporal_attention_type"], cfg["crossview_attention_type"]
    m.disable_view_emb_on_temporal_module = cfg.get("disable_view_emb_on_temporal_module", False)

    def vt_block(prefix):
        b = object.__new__(VT)
        torch.nn.Module.__init__(b)
        for nm in ("norm_in", "norm1", "norm3"):
            ln = torch.nn.LayerNorm(D)
            ln.weight.data.copy_(sd[f"{prefix}.{nm}.weight"])
            ln.bias.data.copy_(sd[f"{prefix}.{nm}.bias"])
            setattr(b, nm, ln)
        b.ff_in = lambda y: O.feed_forward(sd, prefix + ".ff_in", y, "geglu")
        b.ff = lambda y: O.feed_forward(sd, prefix + ".ff", y, "geglu")
        b.attn1 = lambda y, encoder_hidden_states=None, attention_mask=None: O.vt_attention(sd, prefix + ".attn1", heads, y, attention_mask)
        return b.eval()

    def mixer(prefix):
        mx = AlphaBlender(alpha=1.0, merge_strategy="learned_with_images")
        mx.mix_factor.data.copy_(sd[prefix + ".mix_factor"])
        return mx
    m.temporal_transformer_blocks = [vt_block(f"temporal_transformer_blocks.{k}") for k in range(len(m.temporal_block_layers))]
    m.crossview_transformer_blocks = [vt_block(f"crossview_transformer_blocks.{k}") for k in range(len(m.crossview_block_layers))]
    m.time_mixers = [mixer(f"time_mixers.{k}") for k in range(len(m.temporal_block_layers))]
    m.view_mixers = [mixer(f"view_mixers.{k}") for k in range(len(m.crossview_block_layers))]
    m.time_pos_embeds = [(lambda x, k=k: O.timestep_embedding_mlp(sd, f"time_pos_embeds.{k}", x)) for k in range(len(m.temporal_block_layers))]
    m.view_pos_embeds = [(lambda x, k=k: O.timestep_embedding_mlp(sd, f"view_pos_embeds.{k}", x)) for k in range(len(m.crossview_block_layers))]

    def norm_out(h, temb):                               # diffusers AdaLayerNormContinuous (leaf)
        emb = O.linear(sd, "norm_out.linear", torch.nn.functional.silu(temb))
        scale, shift = emb.chunk(2, dim=1)
        return O.layer_norm_noaffine(h) * (1 + scale)[:, None] + shift[:, None]
    m.norm_out = norm_out
    m.proj_out = lambda h: O.linear(sd, "proj_out", h)
    return m


def main():
    install_stub()
    sys.path.insert(0, "/root/reference/src")
    from dwm.models.crossview_temporal import AlphaBlender, VTSelfAttentionBlock
    from dwm.models.crossview_temporal_dit import DiTCrossviewTemporalConditionModel as RefDiT
    out = {}
    with torch.no_grad():
        for tt in ("rowwise", "pointwise", "full"):
            cfg = small_config(temporal_attention_type=tt)
            sd = O.make_state_dict(small_config(), 0)
            inp = small_inputs(cfg, 0)
            inp["disable_temporal"] = torch.tensor([False, True])
            m = build(RefDiT, VTSelfAttentionBlock, AlphaBlender, cfg, sd)
            res, _, _ = RefDiT.forward(m, **inp)
            mine = O.dit_forward(sd, cfg, **inp)
            print(tt, "reference forward vs oracle forward: max abs diff", float((res[0] - mine).abs().max()))
            out[tt] = dict(output=res[0].clone(), disable_temporal=inp["disable_temporal"])
        # layout branch: the REAL ImageAdapter.forward feeding the residual insertion of the REAL model forward (:459-462, :491-494)
        from dwm.models.adapters import ImageAdapter as RefAdapter
        acfg = dict(in_channels=6, channels=[128, 128, 128], is_downblocks=[True, False, False], num_res_blocks=2,
                    downscale_factor=8, use_zero_convs=True)
        cfg = small_config(condition_image_adapter_config=acfg)
        sd = O.make_state_dict(cfg, 0)
        inp = small_inputs(cfg, 0)
        inp["condition_image_tensor"] = torch.rand(2, 3, 3, 6, 64, 96, generator=torch.Generator().manual_seed(5))
        m = build(RefDiT, VTSelfAttentionBlock, AlphaBlender, cfg, sd)
        m.condition_image_adapter = build_adapter(RefAdapter, cfg, sd)
        res, _, _ = RefDiT.forward(m, **inp)
        print("layout: reference forward vs oracle forward: max abs diff", float((res[0] - O.dit_forward(sd, cfg, **inp)).abs().max()))
        out["layout"] = dict(output=res[0].clone(), adapter_config=acfg)
        # 5-D inputs [B, T, C, H, W] (no view axis): the should_add_dim branch inserts V = 1
        sd = O.make_state_dict(small_config(), 0)
        cfg = small_config()
        inp = small_inputs(cfg, 0, V=1)
        five = {k: (v.squeeze(2) if torch.is_tensor(v) and v.dim() >= 3 and k != "crossview_attention_mask" else v) for k, v in inp.items()}
        five["disable_temporal"] = torch.zeros(2, 1, dtype=torch.bool)      # the reference unsqueezes it too (:400-401)
        m = build(RefDiT, VTSelfAttentionBlock, AlphaBlender, cfg, sd)
        res, _, _ = RefDiT.forward(m, **five)
        print("5-D vs oracle on the V = 1 6-D input:", float((res[0] - O.dit_forward(sd, cfg, **inp)).abs().max()), list(res[0].shape))
        out["five_dim"] = dict(output=res[0].clone())
        # explicit perspective modelling (:440-458): the REAL get_rays and RayEncoder.forward (pure torch) inside the REAL forward
        from dwm.models.crossview_temporal_dit import RayEncoder as RefRayEncoder, get_rays as ref_get_rays
        cfg = small_config(perspective_modeling_type="explicit")
        sd = O.make_state_dict(cfg, 0)
        inp = small_inputs(cfg, 0)
        inp.pop("added_time_ids")
        cams = O.make_camera_inputs(2, 3, 3, seed=0)
        inp.update(cams)
        m = build(RefDiT, VTSelfAttentionBlock, AlphaBlender, cfg, sd)
        D = cfg["num_attention_heads"] * cfg["attention_head_dim"]
        m.rayencoder = RefRayEncoder(cond_proj_dim=72, in_channels=D)
        m.rayencoder.proj.weight.data.copy_(sd["rayencoder.proj.weight"])
        res, _, _ = RefDiT.forward(m, **inp)
        print("explicit: reference forward vs oracle forward: max abs diff", float((res[0] - O.dit_forward(sd, cfg, **inp)).abs().max()))
        hh, ww = inp["sample"].shape[-2] // 2, inp["sample"].shape[-1] // 2
        K = cams["camera_intrinsics_norm"].clone()
        K[..., 0, 0] *= ww
        K[..., 1, 1] *= hh
        K[..., 0, 2] *= ww
        K[..., 1, 2] *= hh
        ro, rd = ref_get_rays(K.flatten(0, 2), cams["camera2referego"].flatten(0, 2), (hh, ww))
        out["explicit"] = dict(output=res[0].clone(), rays_o=ro.clone(), rays_d=rd.clone(),
                               raymap=m.rayencoder(ro, rd).clone(), **cams)
    torch.save(out, os.path.join(HERE, "reference_forward.pt"))
    print("wrote reference_forward.pt", {k: list(v["output"].shape) for k, v in out.items()})


if __name__ == "__main__":
    main()
