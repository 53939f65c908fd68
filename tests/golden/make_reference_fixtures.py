"""Golden vectors produced by EXECUTING reference code (runs only where /root/reference exists; the fixtures it writes are
committed and are what the tests read - nothing under tests/ touches /root/reference at run time).

`diffusers` is not installed here, so only the reference's OWN logic can be executed: the modules are imported behind an
import-only stub of the `diffusers` namespace (every attribute resolves to an empty torch.nn.Module subclass or an identity
decorator; none of it is ever called), and the functions run are the ones that contain no diffusers arithmetic:

  * dwm.models.crossview_temporal.AlphaBlender                                   (crossview_temporal.py:9-72)
  * DiTCrossviewTemporalConditionModel.forward_crossview_block_and_mix_result    (crossview_temporal_dit.py:223-327)
    and .forward_temporal_block_and_mix_result (:329-370) - the einops rearranges, the expansion of the [B, V, V] view
    mask to token resolution, the AlphaBlender mix - driven with a plain-torch attention block (2 heads x 64, identity
    projections) in place of the diffusers-built VTSelfAttentionBlock; the tensors the block RECEIVES are recorded;
  * dwm.functional.take_sequence_clip / memory_efficient_split_call              (functional.py:172-193)
  * dwm.common.create_instance_from_config / get_class                            (common.py:133-179; the JSON reflection boundary)

usage: python tests/golden/make_reference_fixtures.py   ->  tests/golden/reference_*.pt
"""
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/src"


class _Stub(types.ModuleType):
    """import-only stand-in: sub-modules on demand, classes as empty nn.Module subclasses, decorators as identity"""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        if name == "register_to_config":
            return lambda f: f
        if name[:1].isupper():
            v = type(name, (torch.nn.Module,), {})
        else:
            v = _Stub(self.__name__ + "." + name)
            sys.modules[v.__name__] = v
        setattr(self, name, v)
        return v


def install_stub():
    root = _Stub("diffusers")
    sys.modules["diffusers"] = root
    for sub in ("models", "models.attention", "models.embeddings", "models.resnet", "models.transformers",
                "models.transformers.transformer_temporal", "models.adapter", "models.attention_processor",
                "models.normalization", "models.unets", "configuration_utils", "schedulers", "utils", "image_processor"):
        cur = root
        for part in sub.split("."):
            cur = getattr(cur, part)


def sdpa_block(heads=2):
    """plain-torch self-attention over the tokens it is handed, identity projections; records its inputs"""
    seen = []

    def block(x, self_attention_mask=None):
        seen.append((x.clone(), None if self_attention_mask is None else self_attention_mask.clone()))
        Bp, L, C = x.shape
        q = x.view(Bp, L, heads, C // heads).transpose(1, 2)
        m = None if self_attention_mask is None else self_attention_mask[:, None]
        o = torch.nn.functional.scaled_dot_product_attention(q, q, q, attn_mask=m)
        return o.transpose(1, 2).reshape(Bp, L, C)
    return block, seen


def main():
    install_stub()
    sys.path.insert(0, REF)
    import dwm.common
    import dwm.functional
    from dwm.models.crossview_temporal import AlphaBlender
    from dwm.models.crossview_temporal_dit import DiTCrossviewTemporalConditionModel as RefDiT

    torch.manual_seed(0)                 # torch.nn.Linear below draws from the global generator
    g = torch.Generator().manual_seed(0)
    B, T, V, h, w, C = 2, 3, 3, 2, 3, 128
    hidden = torch.randn(B * T * V, h * w, C, generator=g)
    view_emb = torch.randn(B * T * V, 1, C, generator=g) * 0.3
    seq_emb = torch.randn(B * T * V, 1, C, generator=g) * 0.3
    mask = torch.rand(B, V, V, generator=g) > 0.35
    mask |= torch.eye(V, dtype=torch.bool)[None]
    out = {"shape": dict(B=B, T=T, V=V, h=h, w=w, C=C), "hidden": hidden, "view_emb": view_emb, "seq_emb": seq_emb, "mask": mask}

    # --- AlphaBlender
    blend = {}
    for strat in AlphaBlender.strategies:
        m = AlphaBlender(alpha=0.7, merge_strategy=strat)
        flag = torch.tensor([True, False])
        a, b = torch.randn(2, 4, 5, 6, generator=g), torch.randn(2, 4, 5, 6, generator=g)
        kw = dict(image_only_indicator=flag) if strat == "learned_with_images" else {}
        blend[strat] = dict(a=a, b=b, flag=flag, alpha=m.get_alpha(**kw).detach(), out=m(a, b, **kw).detach())
    out["alpha_blender"] = blend

    # --- cross-view / temporal rearrange + mask + mix
    mixer = AlphaBlender(alpha=2.0, merge_strategy="learned_with_images")
    cases = {}
    for ct in ("rowwise", "full"):
        for flags in ((False, False), (True, False)):
            blk, seen = sdpa_block()
            self_ = types.SimpleNamespace(crossview_attention_type=ct)
            dis = torch.tensor(flags)
            y = RefDiT.forward_crossview_block_and_mix_result(
                self_, blk, mixer, hidden, view_emb, B, T, V, w, h, dis, mask if ct == "rowwise" else None, None)
            cases[f"crossview_{ct}_{int(flags[0])}"] = dict(disable=dis, block_in=seen[0][0], block_mask=seen[0][1], out=y.detach())
    for tt in ("full", "rowwise", "pointwise"):
        blk, seen = sdpa_block()
        self_ = types.SimpleNamespace(temporal_attention_type=tt)
        dis = torch.tensor([False, True])
        y = RefDiT.forward_temporal_block_and_mix_result(self_, blk, mixer, hidden, seq_emb, B, T, V, w, dis)
        cases[f"temporal_{tt}"] = dict(disable=dis, block_in=seen[0][0], block_mask=None, out=y.detach())
    out["mix_factor"] = mixer.mix_factor.detach()
    out["blocks"] = cases

    # --- functional helpers
    t = torch.arange(2 * 7 * 3).view(2, 7, 3).float()
    out["take_sequence_clip"] = dict(
        tensor=t, clip_2_5=dwm.functional.take_sequence_clip(t, 2, 5), vec=dwm.functional.take_sequence_clip(torch.arange(4.0), 1, 3),
        scalar=dwm.functional.take_sequence_clip(2.5, 1, 3), nested=dwm.functional.take_sequence_clip([[1, 2, 3, 4], [5, 6, 7, 8]], 1, 3))
    x = torch.randn(7, 3, generator=g)
    lin = torch.nn.Linear(3, 2)
    out["split_call"] = dict(x=x, weight=lin.weight.detach(), bias=lin.bias.detach(),
                             full=dwm.functional.memory_efficient_split_call(lin, x, lambda blk, tns: blk(tns) * 2, -1).detach(),
                             split3=dwm.functional.memory_efficient_split_call(lin, x, lambda blk, tns: blk(tns) * 2, 3).detach())
    # --- JSON reflection (the drop-in boundary): nested {"_class_name": ...} configs
    cfg = {"_class_name": "torch.nn.Sequential", "_args": None}
    inst = dwm.common.create_instance_from_config({"_class_name": "torch.nn.Linear", "in_features": 5, "out_features": 3, "bias": False})
    nested = dwm.common.create_instance_from_config(
        {"_class_name": "torch.nn.ModuleDict", "modules": {"a": {"_class_name": "torch.nn.ReLU"}, "b": {"_class_name": "torch.nn.Linear", "in_features": 2, "out_features": 2}}})
    out["reflection"] = dict(linear_type=type(inst).__name__, linear_shape=tuple(inst.weight.shape), linear_bias=inst.bias is None,
                             nested_types={k: type(v).__name__ for k, v in nested.items()},
                             get_class=dwm.common.get_class("torch.nn.GELU").__name__)
    torch.save(out, os.path.join(HERE, "reference_blocks.pt"))
    print("wrote reference_blocks.pt:", {k: (list(v["out"].shape)) for k, v in cases.items()})


if __name__ == "__main__":
    main()
