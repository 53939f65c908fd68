"""Golden vectors produced by executing REFERENCE code (tests/golden/make_reference_fixtures.py; the reference's own
logic run behind an import-only `diffusers` stub): AlphaBlender, the cross-view / temporal rearrange + mask + mix methods
of DiTCrossviewTemporalConditionModel, dwm.functional helpers, dwm.common reflection.  Checked here against the oracle
restatement AND against the product's host logic (row maps, group masks, blender alphas, clip / split helpers).
These parts of the parity chain are pinned; the diffusers-internal arithmetic is not (see oracle headers)."""
import importlib
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ctsd_oracle as O          # noqa: E402
from tests.common import GOLDEN                # noqa: E402


@pytest.fixture(scope="module")
def fx():
    return torch.load(os.path.join(GOLDEN, "reference_blocks.pt"))


def sdpa_block(x, mask=None, heads=2):
    Bp, L, C = x.shape
    q = x.view(Bp, L, heads, C // heads).transpose(1, 2)
    o = torch.nn.functional.scaled_dot_product_attention(q, q, q, attn_mask=None if mask is None else mask[:, None])
    return o.transpose(1, 2).reshape(Bp, L, C)


def test_alpha_blender_oracle_and_product(fx):
    from opendwm_amd.blocks import AlphaBlender
    for strat, d in fx["alpha_blender"].items():
        m = AlphaBlender(0.7, merge_strategy=strat)
        got = m.get_alpha(d["flag"] if strat == "learned_with_images" else None, 2)
        want = d["alpha"].expand(2) if d["alpha"].numel() == 1 else d["alpha"]
        assert torch.allclose(got.float().cpu(), want.float(), atol=1e-6), strat
    d = fx["alpha_blender"]["learned_with_images"]
    out = O.alpha_blender({"m.mix_factor": torch.tensor([0.7])}, "m", d["a"], d["b"], d["flag"])
    assert torch.allclose(out, d["out"], atol=1e-6)


@pytest.mark.parametrize("name", ["crossview_rowwise_0", "crossview_rowwise_1", "crossview_full_0", "crossview_full_1",
                                  "temporal_full", "temporal_rowwise", "temporal_pointwise"])
def test_oracle_rearrange_mask_mix_equals_reference(fx, name, monkeypatch):
    s, c = fx["shape"], fx["blocks"][name]
    seen = []

    def blk(sd, p, heads, x, mask=None):
        seen.append((x, mask))
        return sdpa_block(x, mask)
    monkeypatch.setattr(O, "vt_self_attention_block", blk)
    kind, typ = name.split("_")[0], name.split("_")[1]
    cfg = {"num_attention_heads": 2, "crossview_attention_type": typ, "temporal_attention_type": typ}
    if kind == "crossview":
        sd = {"view_mixers.0.mix_factor": fx["mix_factor"]}
        out = O.crossview_block_and_mix(sd, cfg, 0, fx["hidden"], fx["view_emb"], s["B"], s["T"], s["V"], s["w"], s["h"],
                                        c["disable"], fx["mask"] if typ == "rowwise" else None)
    else:
        sd = {"time_mixers.0.mix_factor": fx["mix_factor"]}
        out = O.temporal_block_and_mix(sd, cfg, 0, fx["hidden"], fx["seq_emb"], s["B"], s["T"], s["V"], s["w"], c["disable"])
    assert torch.equal(seen[0][0], c["block_in"])
    if c["block_mask"] is not None:
        assert torch.equal(seen[0][1], c["block_mask"])
    assert torch.allclose(out, c["out"], atol=1e-6)


@pytest.mark.parametrize("name", ["crossview_rowwise_0", "crossview_full_0", "temporal_full", "temporal_rowwise", "temporal_pointwise"])
def test_product_row_maps_reproduce_reference_rearranges(fx, name):
    """the attention kernel never materialises the rearranged tensor: RowMap.rows() is the host statement of its addressing"""
    from opendwm_amd import ops
    s, c = fx["shape"], fx["blocks"][name]
    B, T, V, h, w = s["B"], s["T"], s["V"], s["h"], s["w"]
    mk = {"crossview_rowwise": ops.rowmap_crossview_rowwise, "crossview_full": ops.rowmap_crossview_full, "temporal_full": ops.rowmap_temporal_full,
          "temporal_rowwise": ops.rowmap_temporal_rowwise, "temporal_pointwise": ops.rowmap_temporal_pointwise}["_".join(name.split("_")[:2])]
    rm = mk(B, T, V, h, w)
    x = fx["hidden"] + (fx["view_emb"] if name.startswith("crossview") else fx["seq_emb"])
    tokens = x.reshape(-1, x.shape[-1])
    got = tokens[rm.rows()].view(rm.n_problems, rm.L0, -1)
    assert torch.equal(got, c["block_in"])
    if c["block_mask"] is not None:                        # group mask [B, V, V] -> the dense mask the reference builds
        p = torch.arange(rm.n_problems)[:, None, None] // rm.p_per_mask
        gq = (torch.arange(rm.L0)[None, :, None] // rm.group_size) % V
        gk = (torch.arange(rm.L0)[None, None, :] // rm.group_size) % V
        assert torch.equal(fx["mask"][p, gq, gk], c["block_mask"])


def test_functional_helpers_equal_reference(fx):
    from opendwm_amd.drivers import LatentDecoder, take_sequence_clip
    d = fx["take_sequence_clip"]
    assert torch.equal(take_sequence_clip(d["tensor"], 2, 5), d["clip_2_5"])
    assert torch.equal(take_sequence_clip(torch.arange(4.0), 1, 3), d["vec"])
    assert take_sequence_clip(2.5, 1, 3) == d["scalar"] and take_sequence_clip([[1, 2, 3, 4], [5, 6, 7, 8]], 1, 3) == d["nested"]
    from oracle import drivers_oracle as DO
    assert torch.equal(DO.take_sequence_clip(d["tensor"], 2, 5), d["clip_2_5"])
    sc = fx["split_call"]

    class FakeVae:                                         # decode = the callable the fixture's split call wrapped
        config = None

        def decode(self, x, return_dict=False):
            return ((x @ sc["weight"].T + sc["bias"]) * 2,)
    dec = LatentDecoder.__new__(LatentDecoder)
    dec.vae, dec.batch = FakeVae(), -1
    assert torch.allclose(dec._split_call(sc["x"]), sc["full"], atol=1e-6)
    dec.batch = 3
    assert torch.allclose(dec._split_call(sc["x"]), sc["split3"], atol=1e-6)


def test_json_reflection_instantiates_the_drop_in_classes(fx):
    """`{"_class_name": "<module>.<Class>", **kwargs}` (dwm/common.py:133-179): the reference resolves the class with
    importlib + getattr and calls it with the remaining (recursively instantiated) entries"""
    r = fx["reflection"]

    def create(cfg):
        if isinstance(cfg, dict) and "_class_name" in cfg:
            mod, cls = cfg["_class_name"].rsplit(".", 1)
            return getattr(importlib.import_module(mod), cls)(**{k: create(v) for k, v in cfg.items() if k != "_class_name"})
        if isinstance(cfg, dict):
            return {k: create(v) for k, v in cfg.items()}
        if isinstance(cfg, list):
            return [create(v) for v in cfg]
        return cfg
    lin = create({"_class_name": "torch.nn.Linear", "in_features": 5, "out_features": 3, "bias": False})
    assert type(lin).__name__ == r["linear_type"] and tuple(lin.weight.shape) == r["linear_shape"] and (lin.bias is None) == r["linear_bias"]
    nested = create({"_class_name": "torch.nn.ModuleDict", "modules": {"a": {"_class_name": "torch.nn.ReLU"},
                                                                        "b": {"_class_name": "torch.nn.Linear", "in_features": 2, "out_features": 2}}})
    assert {k: type(v).__name__ for k, v in nested.items()} == r["nested_types"]
    from tests.common import small_config
    with torch.device("meta"):
        m = create({"_class_name": "opendwm_amd.dit.DiTCrossviewTemporalConditionModel", **small_config()})
        v = create({"_class_name": "opendwm_amd.vae_cogvideox.AutoencoderKLCogVideoX", "block_out_channels": [64, 64, 128, 128]})
    assert type(m).__name__ == "DiTCrossviewTemporalConditionModel" and type(v).__name__ == "AutoencoderKLCogVideoX"


# ------------------------------------------------------------------------------------------------------------------
# generation control flow: fixtures from the REAL ctsd.py methods (tests/golden/make_reference_driver_fixtures.py)
@pytest.fixture(scope="module")
def dfx():
    return torch.load(os.path.join(GOLDEN, "reference_drivers.pt"))


@pytest.fixture()
def fake_model(monkeypatch):
    def fwd(sd, cfg, sample, timestep, c=None, scale=None, **kw):
        return (0.1 * sample + 1e-4 * timestep[..., None, None, None] + 0.01 * c[..., None, None, None]) * scale
    monkeypatch.setattr(O, "dit_forward", fwd)


def _cond(batch):
    return {k: v for k, v in batch.items() if k != "pts"}


@pytest.mark.parametrize("mode", ["full", "reference_frames", "diffusion_forcing", "diffusion_forcing_warmup"])
def test_oracle_denoise_loop_equals_reference_inference_pipeline(dfx, fake_model, mode):
    """oracle.denoise (and through it every GPU denoise test) against CrossviewTemporalSD.inference_pipeline itself"""
    d = dfx["inference_pipeline"][mode]
    kw = dict(d["kwargs"])
    start, stop, take = kw.pop("start_timestep", 0), kw.pop("stop_timestep", None), kw.pop("take_time", 0)
    df = mode.startswith("diffusion_forcing")
    noise = torch.randn(tuple(d["shape"]), generator=torch.Generator().manual_seed(d["seed"]))
    out = O.denoise(None, None, noise, _cond(d["batch"]), d["steps"], dfx["guidance"], stop=stop, start=start,
                    diffusion_forcing=df, take_time=take, **kw)
    assert torch.allclose(out, d["latents"], atol=1e-6)
    want_img = d["latents"][:, take].flatten(0, 1) if df else d["latents"].flatten(0, 2)
    assert torch.equal(d["images"], want_img)


@pytest.mark.parametrize("name", ["image_full", "image_split2", "image_df", "clip_full", "clip_split1", "clip_df"])
def test_latent_decoder_equals_reference_decode_tail(dfx, name):
    """drivers.LatentDecoder against the decode tail of the REAL inference_pipeline (ctsd.py:1604-1647) run with
    stand-in VAEs whose output depends on channel and frame position: scaling / shift factors, the `(b t v)` image batch
    or `(b v) c t h w` clips, memory_efficient_split_call chunks, the diffusion-forcing decode of the frame `take_time`
    (for the temporal VAE as [frame, zeros], first half kept), reference frames spliced in before decoding."""
    from opendwm_amd.drivers import LatentDecoder
    from tests.golden.make_reference_driver_fixtures import FakeClipVae, FakeImageVae
    d = dfx["decode_tail"][name]
    dec = LatentDecoder.__new__(LatentDecoder)
    dec.vae = FakeClipVae() if d["temporal"] else FakeImageVae()
    dec.batch, dec.postprocess, dec.group, dec.is_temporal_vae = d["memory_efficient_batch"], False, None, d["temporal"]
    lat = d["latents"]
    got = dec(lat[:, d["take_time"]:d["take_time"] + 1], diffusion_forcing=True) if d["df"] else dec(lat)
    assert got.shape == d["images"].shape and torch.equal(got, d["images"])


class _LoopDenoiser:
    def __init__(self, steps, g):
        self.steps, self.g, self.calls = steps, g, []

    def run(self, latents, conditions, stop=None, start=0, **kw):
        self.calls.append((start, stop, kw.get("take_time", 0)))
        return O.denoise(None, None, latents, conditions, self.steps, self.g, stop=stop, start=start, **kw)


class _RefVae:
    """FakeRefVae of make_reference_driver_fixtures.py: reference-frame encoder [N, 3, 6, 8] -> [N, 2, 3, 4], identity decode"""

    def __init__(self):
        import types
        self.config = types.SimpleNamespace(scaling_factor=0.7, shift_factor=0.1)
        self.dtype = torch.float32

    def encode(self, x):
        import types
        p = torch.nn.functional.avg_pool2d(x, 2)
        y = p[:, :2] + 0.3 * p[:, 2:3]
        return types.SimpleNamespace(latent_dist=types.SimpleNamespace(mode=lambda: y, sample=lambda: y + 0.01))

    def decode(self, x, return_dict=False):
        return (x,)


@pytest.mark.parametrize("name", ["full_ref1_given", "full_ref2_given_split", "full_ref1", "full_ref2", "df_clear0", "df_clear1"])
def test_autoregressive_drivers_equal_reference(dfx, fake_model, name):
    """`*_given`: generate_frames_for_reference = False - the reference frames are ENCODED (ctsd.py:1677-1703,
    `.latent_dist.mode()`, shift / scaling, split-call) and carried into the first window: drivers.LatentEncoder must
    produce the very latents the reference hands to its first inference_pipeline call, and drivers.LatentDecoder the
    emitted images (decode of latents / scaling + shift)."""
    from oracle import drivers_oracle as DO
    from opendwm_amd.drivers import AutoregressiveDriver, LatentDecoder, LatentEncoder
    d = dfx["autoregressive"][name]
    cfg = dict(d["config"], inference_steps=d["steps"])
    cond = {k: v for k, v in d["batch"].items() if k not in ("pts", "vae_images")}
    G = dfx["guidance"]
    il, post, decode = None, (lambda x: x), None
    if d["reference_latents"] is not None:
        vae = _RefVae()
        ref = cfg.get("reference_frame_count", 1)
        il = LatentEncoder(vae, d["memory_efficient_batch"], is_temporal_vae=False)(d["batch"]["vae_images"][:, :ref] * 2 - 1, sample=False)
        assert torch.equal(il, d["reference_latents"])
        post = lambda x: x / 0.7 + 0.1
        decode = LatentDecoder.__new__(LatentDecoder)
        decode.vae, decode.batch, decode.postprocess, decode.group, decode.is_temporal_vae = vae, d["memory_efficient_batch"], False, None, False

    def window(latent_shape, c, il_, ref, start, stop, take_time, noise):
        lat0 = noise if noise is not None else torch.zeros(tuple(latent_shape))
        lat = O.denoise(None, None, lat0, c, d["steps"], G, stop=stop, start=start, image_latents=il_, reference_frame_count=ref,
                        diffusion_forcing=d["df"], take_time=take_time, clear_reference_frame_count=cfg.get("clear_reference_frame_count", 0))
        return {"latents": lat, "images": post(lat[:, take_time].flatten(0, 1) if d["df"] else lat.flatten(0, 2))}
    want = DO.autoregressive(window, d["shape"], cond, d["total"], cfg, d["df"], torch.Generator().manual_seed(d["seed"]), image_latents=il)
    assert want["images"].shape == d["images"].shape and torch.allclose(want["images"], d["images"], atol=1e-6)
    den = _LoopDenoiser(d["steps"], G)
    got = AutoregressiveDriver(den, cfg, diffusion_forcing=d["df"], decode=decode, generator=torch.Generator().manual_seed(d["seed"])).run(
        d["shape"], cond, d["total"], "cpu", image_latents=il)
    assert torch.allclose(got["images"], d["images"], atol=1e-6)
    ref_calls = [(s, st, tt) for s, st, tt, _ in d["calls"]]
    if d["df"]:
        assert den.calls == ref_calls                    # (start, stop, take_time) of every window as the reference issued them
    else:
        assert len(den.calls) == len(ref_calls)


@pytest.mark.parametrize("name", ["fifo5", "fifo8"])
def test_streaming_drivers_equal_reference(dfx, fake_model, name):
    from oracle import drivers_oracle as DO
    from opendwm_amd.drivers import StreamingDriver
    d = dfx["streaming"][name]
    cfg = dict(d["config"], inference_steps=d["steps"])
    cfg["autoregression_data_exception_for_take_sequence"] = ["scale"]
    cond = _cond(d["batch"])
    G = dfx["guidance"]

    def window(latent_shape, conditions, latents, start, stop, take_time):
        lat = O.denoise(None, None, latents, conditions, d["steps"], G, stop=stop, start=start, image_latents=latents,
                        diffusion_forcing=True, take_time=take_time)
        return lat, (lat[:, take_time].flatten(0, 1) if stop >= d["steps"] else None)
    want = DO.Streaming(window, cfg, torch.Generator().manual_seed(d["seed"])).fifo(d["shape"], cond, d["total"])
    assert torch.allclose(want, d["images"], atol=1e-6)
    den = _LoopDenoiser(d["steps"], G)
    drv = StreamingDriver(den, cfg, generator=torch.Generator().manual_seed(d["seed"]))
    got = drv.fifo(d["shape"], cond, d["total"], "cpu")
    assert torch.allclose(got, d["images"], atol=1e-6)
    assert torch.allclose(drv.latents, d["final_latents"], atol=1e-6)


# ------------------------------------------------------------------------------------------------------------------
# model forward composition: fixture from the REAL DiTCrossviewTemporalConditionModel.forward / VTSelfAttentionBlock.forward
# with oracle leaves (tests/golden/make_reference_forward_fixture.py)
@pytest.mark.parametrize("tt", ["rowwise", "pointwise", "full"])
def test_oracle_forward_composition_equals_reference_forward(tt):
    from tests.common import small_config, small_inputs
    fxf = torch.load(os.path.join(GOLDEN, "reference_forward.pt"))
    cfg = small_config(temporal_attention_type=tt)
    sd = O.make_state_dict(small_config(), 0)
    inp = small_inputs(cfg, 0)
    inp["disable_temporal"] = fxf[tt]["disable_temporal"]
    out = O.dit_forward(sd, cfg, **inp)
    assert torch.allclose(out, fxf[tt]["output"], atol=1e-6)


def test_oracle_explicit_perspective_equals_reference():
    """explicit perspective modelling (crossview_temporal_dit.py:11-102, 440-458): oracle.get_rays / ray_encoder and the whole
    oracle forward against the REAL get_rays, RayEncoder.forward and model forward (pure-torch reference code, executed)"""
    from tests.common import small_config, small_inputs
    fx = torch.load(os.path.join(GOLDEN, "reference_forward.pt"))["explicit"]
    cfg = small_config(perspective_modeling_type="explicit")
    sd = O.make_state_dict(cfg, 0)
    assert "rayencoder.proj.weight" in sd and "view_embedding.linear_1.weight" not in sd
    inp = small_inputs(cfg, 0)
    inp.pop("added_time_ids")
    cams = O.make_camera_inputs(2, 3, 3, seed=0)
    assert torch.equal(cams["camera2referego"], fx["camera2referego"])
    hh, ww = inp["sample"].shape[-2] // 2, inp["sample"].shape[-1] // 2
    K = cams["camera_intrinsics_norm"].clone()
    K[..., 0, 0] *= ww
    K[..., 1, 1] *= hh
    K[..., 0, 2] *= ww
    K[..., 1, 2] *= hh
    ro, rd = O.get_rays(K.flatten(0, 2), cams["camera2referego"].flatten(0, 2), (hh, ww))
    assert torch.allclose(ro, fx["rays_o"], atol=1e-6) and torch.allclose(rd, fx["rays_d"], atol=1e-6)
    assert torch.allclose(O.ray_encoder(sd, ro, rd), fx["raymap"], atol=1e-5)
    out = O.dit_forward(sd, cfg, **inp, **cams)
    assert (out - fx["output"]).abs().max().item() < 1e-5
    # the product's host half: the 21 camera scalars per image handed to dwm_ray_features reproduce the rays
    from opendwm_amd.dit import DiTCrossviewTemporalConditionModel, RayEncoder
    rows = RayEncoder.camera_rows(cams["camera_intrinsics_norm"], cams["camera2referego"], hh, ww)
    I = rows.shape[0]
    ys, xs = torch.meshgrid(torch.arange(hh).float() + 0.5, torch.arange(ww).float() + 0.5, indexing="ij")
    pts = torch.stack([xs, ys, torch.ones_like(xs)], -1).view(1, hh * ww, 3, 1)
    d = (rows[:, 9:18].view(I, 1, 3, 3) @ (rows[:, :9].view(I, 1, 3, 3) @ pts)).squeeze(-1)
    d = d / d.norm(dim=-1, keepdim=True)
    assert torch.allclose(d.view(I, hh, ww, 3), fx["rays_d"], atol=1e-5) and torch.allclose(rows[:, 18:], fx["rays_o"], atol=1e-6)
    m = DiTCrossviewTemporalConditionModel(**cfg)
    assert set(m.state_dict().keys()) == set(sd.keys())


def test_reference_keeps_the_view_axis_for_5d_inputs():
    """`result = [output]` is built before the squeeze (crossview_temporal_dit.py:620-630): 5-D inputs come back 6-D in the
    tuple form; the product mirrors that (opendwm_amd/dit.py) - checked on the GPU in test_hip_gpu.py"""
    fxf = torch.load(os.path.join(GOLDEN, "reference_forward.pt"))
    assert fxf["five_dim"]["output"].dim() == 6 and fxf["five_dim"]["output"].shape[2] == 1


@pytest.mark.parametrize("name", ["rowwise", "pointwise", "layout"])
def test_oracle_unet_composition_equals_reference_forward(name):
    """tests/golden/reference_unet_forward.pt: the REAL UNetCrossviewTemporalConditionModel.forward with the real down / mid /
    up block, ResBlock, TransformerModel and TemporalBasicTransformerBlock classes over oracle leaf modules
    (make_reference_unet_fixture.py)"""
    from oracle import unet_oracle as U
    from tests.golden.make_golden import unet_small_config
    fxu = torch.load(os.path.join(GOLDEN, "reference_unet_forward.pt"))[name]
    cfg = dict(unet_small_config(), **fxu["over"])
    sd = U.make_unet_state_dict(cfg, 0)
    inp = U.make_unet_inputs(cfg, 2, 2, 3, 8, 16, text_len=10)
    inp["disable_crossview"], inp["disable_temporal"] = fxu["flags"]
    if name == "pointwise":
        inp["crossview_attention_mask"] = None
    if name == "layout":              # the REAL ImageAdapter.forward feeding the REAL forward's residual insertion (:719-729, :753-754)
        inp["condition_image_tensor"] = torch.rand(2, 2, 3, 3, 64, 128, generator=torch.Generator().manual_seed(5))
    out = U.unet_forward(sd, cfg, **inp)
    assert torch.allclose(out, fxu["output"], atol=5e-5)


# ------------------------------------------------------------------------------------------------------------------
# training step: fixture from the REAL CrossviewTemporalSD.train_step (tests/golden/make_reference_train_fixture.py)
class _EncVae:
    """the stand-in VAEs of make_reference_train_fixture.py (2-D: average pooling, shift 0.1, scale 1.5; temporal:
    "(b v) c t h w" clips with a per-frame offset, no shift, scale 0.8)"""

    def __init__(self, temporal):
        import types
        self.temporal = temporal
        self.config = types.SimpleNamespace(shift_factor=None if temporal else 0.1, scaling_factor=0.8 if temporal else 1.5)

    def encode(self, x):
        import types
        if self.temporal:
            y = torch.nn.functional.avg_pool3d(x, (1, 8, 8)) + torch.arange(x.shape[2], dtype=x.dtype).view(1, 1, -1, 1, 1) * 0.05
        else:
            y = torch.nn.functional.avg_pool2d(x, 8)
        return types.SimpleNamespace(latent_dist=types.SimpleNamespace(sample=lambda: y, mode=lambda: y))


@pytest.mark.parametrize("name", ["plain", "loss_coef", "temporal_vae"])
def test_training_pair_and_loss_equal_reference_train_step(name, monkeypatch):
    """the product's encode call site (drivers.LatentEncoder: image batch / "(b v) c t h w" clips, split-call chunks,
    shift / scaling), flow-matching pair and the oracle's loss / gradient against the REAL train_step"""
    from opendwm_amd.drivers import LatentEncoder
    from opendwm_amd.pipeline import CTSDTrainer, flow_match_train_sigmas, sample_timestep_indices
    d = torch.load(os.path.join(GOLDEN, "reference_train_step.pt"))[name]
    img = d["batch"]["vae_images"]
    B, T, V = img.shape[:3]
    lat = LatentEncoder(_EncVae(d["temporal_vae"]), d["memory_efficient_batch"], is_temporal_vae=d["temporal_vae"])(img * 2 - 1, sample=True)
    assert lat.shape[:3] == (B, T, V)
    noise = torch.randn(lat.shape, generator=torch.Generator().manual_seed(d["generator_seed"]))
    torch.manual_seed(d["global_seed"])
    idx = sample_timestep_indices((B,))                                # global generator, like the reference's torch.normal
    sig = flow_match_train_sigmas()
    assert torch.allclose(sig[idx] * 1000, d["timesteps"][:, 0, 0], atol=1e-4)
    # product: the flow-matching pair
    tr = CTSDTrainer.__new__(CTSDTrainer)
    tr.sigmas, tr.num_train_timesteps, tr.weighting_scheme = sig, 1000, "logit_normal"
    noisy, ts, sg, _ = tr.make_training_pair(lat, timestep_indices=idx, noise=noise)
    assert torch.allclose(noisy, d["noisy_latents"], atol=1e-6) and torch.allclose(ts, d["timesteps"], atol=1e-4)
    # oracle: loss and its gradient w.r.t. the stand-in model's scale
    w = torch.tensor(0.3, requires_grad=True)

    def fwd(sd, cfg, sample, timestep, c=None, **kw):
        return w * (sample + 1e-3 * timestep[..., None, None, None] + 0.05 * c[..., None, None, None])
    monkeypatch.setattr(O, "dit_forward", fwd)
    coef = d["training_config"].get("loss_coef_dict", {}).get("sd", 1.0)
    loss = O.train_loss(None, None, lat, {"c": d["batch"]["c"]}, idx, noise, loss_coef=coef)
    assert abs(loss.item() - d["loss"].item()) < 1e-6
    loss.backward()
    g = w.grad
    clip = d["training_config"].get("max_norm_for_grad_clip")
    if clip is not None:
        g = g * min(1.0, clip / (g.abs().item() + 1e-6))
    assert abs((0.3 - d["lr"] * g).item() - d["w_after"].item()) < 1e-6


def test_oracle_layout_branch_equals_reference_forward():
    """the REAL ImageAdapter.forward (adapters.py:40-60) feeding the REAL model forward's residual insertion"""
    from tests.common import small_config, small_inputs
    fxf = torch.load(os.path.join(GOLDEN, "reference_forward.pt"))["layout"]
    cfg = small_config(condition_image_adapter_config=fxf["adapter_config"])
    sd = O.make_state_dict(cfg, 0)
    inp = small_inputs(cfg, 0)
    inp["condition_image_tensor"] = torch.rand(2, 3, 3, 6, 64, 96, generator=torch.Generator().manual_seed(5))
    assert torch.allclose(O.dit_forward(sd, cfg, **inp), fxf["output"], atol=1e-6)


# ------------------------------------------------------------------------------------------------------------------
# condition builder: fixtures from the REAL CrossviewTemporalSD.get_conditions (tests/golden/make_reference_condition_fixtures.py)
CONDITION_CASES = ["layout_cfg", "layout_nocfg", "text_only_first_frame_images", "masks", "action_mask_off", "streaming_first",
                   "streaming_next", "explicit_view", "explicit_view_no_ego", "temporal_vae_5_to_2", "temporal_vae_4_to_2"]


@pytest.mark.parametrize("name", CONDITION_CASES)
def test_build_conditions_equals_reference_get_conditions(name):
    """opendwm_amd.conditions.build_conditions against get_conditions itself (ctsd.py:159-453, text branch excluded): layout
    images with the unconditional colour and the CFG doubling, fps / camera / action ids (13 added time ids of the layout
    checkpoints, 11 of the text-only ones; -1000 action ids for the unconditional half, for masked or motionless samples),
    streaming mode with the previous ego pose, explicit-view camera matrices, flags, masks, temporal-VAE frame striding.
    The index lists are the ones of the reference's example JSONs."""
    from opendwm_amd.conditions import build_conditions
    d = torch.load(os.path.join(GOLDEN, "reference_conditions.pt"))[name]
    batch = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in d["batch"].items()}
    got = build_conditions(d["common_config"], d["latent_shape"], batch, torch.device("cpu"), torch.float32, **d["kwargs"])
    want = d["result"]
    assert set(got) == set(want)
    for k, w in want.items():
        g = got[k]
        if w is None:
            assert g is None, k
            continue
        assert g is not None and g.shape == w.shape and g.dtype == w.dtype, (k, None if g is None else (g.shape, g.dtype), w.shape, w.dtype)
        assert torch.equal(g, w) if not w.is_floating_point() else torch.allclose(g, w, rtol=1e-6, atol=1e-6), k
    for k, v in d["batch"].items():                      # the caller's batch is left untouched
        assert not torch.is_tensor(v) or torch.equal(batch[k], v), k


def test_build_conditions_passes_text_embeddings_through():
    from opendwm_amd.conditions import build_conditions
    d = torch.load(os.path.join(GOLDEN, "reference_conditions.pt"))["temporal_vae_5_to_2"]
    B2, T = 4, 5
    ehs, pooled = torch.randn(B2, T, 6, 3, 8), torch.randn(B2, T, 6, 4)
    got = build_conditions(d["common_config"], d["latent_shape"], d["batch"], "cpu", torch.bfloat16, encoder_hidden_states=ehs,
                           pooled_projections=pooled, **d["kwargs"])
    # strided like every other per-frame condition (frames 0 and 1 of 5 for 2 latent frames), cast to the model dtype
    assert got["encoder_hidden_states"].shape == (B2, 2, 6, 3, 8) and got["encoder_hidden_states"].dtype == torch.bfloat16
    assert torch.equal(got["pooled_projections"], torch.cat([pooled[:, :1], pooled[:, 1::4]], 1).to(torch.bfloat16))


# ------------------------------------------------------------------------------------------------------------------
# training task mixer: the REAL try_make_input_for_prediction (ctsd.py:619-741) and train_step in its diffusion-forcing /
# "ctsd" styles (tests/golden/make_reference_train_fixture.py)
@pytest.mark.parametrize("name", ["none_with_augment_draws", "diffusion_forcing", "ctsd_int", "ctsd_dict"])
def test_task_mixer_equals_reference(name):
    from opendwm_amd.pipeline import make_input_for_prediction
    d = torch.load(os.path.join(GOLDEN, "reference_train_step.pt"))["task_mixer"][name]
    made, ts, extra, ind = make_input_for_prediction(d["noisy"].clone(), d["latents"].clone(), d["timesteps"].clone(), d["training_config"],
                                                     d["common_config"], torch.Generator().manual_seed(d["seed"]), d["reference_latent_count"])
    assert torch.equal(made, d["made_noisy"]) and torch.equal(ts, d["made_timesteps"]) and torch.equal(ind, d["indicator"])
    if d["additional"] is None:
        assert extra is None
    else:
        assert set(extra) == set(d["additional"]) and all(torch.equal(extra[k], v) for k, v in d["additional"].items())


@pytest.mark.parametrize("name", ["df_style", "ctsd_style"])
def test_trainer_task_styles_equal_reference_train_step(name):
    """CTSDTrainer's pieces in the reference's draw order - noise, timestep indices (per frame in the diffusion-forcing style),
    condition dropout masks, task mixer - give the very tensors the REAL train_step hands to its model (input, timesteps,
    disable_temporal); loss (with the reference-frame loss mask) and the SGD update follow in closed form for the stand-in model."""
    from opendwm_amd.drivers import LatentEncoder
    from opendwm_amd.pipeline import CTSDTrainer, flow_match_train_sigmas, make_input_for_prediction, sample_timestep_indices
    d = torch.load(os.path.join(GOLDEN, "reference_train_step.pt"))[name]
    img = d["batch"]["vae_images"]
    B, T, V = img.shape[:3]
    lat = LatentEncoder(_EncVae(False), -1, is_temporal_vae=False)(img * 2 - 1, sample=True)
    tr = CTSDTrainer.__new__(CTSDTrainer)
    tr.sigmas, tr.num_train_timesteps, tr.weighting_scheme = flow_match_train_sigmas(), 1000, "logit_normal"
    tr.common_config, tr.training_config = d["common_config"], d["training_config"]
    gen = torch.Generator().manual_seed(d["generator_seed"])
    noise = torch.randn(lat.shape, generator=gen)
    torch.manual_seed(d["global_seed"])
    per_frame = d["common_config"].get("frame_prediction_style") == "diffusion_forcing"
    idx = sample_timestep_indices((B, T) if per_frame else (B,))
    noisy, ts, sig, _ = tr.make_training_pair(lat, timestep_indices=idx, noise=noise)
    tr.draw_condition_masks(B, gen)                                        # consumed in the reference's order
    rlc = d["training_config"].get("reference_frame_count", 0)
    made, mts, extra, ind = make_input_for_prediction(noisy, lat, ts, d["training_config"], d["common_config"], gen, rlc)
    assert torch.allclose(made, d["noisy_latents"], atol=1e-6) and torch.allclose(mts, d["timesteps"], atol=1e-4)
    assert torch.equal(extra["disable_temporal"], d["seen_kwargs"]["disable_temporal"])
    w = torch.tensor(0.3, requires_grad=True)
    pred = w * (made + 1e-3 * mts[..., None, None, None] + 0.05 * d["batch"]["c"][..., None, None, None])
    x0, target = pred * (-sig) + made, lat
    if d["training_config"].get("disable_reference_frame_loss", False):
        keep = ~ind.view(B, T, V, 1, 1, 1)
        x0, target = x0 * keep, target * keep
    loss = torch.nn.functional.mse_loss(x0, target)
    assert abs(loss.item() - d["loss"].item()) < 1e-6
    loss.backward()
    assert abs((0.3 - d["lr"] * w.grad).item() - d["w_after"].item()) < 1e-6


def test_streaming_frame_ingestion_equals_reference(monkeypatch):
    """StreamingDriver.send_frame - dataset frames in, generated frames out - against the REAL fifo_inference_pipeline /
    send_frame_condition / get_conditions (text branch included) with stand-in text encoders and denoiser: prompts flattened
    with the CFG "" half first, the two CLIP embeddings padded and stacked with T5 (conditions.assemble_sd3_text), text
    re-embedded every 3rd frame only and the newest queued embedding reused in between, action ids against the previous
    frame's ego pose, layout images with the unconditional colour, every condition queued / slid / replaced as configured."""
    from opendwm_amd.conditions import assemble_sd3_text
    from opendwm_amd.drivers import StreamingDriver, take_sequence_clip
    from tests.golden.make_reference_condition_fixtures import stream_model, text_embedding
    d = torch.load(os.path.join(GOLDEN, "reference_conditions.pt"))["streaming_ingest"]
    monkeypatch.setattr(O, "dit_forward", lambda sd, cfg, sample, timestep, **kw: stream_model(sample.float(), timestep.float(), **kw))
    embedded = []

    def embed_text(flat, shape, view_count):
        embedded.append(list(flat))
        clip = [torch.stack([text_embedding(p, 3, dim, seed) for p in flat]) for dim, seed in ((4, 1), (5, 2))]
        pooled = [torch.stack([text_embedding(p, 1, dim, seed + 7)[0] for p in flat]) for dim, seed in ((4, 1), (5, 2))]
        t5 = torch.stack([text_embedding(p, 4, 12, 3) for p in flat])
        return assemble_sd3_text(clip, pooled, t5, shape, 1, view_count, torch.float32)

    icfg = d["inference_config"]
    den = _LoopDenoiser(d["steps"], icfg["guidance_scale"])
    drv = StreamingDriver(den, icfg, generator=torch.Generator().manual_seed(d["seed"]))
    drv.reset_streaming(d["shape"], "cpu")
    exc, out = icfg["autoregression_data_exception_for_take_sequence"], []
    for i in range(d["total"]):
        frame = {k: (v if k in exc else take_sequence_clip(v, i, i + 1)) for k, v in d["batch"].items()}
        drv.send_frame(frame, d["common_config"], embed_text, dtype=torch.float32)
        f = drv.receive_frame()
        if f is not None:
            out.append(f)
    drv.send_frame(None, d["common_config"])
    while (f := drv.receive_frame()) is not None:
        out.append(f)
    got = torch.cat(out)
    assert got.shape == d["images"].shape and torch.allclose(got, d["images"], atol=1e-6)
    assert torch.allclose(drv.latents, d["final_latents"], atol=1e-6)
    assert len(embedded) == 3 and embedded[0][:6] == [""] * 6 and embedded[1][6] == "frame 3 view 0"     # frames 0, 3, 6
    for k, w in d["final_conditions"].items():
        g = drv.conditions[k]
        assert (g is None and w is None) or (g.shape == w.shape and torch.allclose(g.float(), w.float(), atol=1e-6)), k


def test_text_flattening_and_assembly_equal_reference():
    """conditions.flatten_clip_text / assemble_sd3_text against flatten_clip_text (ctsd.py:39-82) and the text branch of
    get_conditions (:205-253) run with stand-in encoders: shared prompts repeated over frames and views, per-view prompts
    unflattened, masked prompts blanked, CFG halves in the reference's order."""
    from opendwm_amd.conditions import assemble_sd3_text, flatten_clip_text
    from tests.golden.make_reference_condition_fixtures import text_embedding
    fx = torch.load(os.path.join(GOLDEN, "reference_conditions.pt"))
    for name, d in fx["flatten_clip_text"].items():
        flat, shape = flatten_clip_text(d["text"], d["mask"], d["cfg"])
        assert flat == d["flat"] and shape == d["shape"], name
    for name, d in fx["text_branch"].items():
        flat, shape = flatten_clip_text(d["text"], d["mask"], True)
        clip = [torch.stack([text_embedding(p, 3, dim, seed) for p in flat]) for dim, seed in ((4, 1), (5, 2))]
        pooled = [torch.stack([text_embedding(p, 1, dim, seed + 7)[0] for p in flat]) for dim, seed in ((4, 1), (5, 2))]
        t5 = torch.stack([text_embedding(p, 4, 12, 3) for p in flat])
        ehs, pp = assemble_sd3_text(clip, pooled, t5, shape, 3, 6, torch.float32)
        assert torch.equal(ehs, d["encoder_hidden_states"]) and torch.equal(pp, d["pooled_projections"]), name
    from opendwm_amd.conditions import assemble_clip_text
    for name, d in fx["text_branch_sd21"].items():                      # SD 2.1 UNet: one CLIP encoder (:186-203)
        flat, shape = flatten_clip_text(d["text"], None, True)
        ehs = assemble_clip_text(torch.stack([text_embedding(p, 5, 8, 11) for p in flat]), shape, 3, 6)
        assert torch.equal(ehs, d["encoder_hidden_states"]), name


def test_autoregressive_windows_from_a_batch_equal_reference(monkeypatch):
    """AutoregressiveDriver over drivers.conditions_from_batch (the conditions of every window built from that window's clip
    of the dataset batch: prompts, layout images, camera / action ids that restart at the window's first frame) against the
    REAL autoregressive_inference_pipeline over the REAL inference_pipeline and get_conditions with stand-in encoders."""
    from opendwm_amd.conditions import assemble_sd3_text
    from opendwm_amd.drivers import AutoregressiveDriver, conditions_from_batch
    from tests.golden.make_reference_condition_fixtures import stream_model, text_embedding
    d = torch.load(os.path.join(GOLDEN, "reference_conditions.pt"))["autoregressive_batch"]
    monkeypatch.setattr(O, "dit_forward", lambda sd, cfg, sample, timestep, **kw: stream_model(sample.float(), timestep.float(), **kw))

    def embed_text(flat, shape, frames, view_count):
        clip = [torch.stack([text_embedding(p, 3, dim, seed) for p in flat]) for dim, seed in ((4, 1), (5, 2))]
        pooled = [torch.stack([text_embedding(p, 1, dim, seed + 7)[0] for p in flat]) for dim, seed in ((4, 1), (5, 2))]
        return assemble_sd3_text(clip, pooled, torch.stack([text_embedding(p, 4, 12, 3) for p in flat]), shape, frames, view_count, torch.float32)

    icfg = d["inference_config"]
    cond = conditions_from_batch(d["batch"], d["common_config"], icfg, d["shape"], "cpu", torch.float32, embed_text)
    den = _LoopDenoiser(d["steps"], icfg["guidance_scale"])
    got = AutoregressiveDriver(den, icfg, generator=torch.Generator().manual_seed(d["seed"])).run(d["shape"], cond, d["total"], "cpu")
    assert got["images"].shape == d["images"].shape and torch.allclose(got["images"], d["images"], atol=1e-6)


@pytest.mark.parametrize("name", ["plain", "loss_coef", "temporal_vae", "df_style", "ctsd_style"])
def test_trainer_loss_end_to_end_equals_reference_train_step(name):
    """CTSDTrainer.draw_training_inputs + CTSDTrainer.loss as a whole (the stand-in model of the fixture as `wrapper`, on the
    CPU): same random streams as the REAL train_step, same loss - to the bf16 rounding of the model input (1e-2 relative)."""
    from opendwm_amd.drivers import LatentEncoder
    from opendwm_amd.pipeline import CTSDTrainer, flow_match_train_sigmas
    d = torch.load(os.path.join(GOLDEN, "reference_train_step.pt"))[name]
    img = d["batch"]["vae_images"]
    lat = LatentEncoder(_EncVae(d["temporal_vae"]), d["memory_efficient_batch"], is_temporal_vae=d["temporal_vae"])(img * 2 - 1, sample=True)
    w = torch.tensor(0.3)

    def wrapper(x, ts, c=None, **kw):
        return [w * (x.float() + 1e-3 * ts[..., None, None, None] + 0.05 * c[..., None, None, None])], None, None
    tr = CTSDTrainer.__new__(CTSDTrainer)
    tr.wrapper, tr.sigmas, tr.num_train_timesteps, tr.weighting_scheme = wrapper, flow_match_train_sigmas(), 1000, "logit_normal"
    tr.common_config, tr.training_config = d.get("common_config", {}), d["training_config"]
    tr.reference_latent_count = d["training_config"].get("reference_frame_count", 0)
    tr.loss_coef = d["training_config"].get("loss_coef_dict", {}).get("sd", 1.0)
    gen = torch.Generator().manual_seed(d["generator_seed"])
    torch.manual_seed(d["global_seed"])
    noise, idx, masks = tr.draw_training_inputs(lat.shape, gen)
    assert set(masks) >= {"text_condition_mask", "_3dbox_condition_mask", "hdmap_condition_mask", "action_condition_mask"}
    loss = tr.loss(lat, {"c": d["batch"]["c"]}, generator=gen, timestep_indices=idx, noise=noise)
    assert abs(loss.item() - d["loss"].item()) / d["loss"].item() < 1e-2, (loss.item(), d["loss"].item())


@pytest.mark.parametrize("case", ["reference_checkpoint", "reference_checkpoint_frozen"])
def test_checkpoint_layout_resumes_from_reference_files(tmp_path, case):
    """CTSDTrainer.load_checkpoint / save_checkpoint against files written by the REAL save_checkpoint +
    distributed_save_optimizer_state (tests/golden/make_reference_checkpoint_fixture.py): <output>/checkpoints/<step>.pth and
    <output>/optimizer/<step>.pth in torch.optim.AdamW's format.  The loaded moments / step count / hyper-parameters
    reproduce the reference's next AdamW step, and what we write back is a file torch.optim.AdamW resumes from.
    "_frozen": a training_config["freezing_pattern"] froze the first layer before the optimizer was built from ALL
    parameters (ctsd.py:1014-1022, 1089-1092) - the state is sparse over the full parameter list."""
    from opendwm_amd.pipeline import CTSDTrainer, freeze_modules
    from opendwm_amd.train import AdamW
    from tests.golden.make_reference_checkpoint_fixture import FREEZING_PATTERN, tiny_model
    root = os.path.join(GOLDEN, case)
    exp = torch.load(os.path.join(root, "expected.pt"))
    frozen = case.endswith("frozen")
    tr = CTSDTrainer.__new__(CTSDTrainer)
    tr.model = tiny_model()
    if frozen:
        assert freeze_modules(tr.model, FREEZING_PATTERN) == ["0"]
        assert [q.requires_grad for q in tr.model.parameters()] == [False, False, True, True]
    with torch.no_grad():
        for q in tr.model.parameters():
            q.zero_()                                           # everything must come from the files
    tr.optimizer = AdamW(tr.model.parameters(), lr=123.0)
    tr.load_checkpoint(root, 3)
    opt = tr.optimizer
    assert (opt.lr, opt.betas, opt.eps, opt.weight_decay, opt.t) == (1e-2, (0.9, 0.95), 1e-8, 0.05, 3)
    assert len(opt.state) == (2 if frozen else 4)
    # one AdamW step by the textbook formula from OUR loaded state == the reference's 4th step
    for q, g, want in zip(tr.model.parameters(), exp["grads_step_4"], exp["params_after_step_4"]):
        if g is None:
            assert q not in opt.state and torch.equal(q.detach(), want)     # frozen: untouched, no state
            continue
        st = opt.state[q]
        t = int(st["step"]) + 1
        m = opt.betas[0] * st["exp_avg"] + (1 - opt.betas[0]) * g
        v = opt.betas[1] * st["exp_avg_sq"] + (1 - opt.betas[1]) * g * g
        upd = (m / (1 - opt.betas[0] ** t)) / ((v / (1 - opt.betas[1] ** t)).sqrt() + opt.eps)
        got = q.detach() * (1 - opt.lr * opt.weight_decay) - opt.lr * upd
        assert torch.allclose(got, want, atol=1e-6)
    # write it back in the same layout: torch.optim.AdamW (= the reference's resume path) takes it and makes the same step
    tr.save_checkpoint(str(tmp_path), 3)
    ref_model = tiny_model()
    if frozen:
        freeze_modules(ref_model, FREEZING_PATTERN)
    ref_model.load_state_dict(torch.load(tmp_path / "checkpoints" / "3.pth", map_location="cpu", weights_only=True))
    ref_opt = torch.optim.AdamW(ref_model.parameters())
    ours, theirs = torch.load(tmp_path / "optimizer" / "3.pth", map_location="cpu", weights_only=True), \
        torch.load(os.path.join(root, "optimizer", "3.pth"), map_location="cpu", weights_only=True)
    assert sorted(ours["state"]) == sorted(theirs["state"]) and ours["param_groups"][0]["params"] == theirs["param_groups"][0]["params"]
    ref_opt.load_state_dict(ours)
    for q, g in zip(ref_model.parameters(), exp["grads_step_4"]):
        q.grad = None if g is None else g.clone()
    ref_opt.step()
    assert all(torch.allclose(q, w, atol=1e-6) for q, w in zip(ref_model.parameters(), exp["params_after_step_4"]))


def test_adamw_is_a_torch_optimizer_with_per_parameter_steps_and_lr_schedulers():
    """host-side contract of train.AdamW (no kernel call: step() needs the GPU): torch LR schedulers attach
    (ctsd.py:1098-1100), a parameter without gradient keeps no state, the state dict is torch's"""
    from opendwm_amd.train import AdamW
    a, b = torch.nn.Parameter(torch.zeros(4)), torch.nn.Parameter(torch.zeros(4))
    opt = AdamW([a, b], lr=1.0)
    assert isinstance(opt, torch.optim.Optimizer)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: 0.5 ** s)
    sched.step()
    assert opt.lr == 0.5 and opt.param_groups[0]["initial_lr"] == 1.0
    assert opt.state_dict()["state"] == {} and opt.t == 0
    opt.zero_grad()
    assert a.grad is None




# ---------------------------------------------------------------- tensor-timestep schedulers (temporal_independent.py)
def test_scheduler_oracle_and_coefficient_tables_equal_reference():
    """oracle/scheduler_oracle.py and the host half of opendwm_amd.schedulers (coefficient rows handed to the HIP kernels)
    against the REAL DDPMScheduler.add_noise / get_velocity and DDIMScheduler.step / _get_variance
    (tests/golden/make_reference_scheduler_fixture.py)."""
    from oracle import scheduler_oracle as SO
    from opendwm_amd.schedulers import DDIMScheduler, DDPMScheduler
    fx = torch.load(os.path.join(GOLDEN, "reference_schedulers.pt"))
    acp = fx["alphas_cumprod"]
    assert torch.equal(DDPMScheduler().alphas_cumprod, acp)                 # the restated scaled_linear table
    d = fx["ddpm"]
    for name, c in d["cases"].items():
        assert torch.allclose(SO.add_noise(acp, d["x0"], d["noise"], c["timesteps"]), c["noisy"], atol=1e-6), name
        assert torch.allclose(SO.get_velocity(acp, d["x0"], d["noise"], c["timesteps"]), c["velocity"], atol=1e-6), name
    dd = fx["ddim"]
    for name, c in dd["cases"].items():
        kw = c["kw"]
        prev, x0 = SO.ddim_step(acp, c["final_alpha_cumprod"], 1000, c["num_inference_steps"], kw["prediction_type"], dd["model_output"],
                                c["timesteps"], dd["sample"], eta=kw.get("eta", 0.0), use_clipped_model_output=kw.get("use_clipped", False),
                                variance_noise=dd["variance_noise"], clip_sample=kw.get("clip_sample", False),
                                clip_sample_range=kw.get("clip_sample_range", 1.0))
        assert torch.allclose(prev, c["prev_sample"], atol=2e-5) and torch.allclose(x0, c["pred_original_sample"], atol=2e-5), name
        # the product's coefficient rows: the kernel formula evaluated in torch must give the reference's step
        sch = DDIMScheduler(prediction_type=kw["prediction_type"], clip_sample=kw.get("clip_sample", False),
                            clip_sample_range=kw.get("clip_sample_range", 1.0), set_alpha_to_one=kw.get("set_alpha_to_one", False))
        sch.set_timesteps(c["num_inference_steps"])
        assert torch.allclose(sch._get_variance(c["timesteps"], c["timesteps"] - 20), c["variance"], atol=1e-7), name
        co = sch.coefficients(c["timesteps"], kw.get("eta", 0.0)).view(*c["timesteps"].shape, 1, 1, 1, 6)
        sa, sb, sap, dr, sd = (co[..., i] for i in range(5))
        s, m = dd["sample"], dd["model_output"]
        if kw["prediction_type"] == "epsilon":
            x0k, eps = (s - sb * m) / sa, m
        elif kw["prediction_type"] == "sample":
            x0k, eps = m, (s - sa * m) / sb
        else:
            x0k, eps = sa * s - sb * m, sa * m + sb * s
        if kw.get("clip_sample"):
            x0k = x0k.clamp(-kw["clip_sample_range"], kw["clip_sample_range"])
        if kw.get("use_clipped"):
            eps = (s - sa * x0k) / sb
        got = sap * x0k + dr * eps + sd * dd["variance_noise"]
        assert torch.allclose(got, c["prev_sample"], atol=2e-5), name
    sch = DDIMScheduler()
    sch.set_timesteps(50)
    assert sch.timesteps[0].item() == 981 and sch.timesteps[-1].item() == 1 and len(sch.timesteps) == 50      # leading + offset 1
    with pytest.raises(ValueError):
        DDIMScheduler().coefficients(torch.tensor([1]))                     # set_timesteps not called: the reference raises too


# ------------------------------------------------------------------------------------------------------------------
# SD 2.1 (UNet) branch of the training step: the REAL train_step with a UNet-typed model and the REAL tensor-timestep
# DDPMScheduler methods (tests/golden/make_reference_train_unet_fixture.py)
@pytest.mark.parametrize("name", ["v_prediction", "epsilon", "df_style", "ctsd_style"])
def test_trainer_unet_branch_equals_reference_train_step(name):
    """CTSDTrainer.draw_training_inputs + CTSDTrainer.loss in the SD 2.1 branch (ctsd.py:1240-1253, 1273-1276, 1358-1360): integer
    timesteps drawn from the pipeline generator right after the noise, DDPM add_noise, epsilon / v_prediction target, the
    (b, t, v) timestep expansion, the task mixer on the same generator, mse on the raw prediction.  The stand-in model of the
    fixture is the `wrapper`; the train scheduler is the oracle's restatement of the tensor-timestep DDPMScheduler (itself
    pinned by reference_schedulers.pt; the HIP form is checked against the same vectors in the GPU leg) - same model input,
    timesteps, loss and SGD update as the REAL train_step."""
    import types
    from oracle import scheduler_oracle as SO
    from opendwm_amd.pipeline import CTSDTrainer
    fx = torch.load(os.path.join(GOLDEN, "reference_train_step_unet.pt"))
    d, acp = fx[name], fx["alphas_cumprod"]
    img = d["batch"]["vae_images"]
    B, T, V = img.shape[:3]
    lat = (torch.nn.functional.avg_pool2d((img * 2 - 1).flatten(0, 2), 8) * 0.18215).unflatten(0, (B, T, V))
    w = torch.tensor(0.3, requires_grad=True)
    seen = []

    def wrapper(x, ts, c=None, **kw):
        seen.append((x.detach().float(), ts.detach()))
        return [w * (x.float() + 1e-3 * ts[..., None, None, None] + 0.05 * c[..., None, None, None])], None, None
    tr = CTSDTrainer.__new__(CTSDTrainer)
    tr.wrapper, tr.num_train_timesteps, tr.is_unet = wrapper, 1000, True
    tr.train_scheduler = types.SimpleNamespace(
        config=types.SimpleNamespace(num_train_timesteps=1000, prediction_type=d["prediction_type"]),
        add_noise=lambda x0, n, t: SO.add_noise(acp, x0, n, t), get_velocity=lambda x0, n, t: SO.get_velocity(acp, x0, n, t))
    tr.common_config, tr.training_config = d["common_config"], d["training_config"]
    tr.reference_latent_count = d["training_config"].get("reference_frame_count", 0)
    tr.loss_coef = d["training_config"].get("loss_coef_dict", {}).get("sd", 1.0)
    gen = torch.Generator().manual_seed(d["generator_seed"])
    noise, ts, _ = tr.draw_training_inputs(lat.shape, gen)
    assert ts.dtype == torch.int64 and ts.shape == ((B, T) if d["common_config"].get("frame_prediction_style") == "diffusion_forcing" else (B,))
    loss = tr.loss(lat, {"c": d["batch"]["c"]}, generator=gen, timestep_indices=ts, noise=noise)
    x_t, t_seen = seen[0]
    assert torch.equal(t_seen, d["timesteps"])
    assert torch.allclose(x_t, d["noisy_latents"], atol=2e-2)                  # the model input travels as bf16
    assert abs(loss.item() - d["loss"].item()) / d["loss"].item() < 1e-2, (loss.item(), d["loss"].item())
    loss.backward()
    assert abs((0.3 - d["lr"] * w.grad).item() - d["w_after"].item()) < 2e-3
