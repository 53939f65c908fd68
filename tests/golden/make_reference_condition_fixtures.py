"""Golden vectors from EXECUTING the reference's condition builder - runs only where /root/reference exists.

The REAL CrossviewTemporalSD.get_conditions (src/dwm/pipelines/ctsd.py:159-453) with its helpers get_camera_transform_ids
(:85-95) and get_action_ids (:97-156) is called with `text_encoder=None` (the text branch :176-253 is the callers' business:
the boundary takes embedded text), `dwm.pipelines.ctsd` imported behind import-only stubs (make_reference_driver_fixtures.py).
The `common_config` index lists are read from the reference's own example JSONs.

usage: python tests/golden/make_reference_condition_fixtures.py  ->  tests/golden/reference_conditions.pt
"""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests.golden.make_reference_driver_fixtures import _Finder          # noqa: E402


def poses(g, *shape, scale=0.5):
    """random rigid transforms [..., 4, 4]: rotation about z + translation"""
    a = torch.randn(*shape, generator=g) * 0.2
    t = torch.randn(*shape, 3, generator=g) * scale
    m = torch.eye(4).repeat(*shape, 1, 1)
    m[..., 0, 0], m[..., 0, 1], m[..., 1, 0], m[..., 1, 1] = a.cos(), -a.sin(), a.sin(), a.cos()
    m[..., :3, 3] = t
    return m


def make_batch(g, B, T, V, sensors=6, images=True):
    ego = poses(g, B, T, sensors)
    ego = torch.cumsum(ego - torch.eye(4), 1) * 0.3 + torch.eye(4)          # a drifting trajectory per sensor
    ego[..., 3, :] = torch.tensor([0.0, 0.0, 0.0, 1.0])
    batch = {
        "pts": torch.zeros(B, T, V),
        "fps": torch.tensor([10.0, 12.0, 2.0][:B]),
        "camera_intrinsics": torch.rand(B, T, V, 3, 3, generator=g) * 800 + 100,
        "camera_transforms": poses(g, B, T, V, scale=1.5),
        "image_size": torch.tensor([448.0, 256.0]).repeat(B, T, V, 1) + torch.rand(B, T, V, 2, generator=g),
        "ego_transforms": ego,
        "crossview_mask": torch.rand(B, V, V, generator=g) > 0.4,
    }
    if images:
        batch["3dbox_images"] = torch.rand(B, T, V, 3, 8, 8, generator=g)
        batch["hdmap_images"] = torch.rand(B, T, V, 3, 8, 8, generator=g)
    return batch


def text_embedding(prompt, L, D, seed):
    """deterministic stand-in for a text encoder: depends on the prompt string only"""
    h = sum((i + 1) * ord(ch) for i, ch in enumerate(prompt)) % 100003
    return torch.randn(L, D, generator=torch.Generator().manual_seed(seed * 100003 + h)) * 0.5


def stream_model(x, ts, encoder_hidden_states=None, pooled_projections=None, added_time_ids=None, condition_image_tensor=None, **kw):
    """cheap denoiser that sees every streamed condition: text (tokens and pooled), action ids, layout images"""
    act = added_time_ids[..., -2:].clamp(-5, 5).mean(-1)
    c = encoder_hidden_states.float().mean((-1, -2)) + 0.5 * pooled_projections.float().mean(-1) + 0.02 * act \
        + 0.3 * condition_image_tensor.float().mean((-1, -2, -3))
    return 0.1 * x + 1e-4 * ts[..., None, None, None] + 0.05 * c[..., None, None, None]


def streaming_ingest(C, layout_cfg):
    """the REAL StreamingCrossviewTemporalSD.fifo_inference_pipeline / send_frame_condition / get_conditions (text branch
    included: flatten_clip_text, the clip / t5 assembly :205-253) with stand-in text encoders and denoiser; prompts change
    from frame to frame and are re-embedded every 3rd frame only (text_prompt_interval)"""
    import diffusers
    from dwm.schedulers.temporal_independent import FlowMatchEulerDiscreteScheduler as Sched
    from tests.golden.make_reference_driver_fixtures import FakeVae, make_scheduler
    diffusers.schedulers.scheduling_flow_match_euler_discrete.FlowMatchEulerDiscreteSchedulerOutput = \
        lambda prev_sample: __import__("types").SimpleNamespace(prev_sample=prev_sample)

    class Model(diffusers.SD3Transformer2DModel):
        depth_net = None

        def forward(self, x, ts, **kw):
            return [stream_model(x.float(), ts.float(), **kw)], None, None

    C.CrossviewTemporalSD.sd3_encode_prompt_with_clip = staticmethod(
        lambda enc, tok, cc, prompt, device, num_images_per_prompt=1:
        (torch.stack([text_embedding(p, 3, enc.dim, enc.seed) for p in prompt]),
         torch.stack([text_embedding(p, 1, enc.dim, enc.seed + 7)[0] for p in prompt])))
    C.CrossviewTemporalSD.sd3_encode_prompt_with_t5 = staticmethod(
        lambda enc, tok, cc, max_sequence_length=77, prompt=None, num_images_per_prompt=1, device=None, joint_attention_dim=4096:
        torch.stack([text_embedding(p, 4, 12, 3) for p in prompt]))
    import types
    steps, T, B, V, total = 8, 4, 1, 6, 7
    p = object.__new__(C.StreamingCrossviewTemporalSD)
    p.model = Model()
    p.model_wrapper, p.model_dtype = p.model, torch.float32
    p.common_config = dict(layout_cfg, frame_prediction_style="diffusion_forcing")
    p.inference_config = dict(inference_steps=steps, guidance_scale=3.0, sequence_length_per_iteration=T, text_prompt_interval=3,
                              autoregression_data_exception_for_take_sequence=["crossview_mask", "fps"],
                              autoregression_condition_exception_for_take_sequence=[
                                  "disable_crossview", "disable_temporal", "crossview_attention_mask", "camera_intrinsics_norm", "camera2referego"])
    p.device, p.generator = torch.device("cpu"), torch.Generator().manual_seed(3)
    p.vae, p.is_temporal_vae = FakeVae(), False
    p.image_processor = types.SimpleNamespace(postprocess=lambda x, output_type=None: x)
    p.test_scheduler = make_scheduler(Sched, steps)
    p.text_encoders = [types.SimpleNamespace(dim=4, seed=1, device="cpu"), types.SimpleNamespace(dim=5, seed=2, device="cpu"), types.SimpleNamespace(device="cpu")]
    p.tokenizers = [None, None, None]
    p.text_encoder = p.tokenizer = None
    import contextlib
    p.get_autocast_context = lambda: contextlib.nullcontext()
    g = torch.Generator().manual_seed(41)
    batch = make_batch(g, B, total, V)
    batch["clip_text"] = [[[f"frame {t} view {v}" for v in range(V)] for t in range(total)] for _ in range(B)]
    shape = (B, T, V, 2, 3, 4)
    inp = clone(batch)
    inp["clip_text"] = batch["clip_text"]
    r = C.StreamingCrossviewTemporalSD.fifo_inference_pipeline(p, shape, batch, "pt")
    print("streaming_ingest", list(r["images"].shape), "queued conditions", {k: (None if v is None else list(v.shape)) for k, v in p.conditions.items()})
    return dict(common_config=p.common_config, inference_config=p.inference_config, batch=inp, shape=shape, total=total, seed=3, steps=steps,
                images=r["images"], final_latents=p.latents, final_conditions=p.conditions)


def autoregressive_batch(C, layout_cfg):
    """the REAL autoregressive_inference_pipeline over the REAL inference_pipeline and get_conditions (stand-in text encoders
    and denoiser as in streaming_ingest): the conditions of every window come from that window's clip of the batch"""
    import contextlib
    import types
    import diffusers
    from dwm.schedulers.temporal_independent import FlowMatchEulerDiscreteScheduler as Sched
    from tests.golden.make_reference_driver_fixtures import FakeVae, make_scheduler

    class Model(diffusers.SD3Transformer2DModel):
        depth_net = None

        def forward(self, x, ts, **kw):
            return [stream_model(x.float(), ts.float(), **kw)], None, None

    steps, T, B, V, total = 3, 4, 1, 6, 10
    p = object.__new__(C.CrossviewTemporalSD)
    p.model = Model()
    p.model_wrapper, p.model_dtype = p.model, torch.float32
    p.common_config = dict(layout_cfg)
    p.inference_config = dict(inference_steps=steps, guidance_scale=3.0, sequence_length_per_iteration=T, reference_frame_count=1,
                              autoregression_data_exception_for_take_sequence=["crossview_mask", "fps"])
    p.device, p.generator = torch.device("cpu"), torch.Generator().manual_seed(5)
    p.vae, p.is_temporal_vae = FakeVae(), False
    p.image_processor = types.SimpleNamespace(postprocess=lambda x, output_type=None: x)
    p.test_scheduler = make_scheduler(Sched, steps)
    p.text_encoders = [types.SimpleNamespace(dim=4, seed=1, device="cpu"), types.SimpleNamespace(dim=5, seed=2, device="cpu"), types.SimpleNamespace(device="cpu")]
    p.tokenizers = [None, None, None]
    p.text_encoder = p.tokenizer = None
    p.get_autocast_context = lambda: contextlib.nullcontext()
    real = C.CrossviewTemporalSD.inference_pipeline
    p.inference_pipeline = lambda *a, _p=p, **kw: real(_p, *a, **kw)
    p.get_latent_sequence_length = lambda n, _p=p: C.CrossviewTemporalSD.get_latent_sequence_length(_p, n)
    g = torch.Generator().manual_seed(43)
    batch = make_batch(g, B, total, V)
    batch["clip_text"] = [[[f"frame {t} view {v}" for v in range(V)] for t in range(total)] for _ in range(B)]
    shape = (B, T, V, 2, 3, 4)
    inp = clone(batch)
    inp["clip_text"] = batch["clip_text"]
    r = C.CrossviewTemporalSD.autoregressive_inference_pipeline(p, shape, batch, "pt")
    print("autoregressive_batch", list(r["images"].shape))
    return dict(common_config=p.common_config, inference_config=p.inference_config, batch=inp, shape=shape, total=total, seed=5, steps=steps,
                images=r["images"])


def clone(x):
    return {k: (v.clone() if torch.is_tensor(v) else v) for k, v in x.items()}


def main():
    sys.meta_path.insert(0, _Finder())
    sys.path.insert(0, "/root/reference/src")
    import dwm.pipelines.ctsd as C
    ex = "/root/reference/examples/"
    layout_cfg = json.load(open(ex + "ctsd_35_df16_6views_video_generation_with_layout.json"))["pipeline"]["common_config"]
    text_cfg = json.load(open(ex + "ctsd_35_6views_video_generation.json"))["pipeline"]["common_config"]
    keep = ("condition_on_all_frames", "uncondition_image_color", "added_time_ids", "camera_intrinsic_embedding_indices",
            "camera_intrinsic_denom_embedding_indices", "camera_transform_embedding_indices", "camera_ego_sensor_indices")
    layout_cfg = {k: layout_cfg[k] for k in keep if k in layout_cfg}
    text_cfg = {k: text_cfg[k] for k in keep if k in text_cfg}
    g = torch.Generator().manual_seed(0)
    B, T, V = 2, 3, 6
    cases = {}

    def run(name, cfg, batch, latent_shape, **kw):
        inp = clone(batch)
        res = C.CrossviewTemporalSD.get_conditions(object(), None, None, cfg, latent_shape, clone(batch), torch.device("cpu"),
                                                   torch.float32, **{k: (v.clone() if torch.is_tensor(v) else v) for k, v in kw.items()})
        cases[name] = dict(common_config=cfg, batch=inp, latent_shape=tuple(latent_shape), kwargs=kw, result=res)
        print(name, {k: (None if v is None else list(v.shape)) for k, v in res.items()})

    shape = (B, T, V, 16, 4, 4)
    run("layout_cfg", layout_cfg, make_batch(g, B, T, V), shape, do_classifier_free_guidance=True)
    run("layout_nocfg", layout_cfg, make_batch(g, B, T, V), shape)
    run("text_only_first_frame_images", dict(text_cfg, condition_on_all_frames=False), make_batch(g, B, T, V), shape,
        do_classifier_free_guidance=True)
    b = make_batch(g, B, T, V)
    b["ego_transforms"][1] = torch.eye(4)                       # sample 1: no ego motion given -> unconditioned action
    run("masks", dict(layout_cfg, disable_crossview=True), b, shape, do_classifier_free_guidance=True,
        _3dbox_condition_mask=torch.tensor([True, False]), hdmap_condition_mask=torch.tensor([[True, False, True], [False, True, True]]),
        action_condition_mask=torch.tensor([True, True]))
    run("action_mask_off", layout_cfg, make_batch(g, B, T, V), shape, action_condition_mask=torch.tensor([False, True]))
    one = make_batch(g, B, 1, V)
    run("streaming_first", layout_cfg, one, (B, 1, V, 16, 4, 4), streaming_mode=True, do_classifier_free_guidance=True)
    run("streaming_next", layout_cfg, make_batch(g, B, 1, V), (B, 1, V, 16, 4, 4), streaming_mode=True,
        prev_ego_transforms=one["ego_transforms"], do_classifier_free_guidance=True)
    ev = make_batch(g, B, T, V, images=False)
    ev["is_uncalibrated"] = torch.tensor([False, True])
    run("explicit_view", dict(text_cfg, explicit_view_modeling=True, disable_temporal=True), ev, shape, do_classifier_free_guidance=True,
        explicit_view_modeling_mask=torch.tensor([True, True]))
    ev2 = make_batch(g, B, T, V, images=False)
    del ev2["ego_transforms"]
    run("explicit_view_no_ego", dict(text_cfg, explicit_view_modeling=True), ev2, shape,
        explicit_view_modeling_mask=torch.tensor([False, True]))
    run("temporal_vae_5_to_2", layout_cfg, make_batch(g, B, 5, V), (B, 2, V, 16, 4, 4), do_classifier_free_guidance=True,
        latents_shape=(B, 2, V, 16, 4, 4))
    run("temporal_vae_4_to_2", text_cfg, make_batch(g, B, 4, V), (B, 2, V, 16, 4, 4), latents_shape=(B, 2, V, 16, 4, 4))
    cases["streaming_ingest"] = streaming_ingest(C, layout_cfg)           # (installs the stand-in text encoders used below)
    cases["autoregressive_batch"] = autoregressive_batch(C, layout_cfg)
    # ---- the text branch alone: flatten_clip_text (:39-82) and the assembly inside get_conditions (:205-253)
    flat_cases = {}
    for name, text, mask, cfg_ in (("per_sample", ["a car", "a bus"], None, False), ("per_sample_cfg", ["a car", "a bus"], None, True),
                                   ("per_sample_masked", ["a car", "a bus"], [True, False], True),
                                   ("nested", [[["a", "b"], ["c", "d"], ["e", "f"]]], None, True),
                                   ("nested_masked", [[["a", "b"], ["c", "d"]], [["g", "h"], ["i", "j"]]], [[[True, False], [True, True]], False], False)):
        flat, shape_ = [], []
        C.CrossviewTemporalSD.flatten_clip_text(text, flat, shape_, text_condition_mask=mask, do_classifier_free_guidance=cfg_)
        flat_cases[name] = dict(text=text, mask=mask, cfg=cfg_, flat=flat, shape=shape_)
    cases["flatten_clip_text"] = flat_cases
    import diffusers
    import types
    encs = [types.SimpleNamespace(dim=4, seed=1, device="cpu"), types.SimpleNamespace(dim=5, seed=2, device="cpu"), types.SimpleNamespace(device="cpu")]
    tb = make_batch(g, B, T, V)
    text_results = {}
    for name, text, mask in (("shared_prompt", ["a car", "a bus"], [True, False]),
                             ("per_view_prompts", [[[f"s{b} t{t} v{v}" for v in range(V)] for t in range(T)] for b in range(B)], None)):
        tb["clip_text"] = text
        r = C.CrossviewTemporalSD.get_conditions(type("M", (diffusers.SD3Transformer2DModel,), {})(), encs, [None] * 3, text_cfg, shape, clone(tb),
                                                 torch.device("cpu"), torch.float32, text_condition_mask=mask, do_classifier_free_guidance=True)
        text_results[name] = dict(text=text, mask=mask, encoder_hidden_states=r["encoder_hidden_states"], pooled_projections=r["pooled_projections"])
    cases["text_branch"] = text_results
    # SD 2.1 (UNet) text branch (:186-203): one CLIP encoder, no pooled projections
    class Ids(list):
        def to(self, device):
            return self
    tok = lambda prompts, **kw: types.SimpleNamespace(input_ids=Ids(prompts))
    tok.model_max_length = 77
    enc = lambda prompts: (torch.stack([text_embedding(p, 5, 8, 11) for p in prompts]),)
    unet_results = {}
    for name, text in (("shared_prompt", ["a car", "a bus"]),
                       ("per_view_prompts", [[[f"s{b} t{t} v{v}" for v in range(V)] for t in range(T)] for b in range(B)])):
        tb["clip_text"] = text
        r = C.CrossviewTemporalSD.get_conditions(type("U", (diffusers.UNetSpatioTemporalConditionModel,), {})(), enc, tok, text_cfg, shape,
                                                 clone(tb), torch.device("cpu"), torch.float32, do_classifier_free_guidance=True)
        assert "pooled_projections" not in r
        unet_results[name] = dict(text=text, encoder_hidden_states=r["encoder_hidden_states"])
    cases["text_branch_sd21"] = unet_results
    torch.save(cases, os.path.join(HERE, "reference_conditions.pt"))
    print("wrote reference_conditions.pt", len(cases), "cases")


if __name__ == "__main__":
    main()
