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
    torch.save(cases, os.path.join(HERE, "reference_conditions.pt"))
    print("wrote reference_conditions.pt", len(cases), "cases")


if __name__ == "__main__":
    main()
