"""The model kwargs of one batch - the caller's side of the hot path: everything `CrossviewTemporalSD.get_conditions`
(src/dwm/pipelines/ctsd.py:159-453) derives from a dataset batch EXCEPT the text encoders, which stay outside this
package (the boundary takes embedded text).  Pure host-side tensor logic on whatever device the batch
lives on; the classifier-free-guidance doubling puts the unconditional half first, as the reference does.

    flatten_clip_text        ctsd.py:39-82     nested per-sample / per-frame / per-view prompts -> flat list (+ "" for CFG)
    assemble_sd3_text        ctsd.py:219-253   the two CLIP embeddings padded to the T5 width and stacked with T5
    assemble_clip_text       ctsd.py:186-203   SD 2.1: one CLIP embedding per prompt, repeated or unflattened
    camera_transform_ids     ctsd.py:85-95     intrinsics / image size and extrinsic entries picked by index lists
    action_ids               ctsd.py:97-156    speed [km/h] and steering from consecutive ego poses (-1000 = unconditioned)
    build_conditions         ctsd.py:255-453   layout images (3-D boxes + HD map, unconditional colour), added_time_ids,
                                               explicit-view camera matrices, flags, masks, temporal-VAE frame striding

Pinned against the reference functions themselves (tests/golden/make_reference_condition_fixtures.py,
tests/test_reference_fixtures_cpu.py)."""
from __future__ import annotations

from typing import Dict, Optional

import torch


def flatten_clip_text(clip_text, text_condition_mask=None, do_classifier_free_guidance: bool = False):
    """(flat prompt list, parsed shape) of a nested prompt structure - a string per sample, or lists per sample / frame /
    view (ctsd.py:39-82).  Under CFG the unconditional prompts ("" for every leaf) come first, the level-0 count doubles;
    prompts whose `text_condition_mask` entry (bool, or nested list of bools) is false are replaced by ""."""
    flat, shape = [], []

    def walk(node, level, mask, cfg):
        count = 0
        if isinstance(node, list) and len(shape) <= level:
            shape.append(0)
        if cfg:
            if isinstance(node, str):
                flat.append("")
                count += 1
            else:
                for child in node:
                    walk(child, level + 1, mask, cfg)
                    count += 1
        if level == 0 or not cfg:
            if isinstance(node, str):
                flat.append(node if mask is None or (isinstance(mask, bool) and mask) else "")
                count += 1
            else:
                for i, child in enumerate(node):
                    walk(child, level + 1, None if mask is None else (mask[i] if isinstance(mask, list) else mask), False)
                    count += 1
        if isinstance(node, list):
            shape[level] = count

    walk(clip_text, 0, text_condition_mask, do_classifier_free_guidance)
    return flat, shape


def assemble_sd3_text(clip_embeddings, clip_pooled, t5_embeddings: torch.Tensor, parsed_shape, sequence_length: int,
                      view_count: int, dtype):
    """SD 3 text conditioning from the encoder outputs for the FLAT prompt list (ctsd.py:219-253): the two CLIP hidden
    states concatenated on the feature axis and zero-padded to the T5 width, stacked with the T5 states on the token
    axis; pooled CLIP vectors concatenated.  One prompt per sample is repeated over frames and views, otherwise the
    flat axis is unflattened to the parsed (sample, frame, view) shape.
    -> encoder_hidden_states [B', T, V, L, D], pooled_projections [B', T, V, P]"""
    clip = torch.cat(list(clip_embeddings), dim=-1)
    pooled = torch.cat(list(clip_pooled), dim=-1)
    clip = torch.nn.functional.pad(clip, (0, t5_embeddings.shape[-1] - clip.shape[-1]))
    text = torch.cat([clip, t5_embeddings], dim=-2)
    if len(parsed_shape) == 1:
        text = text[:, None, None].repeat(1, sequence_length, view_count, 1, 1).to(dtype=dtype)
        pooled = pooled[:, None, None].repeat(1, sequence_length, view_count, 1).to(dtype=dtype)
    else:
        text = text.unflatten(0, parsed_shape).to(dtype=dtype)
        pooled = pooled.unflatten(0, parsed_shape).to(dtype=dtype)
    return text, pooled


def assemble_clip_text(text_embeddings: torch.Tensor, parsed_shape, sequence_length: int, view_count: int) -> torch.Tensor:
    """SD 2.1 text conditioning (ctsd.py:186-203): the CLIP hidden states of the FLAT prompt list [N, 77, D], one prompt per
    sample repeated over frames and views, otherwise unflattened to the parsed (sample, frame, view) shape
    -> encoder_hidden_states [B', T, V, 77, D]"""
    if len(parsed_shape) == 1:
        return text_embeddings[:, None, None].repeat(1, sequence_length, view_count, 1, 1)
    return text_embeddings.unflatten(0, parsed_shape)


def camera_transform_ids(batch: Dict, common_config: dict) -> torch.Tensor:
    """[B, T, V, n_intrinsic + n_extrinsic] (ctsd.py:85-95)"""
    k = batch["camera_intrinsics"].flatten(-2, -1)[..., common_config["camera_intrinsic_embedding_indices"]]
    size = batch["image_size"][..., common_config["camera_intrinsic_denom_embedding_indices"]]
    e = batch["camera_transforms"].flatten(-2, -1)[..., common_config["camera_transform_embedding_indices"]]
    return torch.cat([k / size, e], -1)


def action_ids(batch: Dict, common_config: dict, action_condition_mask: Optional[torch.Tensor] = None,
               streaming_mode: bool = False, prev_ego_transforms: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[B, T, V, 2] = (speed km/h, steering) of every frame from the pose change since the frame before (the first frame
    repeats the second one's); -1000 where unconditioned or (steering) standing still (ctsd.py:97-156)."""
    if streaming_mode:
        if batch["ego_transforms"].shape[1] != 1:
            raise ValueError("streaming mode takes one frame at a time")
        ego = torch.cat([batch["ego_transforms"] if prev_ego_transforms is None else prev_ego_transforms,
                         batch["ego_transforms"]], 1)
    else:
        ego = batch["ego_transforms"]
    pose = ego[:, :, common_config["camera_ego_sensor_indices"]]
    conditioned = (pose - torch.eye(4)[None, None, None]).sum((1, 2, 3, 4)).abs() > 1e-3
    if action_condition_mask is not None:
        conditioned = torch.logical_and(conditioned, action_condition_mask)
    rel = torch.linalg.solve(pose[:, :-1], pose[:, 1:])
    rel = torch.cat([rel[:, :1], rel], 1)
    dist = torch.norm(rel[..., :3, 3], dim=-1, keepdim=True)
    speed = 3.6 * dist * batch["fps"][:, None, None, None]                                # m/s -> km/h
    angle = torch.atan2(rel[..., 1, 0:1] - rel[..., 0, 1:2], rel[..., 0, 0:1] + rel[..., 1, 1:2])
    wheel_base, steering_ratio = 2.7, 14
    steering = torch.where(dist.abs() > 0.01, angle / dist * wheel_base * steering_ratio, -1000.0 * torch.ones_like(angle))
    ids = torch.cat([speed, steering], -1)
    ids = torch.where(conditioned[:, None, None, None], ids, -1000.0 * torch.ones_like(ids))
    return ids.chunk(2, dim=1)[-1] if streaming_mode else ids


def _layout_images(images: torch.Tensor, all_frames: bool, mask: Optional[torch.Tensor], color: float, cfg: bool, device):
    x = (images if all_frames else images[:, :1]).to(device).clone()          # the reference writes into the batch tensor
    if mask is not None:
        x[mask.logical_not().to(device)] = color
    return torch.cat([torch.ones_like(x) * color, x]) if cfg else x


def build_conditions(common_config: dict, latent_shape, batch: Dict, device, dtype, *,
                     encoder_hidden_states: Optional[torch.Tensor] = None,
                     pooled_projections: Optional[torch.Tensor] = None,
                     _3dbox_condition_mask: Optional[torch.Tensor] = None,
                     hdmap_condition_mask: Optional[torch.Tensor] = None,
                     action_condition_mask: Optional[torch.Tensor] = None,
                     explicit_view_modeling_mask: Optional[torch.Tensor] = None,
                     streaming_mode: bool = False, prev_ego_transforms: Optional[torch.Tensor] = None,
                     do_classifier_free_guidance: bool = False, latents_shape=None) -> Dict[str, Optional[torch.Tensor]]:
    """get_conditions (ctsd.py:159-453) with the text branch (:176-253) replaced by its result: `encoder_hidden_states`
    [B', T, V, L, D] / `pooled_projections` [B', T, V, P] as the text encoders produced them (B' = 2 B under CFG, empty
    prompt first).  Returns the keyword arguments of the model forward (crossview_temporal_dit.py:372-391)."""
    batch_size, view_count = latent_shape[0], latent_shape[2]
    sequence_length = batch["pts"].shape[1]
    cfg = do_classifier_free_guidance
    if cfg:
        batch_size *= 2

    all_frames = common_config.get("condition_on_all_frames", False)
    color = common_config.get("uncondition_image_color", 0)
    layout = []
    if "3dbox_images" in batch:
        layout.append(_layout_images(batch["3dbox_images"], all_frames, _3dbox_condition_mask, color, cfg, device))
    if "hdmap_images" in batch:
        layout.append(_layout_images(batch["hdmap_images"], all_frames, hdmap_condition_mask, color, cfg, device))
    condition_image_tensor = torch.cat(layout, -3) if layout else None

    added_time_ids = None
    kind = common_config.get("added_time_ids")
    if kind in ("fps_camera_transforms", "fps_camera_transforms_action"):
        parts = [batch["fps"][:, None, None, None].repeat(1, sequence_length, view_count, 1), camera_transform_ids(batch, common_config)]
        if kind == "fps_camera_transforms_action":
            parts.append(action_ids(batch, common_config, action_condition_mask, streaming_mode, prev_ego_transforms))
        added_time_ids = torch.cat(parts, -1)
        if cfg:
            uncond = added_time_ids
            if kind == "fps_camera_transforms_action":                    # the action is allowed to be guidance scaled
                uncond = torch.cat([added_time_ids[..., :-2], -1000 * torch.ones_like(added_time_ids[..., -2:])], -1)
            added_time_ids = torch.cat([uncond, added_time_ids], 0)
        added_time_ids = added_time_ids.to(device)

    explicit = common_config.get("explicit_view_modeling", False)
    camera_intrinsics_norm = camera2referego = None
    if explicit:
        cam = batch["camera_transforms"]
        if "ego_transforms" not in batch:
            ego = torch.eye(4).to(cam)[None, None, None].expand(cam.shape[0], cam.shape[1], cam.shape[2], -1, -1)
        else:
            ego = batch["ego_transforms"][:, :, -cam.shape[2]:]
        camera2referego = torch.linalg.inv(ego[:, 0, 0][:, None, None]) @ (ego @ cam)
        camera_intrinsics_norm = batch["camera_intrinsics"].clone()
        size = batch["image_size"]
        camera_intrinsics_norm[..., 0, 0] /= size[..., 0]
        camera_intrinsics_norm[..., 1, 1] /= size[..., 1]
        camera_intrinsics_norm[..., 0, 2] /= size[..., 0]
        camera_intrinsics_norm[..., 1, 2] /= size[..., 1]
        eye3, eye4 = torch.eye(3).to(cam), torch.eye(4).to(cam)
        if "is_uncalibrated" in batch:
            camera_intrinsics_norm[batch["is_uncalibrated"]] = eye3
            camera2referego[batch["is_uncalibrated"]] = eye4
        if explicit_view_modeling_mask is not None:
            off = explicit_view_modeling_mask.logical_not().to(device)
            camera_intrinsics_norm[off] = eye3
            camera2referego[off] = eye4
        if cfg:
            camera_intrinsics_norm = torch.cat([camera_intrinsics_norm, camera_intrinsics_norm], 0)
            camera2referego = torch.cat([camera2referego, camera2referego], 0)
        camera_intrinsics_norm, camera2referego = camera_intrinsics_norm.to(device), camera2referego.to(device)

    has_depth_input = "camera_intrinsics" in batch and "camera_transforms" in batch
    camera_intrinsics = camera_transforms = None
    if has_depth_input:
        camera_intrinsics, camera_transforms = batch["camera_intrinsics"].to(device), batch["camera_transforms"].to(device)
        if cfg:
            camera_intrinsics = torch.cat([camera_intrinsics, camera_intrinsics])
            camera_transforms = torch.cat([camera_transforms, camera_transforms])

    mask = None
    if "crossview_mask" in batch:
        mask = (torch.cat([batch["crossview_mask"], batch["crossview_mask"]]) if cfg else batch["crossview_mask"]).to(device)
    result = {
        "encoder_hidden_states": None if encoder_hidden_states is None else encoder_hidden_states.to(device=device, dtype=dtype),
        "condition_image_tensor": condition_image_tensor,
        "disable_crossview": torch.tensor([common_config.get("disable_crossview", False)], device=device).repeat(batch_size),
        "disable_temporal": torch.tensor([common_config.get("disable_temporal", False)], device=device).repeat(batch_size),
        "crossview_attention_mask": mask,
        "camera_intrinsics": camera_intrinsics, "camera_transforms": camera_transforms,
        "camera_intrinsics_norm": camera_intrinsics_norm, "camera2referego": camera2referego,
        "added_time_ids": added_time_ids,
    }
    if pooled_projections is not None:
        result["pooled_projections"] = pooled_projections.to(device=device, dtype=dtype)

    # a temporal VAE has fewer latent frames than the batch has frames: keep the conditions of the frames the latents
    # stand for (first frame + every stride-th one, :441-451)
    if latents_shape is not None and latents_shape[1] != sequence_length:
        pre = 1 if sequence_length % 2 == 1 else 0
        stride = (sequence_length - pre) // (latents_shape[1] - pre)
        for k, v in result.items():
            if v is not None and v.ndim > 1 and v.shape[1] == sequence_length:
                result[k] = torch.cat([v[:, :pre], v[:, pre::stride]], dim=1)
    return result
