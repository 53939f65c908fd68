"""Shared helpers of the test-suite (CPU and GPU legs)."""
import os

import torch

from oracle import ctsd_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def small_config(**over):
    """Full module graph (dual blocks, context-pre-only last block, one cross-view and two
    temporal VT blocks, implicit camera embedding) at 2 heads x 64 so the fp32 CPU oracle
    runs in well under a second.  Every K is a multiple of 64 (GEMM kernel requirement)."""
    cfg = O.make_config(
        num_layers=4, dual_attention_layers=[0, 1], num_attention_heads=2,
        caption_projection_dim=128, joint_attention_dim=128, pooled_projection_dim=64,
        pos_embed_max_size=32, sample_size=32, projection_class_embeddings_input_dim=11 * 256,
        crossview_block_layers=[1], temporal_block_layers=[2, 3])
    cfg.update(over)
    return cfg


SMALL_SHAPE = dict(B=2, T=3, V=3, H=8, W=12, text_len=10)


def small_inputs(cfg, seed=0, **shape_over):
    s = dict(SMALL_SHAPE)
    s.update(shape_over)
    return O.make_inputs(cfg, s["B"], s["T"], s["V"], s["H"], s["W"], seed=seed, text_len=s["text_len"])


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """||a - b||_F / ||b||_F in fp64 (the 'rel' of BASELINE.json's tolerance)."""
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def to_dev(d, device, float_dtype=None):
    out = {}
    for k, v in d.items():
        if torch.is_tensor(v):
            v = v.to(device)
            if float_dtype is not None and v.is_floating_point():
                v = v.to(float_dtype)
        out[k] = v
    return out
