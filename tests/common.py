"""Shared helpers of the test-suite (CPU and GPU legs)."""
import os

import torch

from oracle import ctsd_oracle as O

# extended cases of the long GPU tests (tests/conftest.py: the driver's run has a 1200 s limit)
HEAVY = bool(os.environ.get("DWM_HEAVY_TESTS"))

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


import contextlib


@contextlib.contextmanager
def capture_segsum_diff():
    """records every dwm_segsum_diff call of a backward (the d(alpha) sums of the AlphaBlender mixers): the kernel's total, the
    fp64 total of ITS OWN inputs and the fp64 sum of the absolute terms - what separates the kernel's arithmetic from the bf16
    rounding of the activations / gradients it is handed"""
    from opendwm_amd import train_ops as T
    calls, f0 = [], T.segsum_diff

    def wrapped(a, b, b2, rows_per_group=None):
        out = f0(a, b, b2, rows_per_group=rows_per_group)
        t = a.double() * (b.double() - b2.double())
        calls.append(dict(kernel=out.double().sum().item(), exact=t.sum().item(), abs=t.abs().sum().item()))
        return out
    T.segsum_diff = wrapped
    try:
        yield calls
    finally:
        T.segsum_diff = f0


def check_mixer_gradients(mixers: dict, calls: list, flat_tol: float):
    """mixers: name -> dict(rel = error of mix_factor.grad against fp32 autograd, cond = sum|terms| / |sum| of its d(alpha)).
    (1) every captured kernel call equals the fp64 sum of its own inputs to fp32 accumulation accuracy, measured against the
    sum of the absolute terms (so the statement holds at any conditioning); (2) where the sum is reasonably conditioned
    (cond < 400) the gradient meets the flat tolerance; (3) the badly conditioned ones (a -9.7 out of +-1e4 terms) still meet a
    bound that grows with the conditioning, 8e-4 x cond - the bf16 rounding of the activations and incoming gradients the sum
    is handed (2^-9 per operand, i.e. up to 2.8e-3 x cond if every term rounded the same way), amplified by the cancellation;
    the observed errors wander between 3.4e-4 and 4.1e-4 x cond from run to run (the order of the fp32 atomics that finish
    the upstream bias gradients, the reduction tree of the LayerNorm backward) - so a wrong operand, sign or scale reaching such
    a mixer (an O(1) relative error at any conditioning below 1250) cannot pass on the strength of (1) alone."""
    worst_kernel = max((abs(c["kernel"] - c["exact"]) / max(c["abs"], 1e-300) for c in calls), default=0.0)
    assert len(calls) >= len(mixers) and worst_kernel < 2e-6, (len(calls), len(mixers), worst_kernel)
    for c in calls:
        if c["abs"] < 50 * abs(c["exact"]):
            assert abs(c["kernel"] - c["exact"]) < 1e-4 * abs(c["exact"]), c
    for n, v in mixers.items():
        if v["cond"] < 400:
            assert v["rel"] < flat_tol, (n, v)
        else:
            assert v["rel"] < 8e-4 * v["cond"], (n, v)
    return worst_kernel
