"""BASELINE.json configs[2] at FULL size (24 layers, d = 1536, 6 views x 16 frames x 32x56 latents, CFG batch 2,
154 text tokens - the workload `bench.py` times): the fp32 CPU oracle cannot run this (0.4 PFLOP per forward), so the
checks are size-independent properties of the function, each of which breaks on an addressing / tiling / layout bug
that only shows at this scale (86 016 token rows, 16 k attention workgroups, GEMM tile grids of 336 x 48):

  * determinism: two forwards are bit-identical;
  * sample independence: the CFG halves (batch elements) do not influence each other - replacing sample 1 leaves
    sample 0's prediction bit-identical;
  * with cross-view and temporal mixing disabled (`disable_crossview / disable_temporal`: AlphaBlender alpha = 1,
    crossview_temporal.py:56-66) every (frame, view) image is an independent SD 3.5 forward: permuting the images
    permutes the predictions, and a 6-image sub-batch reproduces the corresponding slice of the 192-image batch
    (to bf16 round-off: its small GEMM grids take the split-K path);
  * the fused CFG + FlowMatch-Euler update (ctsd.py:1548-1575) equals its fp32 formula on the full latent tensor and
    is linear in the sigma step.
The bf16 kernels are row-wise deterministic (fixed reduction order per output row), so the equalities are exact.
"""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu
bf16 = torch.bfloat16


@pytest.fixture(scope="module")
def full(request):
    if not torch.cuda.is_available():
        pytest.fail("the gpu-marked tests need a HIP device (torch.cuda.is_available() is False)")
    import bench
    from opendwm_amd import _lib
    _lib.load()
    dev = torch.device("cuda:0")
    model = bench.build_model(dict(bench.MODEL_KWARGS), dev, seed=0)
    cond = bench.make_conditions(dev, seed=0)
    w = bench.WORKLOAD
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(2 * w["B"], w["T"], w["V"], w["C"], w["H"], w["W"], device=dev, generator=g).to(bf16)
    ts = torch.full((2 * w["B"], w["T"], w["V"]), 500.0, device=dev)
    yield model, cond, x, ts, dev
    del model
    torch.cuda.empty_cache()


def _log(name, **kw):
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/gpu_parity.log", "a") as f:
        f.write(name + " " + " ".join(f"{k}={v}" for k, v in kw.items()) + "\n")


def fwd(model, x, ts, cond):
    return model(x, ts, **cond)[0][0]


def test_full_size_determinism_and_sample_independence(full):
    model, cond, x, ts, dev = full
    a, b = fwd(model, x, ts, cond), fwd(model, x, ts, cond)
    assert a.shape == x.shape and torch.isfinite(a.float()).all()
    assert torch.equal(a, b)
    x2 = x.clone()
    x2[1] = torch.randn_like(x2[1].float()).to(bf16)
    cond2 = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in cond.items()}
    cond2["encoder_hidden_states"][1] *= -1.0
    cond2["pooled_projections"][1] *= 0.5
    c = fwd(model, x2, ts, cond2)
    _log("fullsize_sample_independence", sample0_equal=bool(torch.equal(a[0], c[0])), sample1_changed=bool(not torch.equal(a[1], c[1])))
    assert torch.equal(a[0], c[0]) and not torch.equal(a[1], c[1])


def test_full_size_images_independent_without_mixing(full):
    model, cond, x, ts, dev = full
    B2, T, V = x.shape[:3]
    off = dict(cond)
    off["disable_crossview"] = torch.ones(B2, dtype=torch.bool, device=dev)
    off["disable_temporal"] = torch.ones(B2, dtype=torch.bool, device=dev)
    base = fwd(model, x, ts, off)
    # permute the frames of every sample (inputs and per-frame conditions alike)
    perm = torch.randperm(T, generator=torch.Generator().manual_seed(1)).to(dev)
    per_frame = ("encoder_hidden_states", "pooled_projections", "added_time_ids")
    offp = {k: (v[:, perm] if k in per_frame else v) for k, v in off.items()}
    outp = fwd(model, x[:, perm].contiguous(), ts, offp)
    eq_perm = bool(torch.equal(outp, base[:, perm]))
    # a 6-image sub-batch (one frame of sample 1) reproduces its slice of the 192-image batch
    t0 = 11
    sub = {k: (v[1:2, t0:t0 + 1] if k in per_frame else v[1:2] if torch.is_tensor(v) and v.shape[0] == B2 else v) for k, v in off.items()}
    outs = fwd(model, x[1:2, t0:t0 + 1].contiguous(), ts[1:2, t0:t0 + 1], sub)
    # (the 11-tile GEMM grids of the sub-batch take the split-K path, i.e. another fp32 summation order: compared to
    # bf16 round-off instead of bit for bit; an addressing error would be O(1))
    sub_rel = ((outs[0, 0].float() - base[1, t0].float()).norm() / base[1, t0].float().norm()).item()
    eq_sub = sub_rel < 1e-2
    # and the mixing branches do matter when enabled
    on = fwd(model, x, ts, cond)
    _log("fullsize_image_independence", frame_permutation_equal=eq_perm, sub_batch_rel=f"{sub_rel:.3e}",
         mixing_changes_output=bool(not torch.equal(on, base)))
    assert eq_perm and eq_sub and not torch.equal(on, base)


def test_full_size_cfg_euler_update(full):
    from opendwm_amd import ops
    model, cond, x, ts, dev = full
    g = torch.Generator(device="cuda").manual_seed(9)
    pred = torch.randn(x.shape, device=dev, generator=g).to(bf16)
    lat = torch.randn(x.shape[1:], device=dev, generator=g)[None].contiguous()
    guidance, ds = 4.0, -0.0371
    u, c = pred[0].float(), pred[1].float()
    want = lat + ds * (u + guidance * (c - u))
    got = lat.clone()
    model_in = torch.empty_like(pred)
    ops.cfg_euler_step(pred, got, guidance, ds, model_in=model_in)
    err = (got - want).abs().max().item()
    half = lat.clone()
    ops.cfg_euler_step(pred, half, guidance, ds / 2)
    ops.cfg_euler_step(pred, half, guidance, ds / 2)
    lin = (half - got).abs().max().item()
    _log("fullsize_cfg_euler", max_abs_err=f"{err:.3e}", two_half_steps_vs_one=f"{lin:.3e}")
    assert err < 1e-5 and lin < 1e-5
    assert torch.equal(model_in[0], got[0].to(bf16)) and torch.equal(model_in[1], got[0].to(bf16))


# ---------------------------------------------------------------- the text+layout model (bench.py's headline variant)
@pytest.fixture(scope="module")
def full_layout(request):
    if not torch.cuda.is_available():
        pytest.fail("the gpu-marked tests need a HIP device (torch.cuda.is_available() is False)")
    import bench
    from opendwm_amd import _lib
    _lib.load()
    dev = torch.device("cuda:0")
    model = bench.build_model(bench.variant_kwargs(True), dev, seed=0)
    cond = bench.make_conditions(dev, seed=0, layout=True)
    w = bench.WORKLOAD
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(2 * w["B"], w["T"], w["V"], w["C"], w["H"], w["W"], device=dev, generator=g).to(bf16)
    ts = torch.full((2 * w["B"], w["T"], w["V"]), 500.0, device=dev)
    yield model, cond, x, ts, dev
    del model
    torch.cuda.empty_cache()


def test_full_size_layout_adapter_and_pointwise_temporal(full_layout):
    """BASELINE configs[2] as written (text+layout): ImageAdapter residuals on the 86 016-row token grid + point-wise
    temporal attention (21 504 x 24 problems of L = 16).
      * recomputing the adapter (what bench.py times) and reusing its cached residuals give bit-identical predictions;
      * the layout images matter, and only for their own sample;
      * with the mixing branches disabled the forward is per image: permuting the frames - latents, text, time ids AND
        condition images - permutes the prediction exactly (checks the adapter's padded-grid convolutions and the
        residual addressing at full size)."""
    model, cond, x, ts, dev = full_layout
    B2, T, V = x.shape[:3]
    model._adapter_cache = (None, None)
    model.cache_adapter_residuals = True
    model.adapter_cache_dtype = torch.float32                      # (default) cached residuals through the fp32 path
    p32 = fwd(model, x, ts, cond)
    model._adapter_cache = (None, None)
    model.adapter_cache_dtype = torch.bfloat16                     # the same arithmetic as the per-step recompute
    a = fwd(model, x, ts, cond)
    assert model._adapter_cache[0] is not None
    b = fwd(model, x, ts, cond)                                    # cached fp32 residuals, added by dwm_add_f32_f32_inplace
    model.cache_adapter_residuals = False
    c = fwd(model, x, ts, cond)                                    # recomputed, zero convolutions adding from their GEMM epilogues
    model.cache_adapter_residuals = True
    assert torch.isfinite(a.float()).all() and torch.equal(a, b) and torch.equal(a, c)
    d32 = ((p32.double() - a.double()).norm() / a.double().norm()).item()
    assert 0.0 < d32 < 1e-2, d32                                   # the fp32-path residuals: close to, not equal to, the bf16 ones
    cond2 = dict(cond)
    img2 = cond["condition_image_tensor"].clone()
    img2[1] = 1.0 - img2[1]
    cond2["condition_image_tensor"] = img2
    d = fwd(model, x, ts, cond2)
    own_sample_only = bool(torch.equal(a[0], d[0]) and not torch.equal(a[1], d[1]))
    off = dict(cond)
    off["disable_crossview"] = torch.ones(B2, dtype=torch.bool, device=dev)
    off["disable_temporal"] = torch.ones(B2, dtype=torch.bool, device=dev)
    base = fwd(model, x, ts, off)
    perm = torch.randperm(T, generator=torch.Generator().manual_seed(2)).to(dev)
    per_frame = ("encoder_hidden_states", "pooled_projections", "added_time_ids", "condition_image_tensor")
    offp = {k: (v[:, perm].contiguous() if k in per_frame else v) for k, v in off.items()}
    outp = fwd(model, x[:, perm].contiguous(), ts, offp)
    eq_perm = bool(torch.equal(outp, base[:, perm]))
    _log("fullsize_layout", cached_equals_recomputed=True, layout_changes_own_sample_only=own_sample_only,
         frame_permutation_equal=eq_perm, temporal_mixing_changes_output=bool(not torch.equal(a, base)))
    assert own_sample_only and eq_perm and not torch.equal(a, base)


def _full_frame_shard_worker(rank, world, port, path):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from opendwm_amd import _lib
    from opendwm_amd.pipeline import CTSDDenoiser
    _lib.load()
    dev = torch.device("cuda:0")
    model = bench.build_model(bench.variant_kwargs(True), dev, seed=0)
    cond = bench.make_conditions(dev, seed=0, layout=True)
    w = bench.WORKLOAD
    lat = torch.randn(w["B"], w["T"], w["V"], w["C"], w["H"], w["W"], device=dev, generator=torch.Generator(device="cuda").manual_seed(5))
    den = CTSDDenoiser(model, guidance_scale=4.0, inference_steps=40, frame_group=dist.group.WORLD)
    out = den.run(lat, cond, stop=2)
    if rank == 0:
        torch.save(out.cpu(), path)
    dist.barrier()
    dist.destroy_process_group()


def test_full_size_frame_shard_two_ranks(full_layout):
    """opendwm_amd.sharding at the headline size: the 16 frames of the one sample on two ranks (8 frames / 8 of the 16
    token rows each, 12 temporal blocks x 2 all-to-alls of 132 MB per rank and step; gloo through host memory here, both
    ranks on the one GPU).  Two denoise steps must reproduce the single-process latents - bit for bit, since every kernel
    is row-wise deterministic and the 43 008-row GEMM grids stay on the non-split path; asserted to bf16 round-off."""
    import tempfile
    import torch.multiprocessing as mp
    from opendwm_amd.pipeline import CTSDDenoiser
    import bench
    model, cond, x, ts, dev = full_layout
    w = bench.WORKLOAD
    lat = torch.randn(w["B"], w["T"], w["V"], w["C"], w["H"], w["W"], device=dev, generator=torch.Generator(device="cuda").manual_seed(5))
    single = CTSDDenoiser(model, guidance_scale=4.0, inference_steps=40).run(lat, cond, stop=2).cpu()
    ctx = mp.get_context("spawn")
    port = 29500 + (os.getpid() + 23) % 2000
    path = os.path.join(tempfile.mkdtemp(), "full_frame_shard.pt")
    procs = [ctx.Process(target=_full_frame_shard_worker, args=(r, 2, port, path)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    sharded = torch.load(path)
    rel = ((sharded.double() - single.double()).norm() / single.double().norm()).item()
    _log("fullsize_frame_shard", ranks=2, bit_equal=bool(torch.equal(sharded, single)), rel=f"{rel:.3e}")
    assert sharded.shape == single.shape and rel < 5e-3
