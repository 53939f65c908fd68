"""Parity at the sizes BASELINE.json quotes, against the fp32 oracle evaluated ON THE DEVICE (plain PyTorch fp32 of
oracle/ctsd_oracle.py; the CPU needs ~15 minutes for one such forward, the GPU seconds):

  * configs[2] at FULL depth and size - 24 layers, d = 1536, latents [2,16,6,16,32,56] (CFG batch), 154 text tokens - for
    BOTH models of that config: text only (row-wise temporal attention) and text+layout (ImageAdapter + point-wise temporal
    attention), i.e. exactly the two forwards `bench.py` times;
  * 40 FlowMatch-Euler steps with classifier-free guidance (the whole loop of ctsd.py:1496-1575) at full width on a reduced
    geometry, bf16 HIP against the fp32 oracle loop: the error after the LAST step is what north_star bounds;
  * configs[1]: the SD 2.1 UNet at full width on 6 views x 6 frames x 32x56 latents;
  * configs[3]: one training forward + backward at full width (3-layer slice, one sample) - parameter gradients against fp32
    autograd through the oracle.
Tolerance: 2e-2 relative (Frobenius) for bf16 storage with fp32 accumulation, BASELINE.json north_star."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ctsd_oracle as O          # noqa: E402  (checker only)
from tests.common import HEAVY, rel_err, to_dev     # noqa: E402

pytestmark = pytest.mark.gpu
bf16 = torch.bfloat16
TOL = 2e-2


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("the gpu-marked tests need a HIP device (torch.cuda.is_available() is False)")
    from opendwm_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _log(name, **kw):
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/gpu_parity.log", "a") as f:
        f.write(name + " " + " ".join(f"{k}={v}" for k, v in kw.items()) + "\n")


def _oracle_on_device(fn):
    """the oracle's loops build their timestep tensors on the host: move them next to the sample"""
    def wrapped(sd, cfg, sample, timestep, **kw):
        return fn(sd, cfg, sample, timestep.to(sample.device), **kw)
    return wrapped


@pytest.mark.parametrize("layout", [False, True], ids=["text_only_rowwise", "text_layout_pointwise"])
def test_full_depth_full_size_forward_vs_oracle_on_device(dev, layout):
    """the model `bench.py` times (same constructor kwargs, same seeded weights, same synthetic conditions), one CFG forward"""
    import bench
    kwargs = bench.variant_kwargs(layout)
    model = bench.build_model(kwargs, dev, seed=0)
    model.cache_adapter_residuals = False            # as bench.py times it: the adapter inside the forward
    cond = bench.make_conditions(dev, seed=0, layout=layout)
    w = bench.WORKLOAD
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(2 * w["B"], w["T"], w["V"], w["C"], w["H"], w["W"], device=dev, generator=g).to(bf16)
    ts = torch.full((2 * w["B"], w["T"], w["V"]), 500.0, device=dev)
    with torch.no_grad():
        out = model(x, ts, **cond)[0][0].float()
    sd = {k: v.detach().float() for k, v in model.state_dict().items()}
    del model
    torch.cuda.empty_cache()
    cfg = O.make_config(**kwargs)
    with torch.no_grad():
        ref = O.dit_forward(sd, cfg, x.float(), ts, **{k: (v.float() if v.is_floating_point() else v) for k, v in cond.items()})
    e = rel_err(out, ref)
    _log("full_depth_forward", variant="text+layout" if layout else "text_only", layers=kwargs["num_layers"],
         latents=list(x.shape), rel=e, finite=bool(torch.isfinite(out).all()))
    del sd, ref
    torch.cuda.empty_cache()
    assert e < TOL, e


@pytest.mark.parametrize("layout", [False, True], ids=["text_only_rowwise", "text_layout_pointwise"])
def test_forty_step_denoise_vs_oracle_loop_on_device(dev, layout):
    """all 40 guided FlowMatch-Euler steps (examples/ctsd_35_6views_video_generation.json:34-35: 40 steps, guidance 4) at
    full width - 8 layers (dual joint blocks, cross-view after 1 and 5, temporal after 2, 3, 6, 7), 6 views x 4 frames x 32x56
    latents: CTSDDenoiser (bf16 model input, fp32 latents, fused CFG + Euler kernel) against O.denoise in fp32"""
    import bench
    from opendwm_amd.pipeline import CTSDDenoiser
    kwargs = bench.variant_kwargs(layout)
    n = 8
    kwargs.update(num_layers=n, dual_attention_layers=list(range(n)), crossview_block_layers=[1, 5], temporal_block_layers=[2, 3, 6, 7])
    model = bench.build_model(kwargs, dev, seed=0)
    model.cache_adapter_residuals = False            # as bench.py times it: the adapter inside every step
    wl = dict(bench.WORKLOAD, T=4)
    cond = bench.make_conditions(dev, seed=3, w=wl, layout=layout)
    g = torch.Generator(device="cuda").manual_seed(9)
    lat = torch.randn(1, wl["T"], wl["V"], wl["C"], wl["H"], wl["W"], device=dev, generator=g)
    den = CTSDDenoiser(model, guidance_scale=4.0, inference_steps=40)
    with torch.no_grad():
        out = den.run(lat, cond).clone()
        one = CTSDDenoiser(model, guidance_scale=4.0, inference_steps=40).run(lat, cond, stop=1).clone()
    sd = {k: v.detach().float() for k, v in model.state_dict().items()}
    del model, den
    torch.cuda.empty_cache()
    cfg = O.make_config(**kwargs)
    fwd0 = O.dit_forward
    O.dit_forward = _oracle_on_device(fwd0)
    try:
        condf = {k: (v.float() if v.is_floating_point() else v) for k, v in cond.items()}
        with torch.no_grad():
            ref1 = O.denoise(sd, cfg, lat, condf, steps=40, guidance_scale=4.0, stop=1)
            ref = O.denoise(sd, cfg, lat, condf, steps=40, guidance_scale=4.0)
    finally:
        O.dit_forward = fwd0
    e1, e40 = rel_err(one, ref1), rel_err(out, ref)
    # the displacement is what the model contributes: error relative to |x_40 - x_0| as well
    move = ((out.double() - ref.double()).norm() / (ref.double() - lat.double()).norm()).item()
    _log("denoise_40_steps", variant="text+layout" if layout else "text_only", layers=n, latents=list(lat.shape), rel_step1=e1,
         rel_step40=e40, rel_to_displacement=move, finite=bool(torch.isfinite(out).all()))
    assert e40 < TOL and e1 < TOL, (e1, e40)


# (layout, seed, frames, cached): the two models of the headline configuration; `cached` = the layout residuals computed once
# per prepare() through the fp32 path (model.cache_adapter_residuals, the default of the model class) instead of inside every
# step (what bench.py times).  Depth, width, views, resolution, text length and the 40 steps are always the full ones; the fp32
# oracle loop costs ~11 s per frame and case on the device, and the driver's `pytest -m gpu` has 1200 s for the whole suite
# (tests/conftest.py), so the default run holds:
#   * the text+layout model - the one the metric is quoted on - at the full 16 frames, adapter recomputed per step;
#   * a second seed (weights, conditions, noise) of it on 2 frames with the layout residuals cached in fp32 (optional for the budget);
#   * the text-only model on 2 frames.
# DWM_HEAVY_TESTS=1 adds the text-only model at 16 frames and seeds 1 and 2 of the per-step adapter mode; their results of
# this round are recorded in profiles/r4a_gpu_parity.log, r4b_gpu_parity.log (5.8e-3; 1.31 / 1.36 / 1.32e-2 over three seeds).
def _case(layout, seed, frames, cached, name, cost, optional=False, priority=5):
    return pytest.param(layout, seed, frames, cached, id=name, marks=pytest.mark.cost(cost, optional=optional, priority=priority))


FULL_DEPTH_CASES = [
    _case(True, 0, 16, False, "text_layout_pointwise", 185),
    _case(False, 0, 2, False, "text_only_rowwise_2f", 25),
    _case(True, 1, 2, True, "text_layout_seed1_2f_cached_fp32_adapter", 28, optional=True, priority=0),
] + ([
    _case(False, 0, 4, False, "text_only_rowwise_4f", 50),
    _case(True, 1, 4, True, "text_layout_seed1_4f_cached_fp32_adapter", 50),
    _case(True, 1, 4, False, "text_layout_seed1_4f", 50),
    _case(False, 0, 16, False, "text_only_rowwise", 170),
    _case(True, 2, 4, False, "text_layout_seed2_4f", 50),
] if HEAVY else [])
# Measured with fp32 residual streams (round 4, profiles/r4a_gpu_parity.log, r4b_gpu_parity.log): text-only 5.8e-3 (bf16 streams:
# 1.11e-2), text+layout 1.31 / 1.36 / 1.32e-2 over three seeds (bf16 streams: 1.63e-2).  What is left in the text+layout model is
# the bf16 error of the ImageAdapter recomputed in every step: step-invariant, different in the two CFG halves (so guidance
# multiplies it by up to 5) and summed coherently over the 40 steps; with the residuals computed once in fp32 it goes away.
TOL_40_STEPS = {False: 1.0e-2, True: 1.6e-2, "cached": 1.0e-2}


@pytest.mark.parametrize("layout,seed,frames,cached", FULL_DEPTH_CASES)
def test_forty_step_denoise_full_depth_full_size_vs_oracle_loop_on_device(dev, layout, seed, frames, cached):
    """What north_star bounds, on the configuration `bench.py` times: ALL 40 guided FlowMatch-Euler steps of the hot loop
    (ctsd.py:1496-1575) through the full 24-layer model on latents [1,16,6,16,32,56] (CFG batch 2, 154 text tokens) - bf16
    CTSDDenoiser against O.denoise in fp32 on the device (~18 PFLOP of fp32 per variant).  The error after the LAST step is
    the tolerance's subject; steps 1 / 10 / 20 / 30 are logged to show how it accumulates."""
    import bench
    from opendwm_amd.pipeline import CTSDDenoiser
    kwargs = bench.variant_kwargs(layout)
    model = bench.build_model(kwargs, dev, seed=seed)
    model.cache_adapter_residuals = bool(cached)
    wl = dict(bench.WORKLOAD, T=frames)
    cond = bench.make_conditions(dev, seed=3 + seed, w=wl, layout=layout)
    g = torch.Generator(device="cuda").manual_seed(9 + seed)
    lat = torch.randn(1, wl["T"], wl["V"], wl["C"], wl["H"], wl["W"], device=dev, generator=g)
    marks = (1, 10, 20, 30, 40)
    den = CTSDDenoiser(model, guidance_scale=4.0, inference_steps=40)
    ours = {}
    with torch.no_grad():
        den.prepare(lat, cond)
        for i in range(40):
            den.step(i)
            if i + 1 in marks:
                ours[i + 1] = den.result().clone()
    sd = {k: v.detach().float() for k, v in model.state_dict().items()}
    del model, den
    torch.cuda.empty_cache()
    cfg = O.make_config(**kwargs)
    fwd0 = O.dit_forward
    O.dit_forward = _oracle_on_device(fwd0)
    errs = {}
    try:
        condf = {k: (v.float() if v.is_floating_point() else v) for k, v in cond.items()}
        ref = lat
        with torch.no_grad():
            for i in range(40):
                ref = O.denoise(sd, cfg, ref, condf, steps=40, guidance_scale=4.0, start=i, stop=i + 1)
                if i + 1 in marks:
                    errs[i + 1] = rel_err(ours[i + 1], ref)
    finally:
        O.dit_forward = fwd0
    move = ((ours[40].double() - ref.double()).norm() / (ref.double() - lat.double()).norm()).item()
    _log("denoise_40_steps_full_depth", variant="text+layout" if layout else "text_only", seed=seed, adapter="cached fp32" if cached else
         "per step" if layout else "none", layers=kwargs["num_layers"],
         latents=list(lat.shape), **{f"rel_step{k}": v for k, v in errs.items()}, rel_to_displacement=move,
         finite=bool(torch.isfinite(ours[40]).all()))
    del sd, ref
    torch.cuda.empty_cache()
    tol = TOL_40_STEPS["cached" if cached else layout]
    assert errs[40] < tol and errs[1] < tol, errs


def _heavy_tailed_init_(model, seed: int):
    """A hostile weight regime for the bf16 path (trained SD 3.5 weights are not Gaussian: heavy tails, and a few hidden
    channels that carry activations tens of times larger than the rest): Student-t (4 degrees of freedom) matrices at the
    variance of bench.synth_init_, and 6 outlier channels - the patch embedding and every residual-writing projection
    (attention out-projections, feed-forward second layers) produce them 16 x larger"""
    g = torch.Generator(device="cuda").manual_seed(seed)
    hot = torch.tensor([7, 130, 517, 802, 1111, 1490], device="cuda")
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("mix_factor"):
                p.fill_(2.0)
            elif p.dim() == 1:
                p.normal_(0.0, 0.05, generator=g)
                if name.endswith(".weight"):
                    p.add_(1.0)
            else:
                fan_in = p[0].numel()
                std = fan_in ** -0.5
                if ".norm1.linear" in name or ".norm1_context.linear" in name or name.startswith("norm_out.linear"):
                    std *= 0.5
                z = torch.randn(p.shape, device=p.device, generator=g)
                chi = torch.randn((4,) + tuple(p.shape), device=p.device, generator=g).square().sum(0)
                t = z / (chi / 4).sqrt() / 2 ** 0.5                    # Student-t(4), unit variance
                p.copy_((t * std).to(p.dtype))
                writes_stream = name.endswith(("to_out.0.weight", "to_add_out.weight", "net.2.weight")) or name.startswith("pos_embed.proj.weight")
                if writes_stream and p.shape[0] == 1536 and "condition_image_adapter" not in name:
                    p[hot] *= 16.0


# (bf16 streams, for comparison, are a recorded measurement - profiles/r4b_gpu_parity.log: 7.2e-3 / 1.44e-2 - not a suite case)
@pytest.mark.parametrize("stream", ["fp32"] + (["bf16"] if os.environ.get("DWM_TEST_BF16_STREAMS") else []))
@pytest.mark.parametrize("layout", [False, True], ids=["text_only_rowwise", "text_layout_pointwise"])
def test_forty_step_denoise_heavy_tailed_weights_with_outlier_channels(dev, layout, stream):
    """the 40-step loop in the stress regime of _heavy_tailed_init_, full width, 8 layers, 6 views x 4 frames: the tolerance must
    hold when the hidden state has channels 16 x larger than the rest (LayerNorm statistics dominated by them) and the weights
    have heavy tails"""
    import bench
    from opendwm_amd.pipeline import CTSDDenoiser
    kwargs = bench.variant_kwargs(layout)
    n = 8
    kwargs.update(num_layers=n, dual_attention_layers=list(range(n)), crossview_block_layers=[1, 5], temporal_block_layers=[2, 3, 6, 7])
    model = bench.build_model(kwargs, dev, seed=0)
    model.cache_adapter_residuals = False            # as bench.py times it: the adapter inside every step
    _heavy_tailed_init_(model, 5)
    model.residual_dtype = torch.float32 if stream == "fp32" else bf16       # (fp32 is the default; bf16 streams for comparison)
    wl = dict(bench.WORKLOAD, T=4)
    cond = bench.make_conditions(dev, seed=4, w=wl, layout=layout)
    lat = torch.randn(1, wl["T"], wl["V"], wl["C"], wl["H"], wl["W"], device=dev, generator=torch.Generator(device="cuda").manual_seed(12))
    with torch.no_grad():
        out = CTSDDenoiser(model, guidance_scale=4.0, inference_steps=40).run(lat, cond).clone()
    sd = {k: v.detach().float() for k, v in model.state_dict().items()}
    del model
    torch.cuda.empty_cache()
    cfg = O.make_config(**kwargs)
    fwd0 = O.dit_forward
    O.dit_forward = _oracle_on_device(fwd0)
    try:
        condf = {k: (v.float() if v.is_floating_point() else v) for k, v in cond.items()}
        with torch.no_grad():
            ref = O.denoise(sd, cfg, lat, condf, steps=40, guidance_scale=4.0)
    finally:
        O.dit_forward = fwd0
    e40 = rel_err(out, ref)
    move = ((out.double() - ref.double()).norm() / (ref.double() - lat.double()).norm()).item()
    _log("denoise_40_steps_heavy_tailed", variant="text+layout" if layout else "text_only", streams=stream, layers=n, latents=list(lat.shape),
         rel_step40=e40, rel_to_displacement=move, finite=bool(torch.isfinite(out).all()))
    assert e40 < TOL, e40


@pytest.mark.cost(250, optional=True, priority=1)
def test_tvae_autoregressive_window_full_size_vs_oracle_on_device(dev):
    """BASELINE.json configs[4], one autoregressive window at FULL size (what `bench.py --tvae-ar` runs twice): the 24-layer
    text+layout model on latents [1,5,6,16,32,56] with the previous window's last latent frame injected clean
    (reference_frame_count 1: ctsd.py:1514-1526, 1623-1627), 40 guided FlowMatch-Euler steps, then the CogVideoX temporal VAE
    at its published widths decoding 6 clips x 17 frames x 256x448 in split calls (memory_efficient_batch 2, :1606-1647) -
    CTSDDenoiser + drivers.LatentDecoder against O.denoise + the fp32 VAE oracle, both evaluated on the device.  Checked
    separately: the window's latents, the decode of the SAME (oracle) latents, and the end-to-end frames.
    (Default run: 20 of the 40 steps - the oracle loop on the device costs ~1.3 s per step and frame, the suite has 1200 s; all 40 under
    DWM_HEAVY_TESTS=1, whose result of round 4 is recorded in profiles/r4l_gpu_parity.log.)"""
    import bench
    from oracle import cogvideox_vae_oracle as CV
    from opendwm_amd.drivers import LatentDecoder
    from opendwm_amd.pipeline import CTSDDenoiser
    from opendwm_amd.vae_cogvideox import AutoencoderKLCogVideoX
    c = bench.TVAE_AR
    kwargs = bench.variant_kwargs(True)
    kwargs.update(projection_class_embeddings_input_dim=256 * c["n_time_ids"])
    model = bench.build_model(kwargs, dev, seed=0)
    wl = dict(bench.WORKLOAD, T=5)
    cond = bench.make_conditions(dev, seed=16, w=wl, n_time_ids=c["n_time_ids"], layout=True)
    g = torch.Generator(device="cuda").manual_seed(21)
    noise = torch.randn(1, 5, wl["V"], wl["C"], wl["H"], wl["W"], device=dev, generator=g)
    ref_frame = torch.randn(1, 1, wl["V"], wl["C"], wl["H"], wl["W"], device=dev, generator=g)
    n_steps = c["inference_steps"] if HEAVY else 20
    with torch.no_grad():
        lat = CTSDDenoiser(model, guidance_scale=c["guidance_scale"], inference_steps=n_steps).run(
            noise, cond, image_latents=ref_frame, reference_frame_count=1).clone()
    sd = {k: v.detach().float() for k, v in model.state_dict().items()}
    del model
    torch.cuda.empty_cache()
    cfg = O.make_config(**kwargs)
    fwd0 = O.dit_forward
    O.dit_forward = _oracle_on_device(fwd0)
    try:
        condf = {k: (v.float() if v.is_floating_point() else v) for k, v in cond.items()}
        with torch.no_grad():
            lat_ref = O.denoise(sd, cfg, noise, condf, steps=n_steps, guidance_scale=c["guidance_scale"],
                                image_latents=ref_frame, reference_frame_count=1)
    finally:
        O.dit_forward = fwd0
    del sd
    torch.cuda.empty_cache()
    e_lat = rel_err(lat, lat_ref)
    assert torch.equal(lat[:, :1], ref_frame)                               # the reference frame comes back untouched
    # the temporal VAE at its published widths
    vae = AutoencoderKLCogVideoX().to(dev).to(bf16).eval()
    bench.synth_init_(vae, 1)
    vsd = {k: v.detach().float() for k, v in vae.state_dict().items()}
    vcfg = CV.make_cogvideox_config()
    dec = LatentDecoder(vae, memory_efficient_batch=c["memory_efficient_batch"], postprocess=False)
    with torch.no_grad():
        img_same = dec(lat_ref).float()                                     # HIP decode of the oracle's latents
        img_e2e = dec(lat).float()                                          # HIP decode of the HIP latents
        # the oracle decodes clips independently ("(b v) c t h w"), and the fp32 3-D convolutions of one full-size clip take
        # ~100 s on the device: one of the six views holds the HIP decode to it in the default run, a second one under
        # DWM_HEAVY_TESTS=1 (the two-view result of this round: profiles/r4b_gpu_parity.log)
        views = (1, 4) if HEAVY else (1,)
        ref_imgs = []
        for v in views:
            z = (lat_ref[:, :, v].to(bf16).float() / vcfg["scaling_factor"]).to(bf16).float().permute(0, 2, 1, 3, 4)      # b c t h w
            ref_imgs.append(CV.decode(vsd, vcfg, z))                        # [1, 3, 17, 256, 448]
        ref_img = torch.stack(ref_imgs, 1).permute(0, 3, 1, 2, 4, 5).flatten(0, 2)       # b v c t h w -> (b t v) c h w
    assert img_same.shape == (17 * wl["V"], 3, 8 * wl["H"], 8 * wl["W"])
    pick = lambda im: im.view(17, wl["V"], *im.shape[1:])[:, list(views)].flatten(0, 1)
    img_same, img_e2e = pick(img_same), pick(img_e2e)
    assert img_same.shape == ref_img.shape
    e_dec, e_e2e = rel_err(img_same, ref_img), rel_err(img_e2e, ref_img)
    _log("tvae_ar_window_full_size", latents=list(lat.shape), frames=list(ref_img.shape), rel_latents=e_lat, rel_decode_same_latents=e_dec,
         rel_frames_end_to_end=e_e2e, finite=bool(torch.isfinite(img_e2e).all()))
    assert e_lat < TOL and e_dec < TOL and e_e2e < 2 * TOL, (e_lat, e_dec, e_e2e)


@pytest.mark.cost(50)
def test_unet_full_width_config1_six_frames_vs_oracle_on_device(dev):
    """BASELINE.json configs[1] as `bench.py --unet` runs it: SD 2.1 cross-view temporal UNet at full width (1.92 B
    parameters), 6 views x 6 frames x 32x56 latents, 77 text tokens, ring cross-view mask.  The two halves of the CFG batch
    are independent samples and the fp32 oracle costs ~40 s per sample on the device: one sample in the default run, the
    CFG batch of 2 under DWM_HEAVY_TESTS=1 (9.7e-3, profiles/r3_gpu_parity.log)"""
    nb = 2 if HEAVY else 1
    from oracle import unet_oracle as U
    from opendwm_amd.unet import UNetCrossviewTemporalConditionModel
    cfg = U.make_unet_config()
    sd = {k: v.to(bf16) for k, v in U.make_unet_state_dict(cfg, 0).items()}
    inp = U.make_unet_inputs(cfg, nb, 6, 6, 32, 56, text_len=77)
    inp = {k: (v.to(bf16).float() if v.is_floating_point() and k not in ("timesteps", "added_time_ids") else v) for k, v in inp.items()}
    m = UNetCrossviewTemporalConditionModel(**cfg)
    m.load_state_dict({k: v.float() for k, v in sd.items()}, strict=True)
    m = m.to(dev).to(bf16).eval()
    di = to_dev(inp, dev)
    with torch.no_grad():
        out = m(di.pop("sample"), di.pop("timesteps"), **di)[0][0].float()
    del m
    torch.cuda.empty_cache()
    sd_dev = {k: v.to(dev).float() for k, v in sd.items()}
    with torch.no_grad():
        ref = U.unet_forward(sd_dev, cfg, **to_dev(inp, dev))
    e = rel_err(out, ref)
    _log("unet_full_width_config1", latents=list(out.shape), rel=e, finite=bool(torch.isfinite(out).all()))
    assert out.shape == (nb, 6, 6, 4, 32, 56) and e < TOL


@pytest.mark.parametrize("depth", ["slice3"] + (["full24"] if HEAVY else []))
def test_full_width_train_gradients_vs_oracle_autograd_on_device(dev, depth):
    """BASELINE.json configs[3] geometry at full width on a 3-layer slice (dual joint blocks 0-2, cross-view block after 1,
    temporal block after 2), one sample of 6 views x 4 frames x 32x56 latents: d<prediction, w>/d(parameter) of the HIP
    training path (checkpointed block Functions, hand-written backward kernels, fp32 master weights) against fp32 autograd
    through the oracle on the device.  DWM_HEAVY_TESTS=1 adds the FULL depth (24 joint blocks, 13 of them dual, 6 cross-view and 12
    temporal blocks - the shipped configuration; the oracle's autograd graph takes ~100 GB of the device): result of round 6 in
    profiles/r6_full_depth_train_gradients.log."""
    from opendwm_amd import train
    from opendwm_amd.dit import DiTCrossviewTemporalConditionModel
    if depth == "full24":
        cfg = O.make_config(pos_embed_max_size=64)
    else:
        cfg = O.make_config(num_layers=3, dual_attention_layers=[0, 1, 2], crossview_block_layers=[1], temporal_block_layers=[2],
                            pos_embed_max_size=64)
    gen = torch.Generator().manual_seed(0)
    sd = {n: O.synth_param(n, s, cfg, gen).to(bf16).float() for n, s in O.param_shapes(cfg).items()}
    inp = O.make_inputs(cfg, 1, 4, 6, 32, 56, seed=0)
    inp = {k: (v.to(bf16).float() if v.is_floating_point() and k not in ("timestep", "added_time_ids") else v) for k, v in inp.items()}
    di = to_dev(inp, dev)
    wgt = torch.randn(inp["sample"].shape, generator=torch.Generator().manual_seed(11)).to(dev)

    m = DiTCrossviewTemporalConditionModel(**cfg)
    m.load_state_dict(sd)
    m = m.to(dev).train()
    kw = dict(di)
    out = train.forward_train(m, kw.pop("sample"), kw.pop("timestep"), kw.pop("encoder_hidden_states"), kw.pop("pooled_projections"),
                              crossview_attention_mask=kw.get("crossview_attention_mask"), added_time_ids=kw.get("added_time_ids"))
    (out.float() * wgt).sum().backward()
    ours = {n: p.grad.detach().double().cpu() for n, p in m.named_parameters() if p.grad is not None}
    param_names = {n for n, _ in m.named_parameters()}           # buffers (the sin-cos position table) take no gradient
    out = out.detach().float()
    del m
    torch.cuda.empty_cache()

    sdo = {k: (v.to(dev).clone().requires_grad_(True) if v.is_floating_point() else v.to(dev)) for k, v in sd.items()}
    ref = O.dit_forward(sdo, cfg, **di)
    (ref * wgt).sum().backward()
    e_fwd = rel_err(out, ref.detach())
    errs, num, den, missing = {}, 0.0, 0.0, []
    for name, v in sdo.items():
        if not (torch.is_tensor(v) and v.requires_grad) or v.grad is None or name not in param_names:
            continue
        if name not in ours:
            missing.append(name)
            continue
        a, b = ours[name], v.grad.double().cpu()
        errs[name] = ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
        num += float((a - b).pow(2).sum())
        den += float(b.pow(2).sum())
    glob = (num / den) ** 0.5
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:6]
    mix = {n: v for n, v in errs.items() if n.endswith("mix_factor")}
    _log("full_width_train_gradients", depth=depth, layers=cfg["num_layers"], fwd=e_fwd, global_rel=glob, n_params=len(errs), worst=worst,
         mixers=mix, missing=missing)
    assert not missing, missing
    assert e_fwd < TOL and glob < 3e-2, (e_fwd, glob, worst)
