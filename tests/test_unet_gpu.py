"""GPU leg (`-m gpu`): the SD 2.1 UNet path - kernel extensions it needs and the
model against the fp32 oracle (oracle/unet_oracle.py)."""
import json
import os

import pytest
import torch
import torch.nn.functional as F

from tests.common import rel_err, to_dev

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
bf16 = torch.bfloat16
TOL_KERNEL = 6e-3
TOL_MODEL = 2e-2


def _log(name, **kv):
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "gpu_parity.log"), "a") as f:
        f.write(json.dumps({"test": name, **kv}) + "\n")
    print(name, kv)


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("the gpu-marked tests need a HIP device (torch.cuda.is_available() is False)")
    from opendwm_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _rand(shape, dev, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dev).to(bf16)


@pytest.mark.parametrize("C", [320, 960, 1920, 2560, 64])
def test_groupnorm_general_channel_counts(dev, C):
    """32 groups of 10 / 30 / 60 / 80 / 2... channels: chunks of 8 channels straddle group boundaries"""
    from opendwm_amd import ops
    I, P, G = 3, 77, 32
    if C // G < 4:
        G = 16
    x = _rand((I * P, C), dev, 1, 1.5) + 0.3
    ga, be = _rand((C,), dev, 2, 0.2) + 1, _rand((C,), dev, 3, 0.2)
    y = ops.groupnorm_silu(x, I, P, ga, be, G, 1e-5, silu=True)
    ref = F.silu(F.group_norm(x.float().view(I, P, C).transpose(1, 2), G, ga.float(), be.float(), 1e-5)).transpose(1, 2).reshape(I * P, C)
    e = rel_err(y, ref)
    _log("groupnorm_general", C=C, rel=e)
    assert e < TOL_KERNEL


def test_groupnorm_temporal_map_and_time_grid(dev):
    """GroupNorm over (T, h, w) per (batch, view) on the [(b t v), (h w), C] layout, written into the T-padded grid, then the
    Conv3d (3,1,1) as a 3-tap implicit GEMM with a per-(b,t,v) additive vector (time embedding) in the epilogue."""
    from opendwm_amd import ops
    B, T, V, N, C, Co = 2, 5, 3, 12, 128, 64
    x = _rand((B * T * V * N, C), dev, 1, 1.2)
    ga, be = _rand((C,), dev, 2, 0.2) + 1, _rand((C,), dev, 3, 0.2)
    tg = ops.TimeGrid(B, T, V * N)
    pad = torch.zeros((tg.rows, C), dtype=bf16, device=dev)
    ops.groupnorm_silu(x, B * V, T * N, ga, be, 32, 1e-5, silu=True, out=pad, out_grid=tg,
                       img_map=(V, N, T * V * N, N, V * N))
    x5 = x.float().view(B, T, V, N, C).permute(0, 2, 4, 1, 3).reshape(B * V, C, T, N, 1)      # [(b v), C, T, N, 1]
    ref_n = F.silu(F.group_norm(x5, 32, ga.float(), be.float(), 1e-5))
    got = pad.view(B, T + 2, V, N, C)
    assert torch.count_nonzero(got[:, 0]) == 0 and torch.count_nonzero(got[:, -1]) == 0
    inner = got[:, 1:-1].permute(0, 2, 4, 1, 3).reshape(B * V, C, T, N, 1)
    e1 = rel_err(inner, ref_n)
    w = _rand((Co, C, 3, 1, 1), dev, 4, (3 * C) ** -0.5)
    bias, temb = _rand((Co,), dev, 5, 0.1), _rand((B * T * V, Co), dev, 6)
    wk = w.view(Co, C, 3).permute(0, 2, 1).reshape(Co, 3 * C).contiguous()                   # tap-major K
    y = ops.gemm(pad, wk, bias, a_grid=tg, conv_taps=tg.tap_shifts(), epilogue=ops.EPI_RESID, res=temb, res_mod=-N)
    ref = F.conv3d(ref_n.to(bf16).float(), w.float(), bias.float(), padding=(1, 0, 0))        # [(b v), Co, T, N, 1]
    ref = ref.view(B, V, Co, T, N).permute(0, 3, 1, 4, 2).reshape(B * T * V * N, Co) + temb.float().repeat_interleave(N, 0)
    e2 = rel_err(y, ref)
    _log("temporal_groupnorm_conv", gn=e1, conv=e2)
    assert e1 < TOL_KERNEL and e2 < TOL_KERNEL


def test_conv_stride2_symmetric_padding(dev):
    from opendwm_amd import ops
    I, h, w, C, Co = 2, 8, 12, 64, 128
    x = _rand((I * h * w, C), dev, 1)
    wt, b = _rand((Co, C, 3, 3), dev, 2, (9 * C) ** -0.5), _rand((Co,), dev, 3, 0.1)
    g = ops.PaddedGrid(I, h, w)
    pad = ops.pad_tokens(x, g)
    y = ops.gemm(pad, wt.permute(0, 2, 3, 1).reshape(Co, 9 * C).contiguous(), b, a_grid=g, conv3x3=True, stride2="sym")
    ref = F.conv2d(x.float().view(I, h, w, C).permute(0, 3, 1, 2), wt.float(), b.float(), stride=2, padding=1)
    ref = ref.permute(0, 2, 3, 1).reshape(-1, Co)
    e = rel_err(y, ref)
    _log("conv_stride2_sym", rel=e)
    assert y.shape == ref.shape and e < TOL_KERNEL


@pytest.mark.parametrize("I,Lq,Lk,heads", [(3, 448, 77, 5), (2, 100, 10, 2), (4, 28, 77, 20), (1, 1792, 77, 5)])
def test_cross_attention(dev, I, Lq, Lk, heads):
    from opendwm_amd import ops
    from oracle import ctsd_oracle as O
    D = heads * 64
    q, kv = _rand((I * Lq, D), dev, 1), _rand((I * Lk, 2 * D), dev, 2)
    out = torch.zeros((I * Lq, D), dtype=bf16, device=dev)
    ops.cross_attention(q, kv[:, :D], kv[:, D:], out, I, heads)
    hd = lambda t, L: t.float().view(I, L, heads, 64).transpose(1, 2)
    ref = O.sdpa(hd(q, Lq), hd(kv[:, :D], Lk), hd(kv[:, D:], Lk)).transpose(1, 2).reshape(I * Lq, D)
    e = rel_err(out, ref)
    _log("cross_attention", I=I, Lq=Lq, Lk=Lk, heads=heads, rel=e)
    assert e < TOL_KERNEL


def _small_unet_cfg(**over):
    from oracle import unet_oracle as U
    cfg = U.make_unet_config(block_out_channels=(128, 256, 512, 512), num_attention_heads=(2, 4, 8, 8), cross_attention_dim=128,
                             projection_class_embeddings_input_dim=11 * 256)
    cfg.update(over)
    return cfg


@pytest.mark.parametrize("rowwise", [True, pytest.param(False, marks=pytest.mark.cost(23, optional=True))])
def test_unet_forward_vs_oracle(dev, rowwise):
    """whole SD 2.1 UNet graph (4 levels, cross-attn down / up blocks, mid block, temporal resnets, row-wise or point-wise
    cross-view / temporal transformer blocks, skip concatenations) at small width against the fp32 oracle"""
    from oracle import unet_oracle as U
    from opendwm_amd.unet import UNetCrossviewTemporalConditionModel
    cfg = _small_unet_cfg(enable_rowwise_crossview=rowwise, enable_rowwise_temporal=rowwise)
    sd = {k: v.to(bf16).float() for k, v in U.make_unet_state_dict(cfg, 0).items()}
    inp = U.make_unet_inputs(cfg, 2, 3, 3, 16, 24, text_len=10)
    inp = {k: (v.to(bf16).float() if v.is_floating_point() and k not in ("timesteps", "added_time_ids") else v) for k, v in inp.items()}
    if not rowwise:
        inp["crossview_attention_mask"] = None          # the reference's un-expanded mask only fits T == 1 there
    ref = U.unet_forward(sd, cfg, **inp)
    m = UNetCrossviewTemporalConditionModel(**cfg)
    missing, unexpected = m.load_state_dict(sd, strict=True), None
    m = m.to(dev).to(bf16).eval()
    di = to_dev(inp, dev)
    out = m(di.pop("sample"), di.pop("timesteps"), **di)
    assert isinstance(out, tuple) and out[0][0].shape == ref.shape and out[0][0].dtype == bf16
    e = rel_err(out[0][0], ref)
    _log("unet_forward", rowwise=rowwise, rel=e)
    assert e < TOL_MODEL
    # flags: disable_temporal switches both the resnet and the transformer time mixers to alpha = 1
    inp2 = dict(inp, disable_temporal=torch.tensor([True, False]), disable_crossview=torch.tensor([False, True]))
    ref2 = U.unet_forward(sd, cfg, **inp2)
    di = to_dev(inp2, dev)
    out2 = m(di.pop("sample"), di.pop("timesteps"), **di)
    e2 = rel_err(out2[0][0], ref2)
    _log("unet_forward_flags", rowwise=rowwise, rel=e2)
    assert e2 < TOL_MODEL and rel_err(ref2, ref) > 1e-2


def test_unet_denoise_loop_vs_oracle(dev):
    """CFG + DPM-Solver++(2M) loop (first-order first step, second-order steps, x0 on the last) against the oracle"""
    from oracle import unet_oracle as U
    from opendwm_amd.pipeline import UNetDenoiser, dpm_solver_coefficients, dpm_solver_tables
    from opendwm_amd.unet import UNetCrossviewTemporalConditionModel
    ts, sg = dpm_solver_tables(10)
    ts_o, sg_o = U.dpm_solver_tables(10)
    assert torch.equal(ts, ts_o) and torch.equal(sg, sg_o)
    for i in (0, 3, 9):
        for pt in ("epsilon", "v_prediction"):
            assert dpm_solver_coefficients(sg, i, pt) == pytest.approx(U.dpm_solver_coefficients(sg_o, i, pt))
    cfg = _small_unet_cfg()
    sd = {k: v.to(bf16).float() for k, v in U.make_unet_state_dict(cfg, 0).items()}
    inp = U.make_unet_inputs(cfg, 2, 2, 3, 16, 24, text_len=10)
    inp = {k: (v.to(bf16).float() if v.is_floating_point() and k not in ("timesteps", "added_time_ids") else v) for k, v in inp.items()}
    lat = inp.pop("sample")[:1]
    inp.pop("timesteps")
    steps = 4
    ref = U.unet_denoise(sd, cfg, lat, inp, steps, 3.0)
    m = UNetCrossviewTemporalConditionModel(**cfg)
    m.load_state_dict(sd)
    m = m.to(dev).to(bf16).eval()
    out = UNetDenoiser(m, 3.0, steps).run(lat.to(dev), to_dev(inp, dev))
    e = rel_err(out, ref)
    _log("unet_denoise_loop", steps=steps, rel=e)
    assert e < 3e-2


@pytest.mark.cost(28, optional=True)
def test_unet_ddim_denoise_loop_vs_oracle(dev):
    """the reference's DEFAULT test scheduler of the UNet (ctsd.py:969-974: DDIMScheduler when inference_config names none):
    UNetDenoiser(scheduler=DDIMScheduler()) - CFG + tensor-timestep DDIM step in one kernel - against the oracle UNet driven
    by oracle/scheduler_oracle.ddim_step (pinned by the executed reference step)"""
    from oracle import scheduler_oracle as SO
    from oracle import unet_oracle as U
    from opendwm_amd.pipeline import UNetDenoiser
    from opendwm_amd.schedulers import DDIMScheduler
    from opendwm_amd.unet import UNetCrossviewTemporalConditionModel
    cfg = _small_unet_cfg()
    sd = {k: v.to(bf16).float() for k, v in U.make_unet_state_dict(cfg, 0).items()}
    inp = U.make_unet_inputs(cfg, 2, 2, 3, 16, 24, text_len=10)
    inp = {k: (v.to(bf16).float() if v.is_floating_point() and k not in ("timesteps", "added_time_ids") else v) for k, v in inp.items()}
    lat = inp.pop("sample")[:1]
    inp.pop("timesteps")
    steps = 4
    sch = DDIMScheduler()
    sch.set_timesteps(steps)
    x = lat.float().clone()
    B, T, V = x.shape[:3]
    for i in range(steps):
        t = sch.timesteps[i]
        out = U.unet_forward(sd, cfg, torch.cat([x, x]), t.float().expand(2 * B, T, V), **inp)
        u, c = out.chunk(2)
        x, _ = SO.ddim_step(sch.alphas_cumprod, sch.final_alpha_cumprod, 1000, steps, "v_prediction", u + 3.0 * (c - u),
                            t.expand(B, T, V), x)
    m = UNetCrossviewTemporalConditionModel(**cfg)
    m.load_state_dict(sd)
    m = m.to(dev).to(bf16).eval()
    got = UNetDenoiser(m, 3.0, steps, scheduler=DDIMScheduler()).run(lat.to(dev), to_dev(inp, dev))
    e = rel_err(got, x)
    _log("unet_ddim_denoise_loop", steps=steps, rel=e)
    assert e < 3e-2


def test_unet_matches_golden_fixture(dev):
    """the committed oracle fixture (tests/golden/unet_small.pt) through the HIP path"""
    from oracle import unet_oracle as U
    from opendwm_amd.unet import UNetCrossviewTemporalConditionModel
    from tests.common import GOLDEN
    from tests.golden.make_golden import unet_small_config
    cfg = unet_small_config()
    sd = U.make_unet_state_dict(cfg, 0)
    inp = U.make_unet_inputs(cfg, 2, 2, 3, 8, 16, text_len=10)
    gold = torch.load(os.path.join(GOLDEN, "unet_small.pt"))
    m = UNetCrossviewTemporalConditionModel(**cfg)
    m.load_state_dict(sd)
    m = m.to(dev).to(bf16).eval()
    di = to_dev(inp, dev)
    out = m(di.pop("sample"), di.pop("timesteps"), **di)[0][0]
    e = rel_err(out, gold["output"])
    _log("unet_vs_golden", rel=e)
    assert e < TOL_MODEL


def test_unet_with_layout_adapter_vs_oracle(dev):
    """condition_image_tensor -> ImageAdapter -> residuals after conv_in and after every down block
    (crossview_temporal_unet.py:717-755): forward against the oracle, and the step-invariant adapter cache"""
    from oracle import unet_oracle as U
    from opendwm_amd.unet import UNetCrossviewTemporalConditionModel
    acfg = dict(in_channels=3, channels=[128, 128, 256, 512, 512], is_downblocks=[False, True, True, True, False],
                num_res_blocks=1, downscale_factor=8, use_zero_convs=True)
    cfg = _small_unet_cfg(condition_image_adapter_config=acfg)
    sd = {k: v.to(bf16).float() for k, v in U.make_unet_state_dict(cfg, 0).items()}
    inp = U.make_unet_inputs(cfg, 2, 2, 3, 8, 16, text_len=10)
    inp["condition_image_tensor"] = torch.rand(2, 2, 3, 3, 64, 128, generator=torch.Generator().manual_seed(7))
    inp = {k: (v.to(bf16).float() if v.is_floating_point() and k not in ("timesteps", "added_time_ids") else v) for k, v in inp.items()}
    ref = U.unet_forward(sd, cfg, **inp)
    plain = U.unet_forward(sd, cfg, **{k: v for k, v in inp.items() if k != "condition_image_tensor"})
    m = UNetCrossviewTemporalConditionModel(**cfg)
    m.load_state_dict(sd, strict=True)
    m = m.to(dev).to(bf16).eval()
    di = to_dev(inp, dev)
    out = m(di.pop("sample"), di.pop("timesteps"), **di)[0][0]
    e = rel_err(out, ref)
    _log("unet_layout_adapter", rel=e, effect=rel_err(ref, plain))
    assert e < TOL_MODEL and rel_err(ref, plain) > 5e-2
    kw = dict(di)
    again = m(to_dev(inp, dev)["sample"], to_dev(inp, dev)["timesteps"], **kw)[0][0]      # same tensor object: cached residuals
    assert torch.equal(again, out) and m._adapter_cache[0] is not None          # also: GroupNorm statistics are order-fixed


@pytest.mark.parametrize("name", ["rowwise", "pointwise"])
def test_unet_forward_vs_reference_forward_fixture(dev, name):
    """against the vector produced by the REFERENCE's own UNet composition (tests/golden/reference_unet_forward.pt)"""
    from oracle import unet_oracle as U
    from opendwm_amd.unet import UNetCrossviewTemporalConditionModel
    from tests.golden.make_golden import unet_small_config
    fxu = torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_unet_forward.pt"))[name]
    cfg = dict(unet_small_config(), **fxu["over"])
    sd = {k: v.to(bf16).float() for k, v in U.make_unet_state_dict(cfg, 0).items()}
    inp = U.make_unet_inputs(cfg, 2, 2, 3, 8, 16, text_len=10)
    inp["disable_crossview"], inp["disable_temporal"] = fxu["flags"]
    if name == "pointwise":
        inp["crossview_attention_mask"] = None
    m = UNetCrossviewTemporalConditionModel(**cfg)
    m.load_state_dict(sd, strict=True)
    m = m.to(dev).to(bf16).eval()
    di = to_dev(inp, dev)
    out = m(di.pop("sample"), di.pop("timesteps"), **di)[0][0]
    e = rel_err(out, fxu["output"])
    _log("unet_reference_forward_fixture", case=name, rel=e)
    assert e < TOL_MODEL


def test_unet_single_frame_six_views(dev):
    """BASELINE.json configs[0] geometry (examples/ctsd_21_6views_image_generation.json: one frame, six views, the
    temporal modules still built): T = 1 makes every temporal attention / Conv3d (3,1,1) see a single frame"""
    from oracle import unet_oracle as U
    from opendwm_amd.unet import UNetCrossviewTemporalConditionModel
    cfg = _small_unet_cfg()
    sd = {k: v.to(bf16).float() for k, v in U.make_unet_state_dict(cfg, 0).items()}
    inp = U.make_unet_inputs(cfg, 2, 1, 6, 8, 8, text_len=10)
    inp = {k: (v.to(bf16).float() if v.is_floating_point() and k not in ("timesteps", "added_time_ids") else v) for k, v in inp.items()}
    ref = U.unet_forward(sd, cfg, **inp)
    m = UNetCrossviewTemporalConditionModel(**cfg)
    m.load_state_dict(sd, strict=True)
    m = m.to(dev).to(bf16).eval()
    di = to_dev(inp, dev)
    out = m(di.pop("sample"), di.pop("timesteps"), **di)[0][0]
    e = rel_err(out, ref)
    _log("unet_single_frame", rel=e)
    assert out.shape == (2, 1, 6, 4, 8, 8) and e < TOL_MODEL


def test_unet_full_width_config0_vs_oracle_on_device(dev):
    """BASELINE.json configs[0] at FULL width (SD 2.1: 320 / 640 / 1280 / 1280 channels, 5 / 10 / 20 / 20 heads, 1.92 B
    parameters), six views x one frame x 256x256 px (latents [1,1,6,4,32,32], CFG batch 2, 77 text tokens): the HIP bf16
    forward against the fp32 oracle evaluated on the same device (the CPU oracle needs minutes for this)."""
    from oracle import unet_oracle as U
    from opendwm_amd.unet import UNetCrossviewTemporalConditionModel
    cfg = U.make_unet_config()
    sd = {k: v.to(bf16) for k, v in U.make_unet_state_dict(cfg, 0).items()}
    inp = U.make_unet_inputs(cfg, 2, 1, 6, 32, 32, text_len=77)
    inp = {k: (v.to(bf16).float() if v.is_floating_point() and k not in ("timesteps", "added_time_ids") else v) for k, v in inp.items()}
    m = UNetCrossviewTemporalConditionModel(**cfg)
    m.load_state_dict({k: v.float() for k, v in sd.items()}, strict=True)
    m = m.to(dev).to(bf16).eval()
    di = to_dev(inp, dev)
    out = m(di.pop("sample"), di.pop("timesteps"), **di)[0][0]
    del m
    torch.cuda.empty_cache()
    sd_dev = {k: v.to(dev).float() for k, v in sd.items()}
    with torch.no_grad():
        ref = U.unet_forward(sd_dev, cfg, **to_dev(inp, dev))
    e = rel_err(out, ref)
    _log("unet_full_width_config0", rel=e, finite=bool(torch.isfinite(out.float()).all()))
    assert out.shape == (2, 1, 6, 4, 32, 32) and e < TOL_MODEL
