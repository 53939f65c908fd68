"""GPU leg (`-m gpu`): the SD 2.1 UNet training branch (src/dwm/pipelines/ctsd.py:1240-1253; opendwm_amd.train_unet) - the
new backward kernels against fp32 torch autograd of the same op, the block Functions and the whole UNet against fp32
autograd through the oracle (oracle/unet_oracle.py), and the trainer's SD 2.1 loss / descent."""
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
TOL_REDUCE = 2e-3


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


# ------------------------------------------------------------------------------------------ kernels
@pytest.mark.parametrize("C,silu,mode", [(320, True, "plain"), (128, True, "padded"), (640, False, "plain"), (256, True, "temporal"),
                                         (1920, True, "plain")])
def test_groupnorm_bwd(dev, C, silu, mode):
    """dwm_groupnorm_bwd vs autograd of F.group_norm (+ SiLU): compact rows, dz read from the padded grid the forward wrote,
    and the (b v) x (t h w) row map of TemporalResnetBlock; channels per group 4 / 8 / 10 / 20 / 60"""
    from opendwm_amd import ops
    from opendwm_amd import train_ops as T
    from opendwm_amd.ops import PaddedGrid
    B, Tn, V, h, w = 2, 3, 2, 6, 8
    I, N = B * Tn * V, h * w
    x = _rand((I * N, C), dev, 1)
    dz = _rand((I * N, C), dev, 2)
    gamma, beta = (_rand((C,), dev, 3) * 0.2 + 1), _rand((C,), dev, 4, 0.2)
    xr = x.float().requires_grad_(True)
    gr, br = gamma.float().requires_grad_(True), beta.float().requires_grad_(True)
    if mode == "temporal":
        # image = (b, v), pixels = (t, h, w): rows of the [(b t v), (h w), C] layout
        x5 = xr.view(B, Tn, V, N, C).permute(0, 2, 4, 1, 3).reshape(B * V, C, Tn, N)
        y = F.group_norm(x5, 32, gr, br, 1e-5)
        y = F.silu(y) if silu else y
        y = y.view(B, V, C, Tn, N).permute(0, 3, 1, 4, 2).reshape(I * N, C)
        kw = dict(img_map=(V, N, Tn * V * N, N, V * N))
        Ii, Pp = B * V, Tn * N
    else:
        y = F.group_norm(xr.view(I, N, C).transpose(1, 2), 32, gr, br, 1e-5)
        y = (F.silu(y) if silu else y).transpose(1, 2).reshape(I * N, C)
        kw = {}
        Ii, Pp = I, N
    (y * dz.float()).sum().backward()
    dg = torch.zeros(C, dtype=torch.float32, device=dev)
    db = torch.zeros(C, dtype=torch.float32, device=dev)
    if mode == "padded":
        grid = PaddedGrid(I, h, w)
        dzp = ops.pad_tokens(dz, grid)
        dx = T.groupnorm_bwd(x, dzp, Ii, Pp, gamma, beta, 32, 1e-5, dg, db, silu=silu, dz_grid=grid)
    else:
        dx = T.groupnorm_bwd(x, dz, Ii, Pp, gamma, beta, 32, 1e-5, dg, db, silu=silu, **kw)
    e = dict(dx=rel_err(dx, xr.grad), dgamma=rel_err(dg, gr.grad), dbeta=rel_err(db, br.grad))
    # accumulate form
    base = _rand((I * N, C), dev, 9)
    dx2 = base.clone()
    T.groupnorm_bwd(x, dz if mode != "padded" else dzp, Ii, Pp, gamma, beta, 32, 1e-5, torch.zeros_like(dg), torch.zeros_like(db), silu=silu,
                    dx=dx2, accumulate=True, dz_grid=grid if mode == "padded" else None, **kw)
    e["acc"] = rel_err(dx2, base.float() + xr.grad)
    _log("groupnorm_bwd", C=C, silu=silu, mode=mode, **e)
    assert e["dx"] < TOL_KERNEL and e["acc"] < TOL_KERNEL and e["dgamma"] < TOL_REDUCE and e["dbeta"] < TOL_REDUCE, e


@pytest.mark.parametrize("I,Lq,Lk,heads", [(3, 448, 77, 5), (2, 100, 10, 2), (4, 28, 77, 4), (1, 300, 130, 3)])
def test_cross_attention_bwd(dev, I, Lq, Lk, heads):
    """dwm_attention_bwd in cross mode (queries = the sample tokens, keys / values = the text tokens) vs fp32 autograd"""
    from opendwm_amd import ops
    from oracle import ctsd_oracle as O
    D = heads * 64
    q, kv = _rand((I * Lq, D), dev, 1), _rand((I * Lk, 2 * D), dev, 2)
    dout = _rand((I * Lq, D), dev, 3)
    out = torch.zeros((I * Lq, D), dtype=bf16, device=dev)
    lse = torch.zeros(I * heads * (Lq + Lk), dtype=torch.float32, device=dev)
    ops.cross_attention(q, kv[:, :D], kv[:, D:], out, I, heads, lse=lse)
    dq, dkv = torch.zeros_like(q), torch.zeros_like(kv)
    ops.cross_attention_bwd(q, kv[:, :D], kv[:, D:], out, dout, dq, dkv[:, :D], dkv[:, D:], I, heads, lse)
    qr, kvr = q.float().requires_grad_(True), kv.float().requires_grad_(True)
    hd = lambda t, L: t.view(I, L, heads, 64).transpose(1, 2)
    ref = O.sdpa(hd(qr, Lq), hd(kvr[:, :D], Lk), hd(kvr[:, D:], Lk)).transpose(1, 2).reshape(I * Lq, D)
    (ref * dout.float()).sum().backward()
    e = dict(fwd=rel_err(out, ref.detach()), dq=rel_err(dq, qr.grad), dk=rel_err(dkv[:, :D], kvr.grad[:, :D]),
             dv=rel_err(dkv[:, D:], kvr.grad[:, D:]))
    _log("cross_attention_bwd", I=I, Lq=Lq, Lk=Lk, heads=heads, **e)
    assert all(v < TOL_KERNEL for v in e.values()), e


# ------------------------------------------------------------------------------------------ model level
def _small_unet_cfg(**over):
    from oracle import unet_oracle as U
    cfg = U.make_unet_config(block_out_channels=(128, 256, 512, 512), num_attention_heads=(2, 4, 8, 8), cross_attention_dim=128,
                             projection_class_embeddings_input_dim=11 * 256)
    cfg.update(over)
    return cfg


def _unet_model(cfg, sd, dev):
    from opendwm_amd.unet import UNetCrossviewTemporalConditionModel
    m = UNetCrossviewTemporalConditionModel(**cfg)
    m.load_state_dict(sd, strict=True)
    return m.to(dev).train()            # fp32 master parameters, bf16 shadows on first use


def _oracle_unet_grads(sd, cfg, inp, wgt, dev, mixer_cond=None):
    from oracle import unet_oracle as U
    sdo = {k: (v.to(dev).clone().requires_grad_(True) if v.is_floating_point() else v.to(dev)) for k, v in sd.items()}
    blend0, recs = U.alpha_blender, []

    def blender(sd_, p, a, b, image_only):
        out = blend0(sd_, p, a, b, image_only)
        rec = dict(p=p, d=(a - b).detach(), on=(~image_only).detach())
        out.register_hook(lambda gr, rec=rec: rec.__setitem__("dy", gr.detach()))
        recs.append(rec)
        return out
    U.alpha_blender = blender
    try:
        ref = U.unet_forward(sdo, cfg, **inp)
        (ref * wgt).sum().backward()
    finally:
        U.alpha_blender = blend0
    if mixer_cond is not None:
        for rec in recs:
            t = (rec["dy"] * rec["d"]).double()
            t = t * rec["on"].view(-1, *([1] * (t.dim() - 1))).to(t.dtype)
            mixer_cond[rec["p"] + ".mix_factor"] = (t.abs().sum() / t.sum().abs().clamp_min(1e-300)).item()
    return ref.detach(), {k: v.grad for k, v in sdo.items() if torch.is_tensor(v) and v.requires_grad}


@pytest.mark.parametrize("rowwise", [True, pytest.param(False, marks=pytest.mark.cost(24, optional=True))])
def test_unet_gradients_vs_oracle_autograd(dev, rowwise):
    """d(loss)/d(every parameter) of the UNet training path (checkpointed block Functions, HIP backward kernels, bf16 compute
    on fp32 masters) against fp32 autograd through the oracle on the small full-graph configuration (4 levels, cross-attn
    down / up blocks, mid block, temporal resnets + mixers, row-wise or point-wise cross-view / temporal blocks, stride-2
    and nearest-2x samplers, skip concatenations); loss = <prediction, fixed random tensor>."""
    from oracle import unet_oracle as U
    cfg = _small_unet_cfg(enable_rowwise_crossview=rowwise, enable_rowwise_temporal=rowwise)
    sd = {k: v.to(bf16).float() for k, v in U.make_unet_state_dict(cfg, 0).items()}
    inp = U.make_unet_inputs(cfg, 2, 2, 3, 8, 16, text_len=10)
    inp = {k: (v.to(bf16).float() if v.is_floating_point() and k not in ("timesteps", "added_time_ids") else v) for k, v in inp.items()}
    if not rowwise:
        inp["crossview_attention_mask"] = None
    di = to_dev(inp, dev)
    wgt = torch.randn(2, 2, 3, cfg["out_channels"], 8, 16, generator=torch.Generator().manual_seed(11)).to(dev)
    cond = {}
    ref, gref = _oracle_unet_grads(sd, cfg, di, wgt, dev, mixer_cond=cond)
    m = _unet_model(cfg, sd, dev)
    kw = dict(di)
    out = m(kw.pop("sample"), kw.pop("timesteps"), **kw)[0][0]           # the reference entry point, train mode
    assert out.grad_fn is not None
    e_fwd = rel_err(out, ref)
    from tests.common import capture_segsum_diff, check_mixer_gradients
    with capture_segsum_diff() as seg_calls:
        (out.float() * wgt).sum().backward()
    errs, num, den, missing = {}, 0.0, 0.0, []
    for name, p in m.named_parameters():
        if name not in gref or gref[name] is None:
            continue
        if p.grad is None:
            missing.append(name)
            continue
        a, b = p.grad.double().cpu(), gref[name].double().cpu()
        errs[name] = ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
        num += float((a - b).pow(2).sum())
        den += float(b.pow(2).sum())
    glob = (num / den) ** 0.5
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:8]
    mixers = {n: dict(rel=errs[n], cond=cond[n]) for n in cond if n in errs}
    # scalar mixer parameters: d(alpha) = <dy, x_spatial - x_mixed> is ONE heavily cancelling sum (sum|terms| / |sum| up to 6000
    # here): the kernel against the fp64 sum of its own inputs at any conditioning, the gradient against a flat 8 % where the
    # sum is reasonably conditioned (dy arrives through a deeper bf16 backward than in the MMDiT: ~1e-2 relative error by then)
    wk = check_mixer_gradients(mixers, seg_calls, 8e-2)
    _log("unet_gradients", rowwise=rowwise, fwd=e_fwd, global_rel=glob, worst=worst, n_params=len(errs), missing=missing, mixers=mixers,
         segsum_kernel_vs_own_inputs=wk)
    assert not missing, missing
    assert e_fwd < 2e-2 and glob < 3e-2, (glob, worst)
    assert all(v < 0.15 for n, v in errs.items() if n not in cond), worst


def test_unet_trainer_sd21_branch_loss_and_descent(dev):
    """CTSDTrainer with the UNet = the SD 2.1 branch of train_step (ctsd.py:1240-1253): DDPM add_noise with per-sample integer
    timesteps, v_prediction target, mse on the raw prediction - loss against the oracle (fp32, same noise / timesteps),
    and four optimizer steps on one batch bring it down."""
    from oracle import unet_oracle as U
    from opendwm_amd.pipeline import CTSDTrainer
    from opendwm_amd.schedulers import DDPMScheduler
    cfg = _small_unet_cfg()
    sd = {k: v.to(bf16).float() for k, v in U.make_unet_state_dict(cfg, 0).items()}
    inp = U.make_unet_inputs(cfg, 2, 2, 3, 8, 16, text_len=10)
    cond = {k: v for k, v in inp.items() if k not in ("sample", "timesteps")}
    g = torch.Generator().manual_seed(3)
    latents = torch.randn(2, 2, 3, 4, 8, 16, generator=g)
    m = _unet_model(cfg, sd, dev)
    tr = CTSDTrainer(m, lr=2e-4, weight_decay=0.0)
    assert tr.is_unet and isinstance(tr.train_scheduler, DDPMScheduler)
    if cfg["in_channels"] != 4:
        pytest.skip("the small UNet takes 4 latent channels")
    noise, ts, _ = tr.draw_training_inputs(latents.shape, torch.Generator().manual_seed(5))
    assert ts.dtype == torch.int64 and ts.shape == (2,)
    dcond = to_dev(cond, dev)
    loss0 = tr.loss(latents.to(dev), dcond, timestep_indices=ts, noise=noise)
    # oracle: same pair through the reference formulas in fp32
    acp = tr.train_scheduler.alphas_cumprod.cpu()[ts].view(2, 1, 1, 1, 1, 1)
    noisy = acp.sqrt() * latents + (1 - acp).sqrt() * noise
    target = acp.sqrt() * noise - (1 - acp).sqrt() * latents
    tfull = ts.view(2, 1, 1).expand(2, 2, 3).float()
    ref = U.unet_forward(sd, cfg, sample=noisy.to(bf16).float(), timesteps=tfull, **{k: v for k, v in cond.items()})
    want = F.mse_loss(ref, target).item()
    _log("unet_train_loss", ours=loss0.item(), oracle=want)
    assert abs(loss0.item() - want) / want < 2e-2
    losses = []
    for _ in range(4):
        losses.append(tr.train_step(latents.to(dev), dcond, timestep_indices=ts, noise=noise).item())
    _log("unet_train_descent", losses=losses)
    assert losses[-1] < losses[0] * 0.98, losses


def test_unet_gradients_with_layout_adapter(dev):
    """the SD 2.1 training configs ship the layout ImageAdapter (channels [c0, c0, c1, c2, c3], AvgPool2d in blocks 1-3,
    zero convs; configs/ctsd/multi_datasets/ctsd_21_tirda_bm_nwa.json): its residuals after conv_in and every down block in
    train mode, gradients of every adapter parameter (incl. the pooling levels) and of the UNet against fp32 autograd"""
    from oracle import unet_oracle as U
    acfg = dict(in_channels=3, channels=[128, 128, 256, 512, 512], is_downblocks=[False, True, True, True, False],
                num_res_blocks=1, downscale_factor=8, use_zero_convs=True)
    cfg = _small_unet_cfg(condition_image_adapter_config=acfg)
    sd = {k: v.to(bf16).float() for k, v in U.make_unet_state_dict(cfg, 0).items()}
    inp = U.make_unet_inputs(cfg, 2, 2, 3, 8, 16, text_len=10)
    inp["condition_image_tensor"] = torch.rand(2, 2, 3, 3, 64, 128, generator=torch.Generator().manual_seed(7))
    inp = {k: (v.to(bf16).float() if v.is_floating_point() and k not in ("timesteps", "added_time_ids") else v) for k, v in inp.items()}
    di = to_dev(inp, dev)
    wgt = torch.randn(2, 2, 3, cfg["out_channels"], 8, 16, generator=torch.Generator().manual_seed(11)).to(dev)
    ref, gref = _oracle_unet_grads(sd, cfg, di, wgt, dev)
    m = _unet_model(cfg, sd, dev)
    kw = dict(di)
    out = m(kw.pop("sample"), kw.pop("timesteps"), **kw)[0][0]
    assert out.grad_fn is not None
    e_fwd = rel_err(out, ref)
    (out.float() * wgt).sum().backward()
    errs, num, den, missing = {}, 0.0, 0.0, []
    for name, p in m.named_parameters():
        if not name.startswith("condition_image_adapter") or name not in gref or gref[name] is None:
            continue
        if p.grad is None:
            missing.append(name)
            continue
        a, b = p.grad.double().cpu(), gref[name].double().cpu()
        errs[name] = ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
        num += float((a - b).pow(2).sum())
        den += float(b.pow(2).sum())
    glob = (num / den) ** 0.5
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    _log("unet_adapter_gradients", fwd=e_fwd, adapter_global_rel=glob, worst=worst, n_adapter_params=len(errs), missing=missing)
    assert not missing, missing
    assert len(errs) >= 5 * 6                      # five blocks: in_conv / block1 / block2 / zero conv, weight + bias
    assert e_fwd < 2e-2 and glob < 4e-2, (glob, worst)
    assert all(v < 0.15 for v in errs.values()), worst
