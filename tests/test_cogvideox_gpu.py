"""AutoencoderKLCogVideoX (temporal VAE of the tvae configs, ctsd.py:953-964) on the GPU against the fp32 CPU oracle
(oracle/cogvideox_vae_oracle.py).  Tolerances: kernels TOL_KERNEL = 4e-3 (one bf16 rounding of fp32-accumulated results),
whole encoder / decoder TOL_MODEL = 2e-2 relative Frobenius error (BASELINE.json's bf16 bound)."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cogvideox_vae_oracle as CV      # noqa: E402
from tests.common import rel_err                    # noqa: E402

pytestmark = pytest.mark.gpu
bf16 = torch.bfloat16
TOL_KERNEL = 4e-3
TOL_MODEL = 2e-2


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("the gpu-marked tests need a HIP device (torch.cuda.is_available() is False)")
    from opendwm_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _rand(shape, dev, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dev).to(bf16)


def _log(name, **kw):
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/gpu_parity.log", "a") as f:
        f.write(name + " " + " ".join(f"{k}={v:.3e}" if isinstance(v, float) else f"{k}={v}" for k, v in kw.items()) + "\n")


def small_cfg(**over):
    cfg = CV.make_cogvideox_config(block_out_channels=(64, 64, 128, 128), layers_per_block=1, norm_num_groups=8)
    cfg.update(over)
    return cfg


def _bf_sd(sd):
    return {k: v.to(bf16).float() for k, v in sd.items()}


def _model(cfg, sd, dev):
    from opendwm_amd.vae_cogvideox import AutoencoderKLCogVideoX
    keep = ("in_channels", "out_channels", "block_out_channels", "latent_channels", "layers_per_block", "norm_eps",
            "norm_num_groups", "temporal_compression_ratio", "scaling_factor", "shift_factor")
    m = AutoencoderKLCogVideoX(**{k: cfg[k] for k in keep})
    missing, unexpected = m.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    return m.to(dev).to(bf16).eval()


def test_frame_mix(dev):
    from opendwm_amd import ops
    x = _rand((5, 6, 40), dev, 1)                       # 5 frames of 240 elements
    out = ops.frame_mix(x, 240, [0, 1, 3], [0, 2, 4], [1.0, 0.5, 0.5], [0.0, 0.5, 0.5]).view(3, 6, 40)
    xf = x.float()
    ref = torch.stack([xf[0], 0.5 * (xf[1] + xf[2]), 0.5 * (xf[3] + xf[4])])
    assert rel_err(out, ref) < TOL_KERNEL
    rep = ops.frame_mix(x, 240, [0, 1, 1, 2, 2], [0] * 5, [1.0] * 5, [0.0] * 5).view(5, 6, 40)
    assert torch.equal(rep, x[[0, 1, 1, 2, 2]])
    with pytest.raises(RuntimeError):
        ops.frame_mix(x, 240, [5], [0], [1.0], [0.0])


@pytest.mark.parametrize("T,B,h,w,C,N", [(3, 2, 4, 6, 64, 64), (2, 1, 5, 3, 128, 192), (1, 3, 4, 4, 64, 8)])
def test_causal_conv3d_27_taps(dev, T, B, h, w, C, N):
    """27-tap implicit GEMM over ops.Grid3D == CogVideoXCausalConv3d: first call replicates frame 0 twice, the second
    call continues from the first call's last two input frames."""
    from opendwm_amd import ops
    x1, x2 = _rand((B, C, T, h, w), dev, 1), _rand((B, C, T, h, w), dev, 2)
    wt, b = _rand((N, C, 3, 3, 3), dev, 3, (27 * C) ** -0.5), _rand((N,), dev, 4)
    sd = {"c.conv.weight": wt.float().cpu(), "c.conv.bias": b.float().cpu()}
    cache = CV.ConvCache()
    refs = [CV.causal_conv3d(sd, "c", x.float().cpu(), cache) for x in (x1, x2)]
    grid = ops.Grid3D(T, B, h, w)
    fr = grid.frame_rows
    wp = wt.permute(0, 2, 3, 4, 1).reshape(N, 27 * C).contiguous()
    prev, errs = None, []
    for x, ref in zip((x1, x2), refs):
        tok = x.permute(2, 0, 3, 4, 1).reshape(-1, C).contiguous()          # (t, b, y, x) rows
        buf = ops.pad_tokens(tok, grid)
        if prev is None:
            buf[:fr].copy_(buf[2 * fr:3 * fr])
            buf[fr:2 * fr].copy_(buf[2 * fr:3 * fr])
        else:
            buf[:2 * fr].copy_(prev)
        prev = buf[T * fr:(T + 2) * fr].clone()
        out = ops.gemm(buf, wp, b, a_grid=grid, conv_taps=grid.tap_shifts())
        got = out.reshape(T, B, h, w, N).permute(1, 4, 0, 2, 3)
        errs.append(rel_err(got, ref))
    _log("causal_conv3d", T=T, B=B, h=h, w=w, C=C, N=N, rel_first=errs[0], rel_cached=errs[1])
    assert max(errs) < TOL_KERNEL


@pytest.mark.parametrize("Tz,T,shift", [(3, 3, 0), (3, 5, 1), (3, 9, 2), (2, 8, 2), (1, 1, 1), (2, 2, 0)])
def test_spatial_norm3d(dev, Tz, T, shift):
    """dwm_groupnorm_spatial == SiLU(CogVideoXSpatialNorm3D(f, zq)) incl. the separate first frame of odd clips"""
    from opendwm_amd import ops
    B, hz, wz, C, zc, G = 2, 3, 4, 64, 16, 8
    h, w = hz << shift, wz << shift
    f, zq = _rand((B, C, T, h, w), dev, 1), _rand((B, zc, Tz, hz, wz), dev, 2)
    sd = {"n.norm_layer.weight": 1 + 0.1 * torch.randn(C), "n.norm_layer.bias": 0.1 * torch.randn(C),
          "n.conv_y.conv.weight": torch.randn(C, zc, 1, 1, 1) * 0.2, "n.conv_y.conv.bias": 1 + 0.1 * torch.randn(C),
          "n.conv_b.conv.weight": torch.randn(C, zc, 1, 1, 1) * 0.2, "n.conv_b.conv.bias": 0.1 * torch.randn(C)}
    sd = _bf_sd(sd)
    ref = F.silu(CV.norm3d(sd, "n", f.float().cpu(), zq.float().cpu(), G, 1e-6, CV.ConvCache()))
    zrows = torch.zeros((Tz * B * hz * wz, 64), dtype=bf16, device=dev)
    zrows[:, :zc] = zq.permute(2, 0, 3, 4, 1).reshape(-1, zc)
    wyb = torch.zeros((2 * C, 64), dtype=bf16, device=dev)
    wyb[:C, :zc] = sd["n.conv_y.conv.weight"].reshape(C, zc).to(dev)
    wyb[C:, :zc] = sd["n.conv_b.conv.weight"].reshape(C, zc).to(dev)
    byb = torch.cat([sd["n.conv_y.conv.bias"], sd["n.conv_b.conv.bias"]]).to(dev).to(bf16)
    mod = ops.gemm(zrows, wyb, byb)
    from opendwm_amd.vae_cogvideox import _Ctx
    ctx = _Ctx(None, B, dev)
    ctx.Tz = Tz
    x = f.permute(2, 0, 3, 4, 1).reshape(-1, C).contiguous()
    grid = ops.Grid3D(T, B, h, w)
    out = ops.groupnorm_silu(x, B, T * h * w, sd["n.norm_layer.weight"].to(dev).to(bf16), sd["n.norm_layer.bias"].to(dev).to(bf16),
                             G, 1e-6, out_grid=grid, img_map=(B, h * w, 0, h * w, B * h * w),
                             zmap=dict(mod=mod, frames=T, videos=B, h=h, w=w, shift=shift, zt=ctx.zt(T)))
    m = ops._lib.RowMap2D()
    grid.fill(m)
    inner = out[2 * grid.frame_rows:].reshape(T * B, h + 2, w + 2, C)[:, 1:-1, 1:-1]
    got = inner.reshape(T, B, h, w, C).permute(1, 4, 0, 2, 3)
    e = rel_err(got, ref)
    border = out[2 * grid.frame_rows:].reshape(T * B, h + 2, w + 2, C)
    _log("spatial_norm3d", Tz=Tz, T=T, shift=shift, rel=e)
    assert e < 1e-2          # one bf16 rounding of the modulation rows + one of the output
    assert torch.count_nonzero(border[:, 0]) == 0 and torch.count_nonzero(border[:, :, 0]) == 0


@pytest.mark.parametrize("frames", [17, 9, 1])
def test_encode_vs_oracle(dev, frames):
    cfg = small_cfg()
    sd = _bf_sd(CV.make_state_dict(cfg, 0))
    m = _model(cfg, sd, dev)
    x = torch.randn(2, 3, frames, 32, 48, generator=torch.Generator().manual_seed(3)).to(bf16).float()
    ref = CV.encode_moments(sd, cfg, x)
    dist = m.encode(x.to(dev)).latent_dist
    e = rel_err(dist.parameters, ref)
    _log("cogvideox_encode", frames=frames, latent_frames=ref.shape[2], rel=e)
    assert dist.parameters.shape == ref.shape and e < TOL_MODEL
    assert torch.equal(dist.mode(), dist.parameters[:, :cfg["latent_channels"]])


@pytest.mark.parametrize("latent_frames", [5, 2, 1])
def test_decode_vs_oracle(dev, latent_frames):
    cfg = small_cfg()
    sd = _bf_sd(CV.make_state_dict(cfg, 1))
    m = _model(cfg, sd, dev)
    z = torch.randn(2, 16, latent_frames, 4, 6, generator=torch.Generator().manual_seed(4)).to(bf16).float()
    ref = CV.decode(sd, cfg, z)
    out = m.decode(z.to(dev), return_dict=False)[0]
    e = rel_err(out, ref)
    _log("cogvideox_decode", latent_frames=latent_frames, frames=ref.shape[2], rel=e)
    assert out.shape == ref.shape and e < TOL_MODEL


def test_full_width_decoder_block_vs_oracle(dev):
    """THUDM/CogVideoX-2b widths (512 / 256 / 256 / 128 channels, 32 groups, 4 resnets per up block) on a small latent"""
    cfg = CV.make_cogvideox_config()
    sd = _bf_sd(CV.make_state_dict(cfg, 2))
    m = _model(cfg, sd, dev)
    z = torch.randn(1, 16, 3, 2, 4, generator=torch.Generator().manual_seed(5)).to(bf16).float()
    ref = CV.decode(sd, cfg, z)
    out = m.decode(z.to(dev), return_dict=False)[0]
    e = rel_err(out, ref)
    _log("cogvideox_decode_full_width", frames=ref.shape[2], rel=e)
    assert out.shape == (1, 3, 9, 16, 32) and e < TOL_MODEL


def test_full_width_clip_vs_oracle_on_device(dev):
    """THUDM/CogVideoX-2b widths end to end - decode AND encode of one view x 5 latent frames x 8x14 latents (17 frames of
    64x112 px) - against the fp32 oracle evaluated on the device (on the host it takes minutes)."""
    cfg = CV.make_cogvideox_config()
    sd = _bf_sd(CV.make_state_dict(cfg, 0))
    m = _model(cfg, sd, dev)
    sdd = {k: v.to(dev) for k, v in sd.items()}
    z = torch.randn(1, 16, 5, 8, 14, generator=torch.Generator().manual_seed(0)).to(bf16).float().to(dev)
    ref = CV.decode(sdd, cfg, z)
    out = m.decode(z, return_dict=False)[0]
    e_dec = rel_err(out, ref)
    x = ref.clamp(-1, 1).to(bf16).float()
    mref = CV.encode_moments(sdd, cfg, x)
    mout = m.encode(x).latent_dist.parameters
    e_enc = rel_err(mout, mref)
    _log("cogvideox_full_width_clip", frames=ref.shape[2], rel_decode=e_dec, rel_encode=e_enc)
    assert out.shape == (1, 3, 17, 64, 112) and e_dec < TOL_MODEL and e_enc < TOL_MODEL


def test_rejects_cpu_and_bad_rank(dev):
    cfg = small_cfg()
    m = _model(cfg, _bf_sd(CV.make_state_dict(cfg, 0)), dev)
    with pytest.raises(RuntimeError):
        m.decode(torch.zeros(1, 16, 1, 4, 6))
    with pytest.raises(ValueError):
        m.decode(torch.zeros(1, 16, 4, 6, device=dev))


def test_latent_decoder_layouts(dev):
    """drivers.LatentDecoder: "(b v) c t h w" clips for the temporal VAE, the [frame, zeros] trick of the
    diffusion-forcing decode (ctsd.py:1606-1622), scaling / shift, postprocess to [0, 1]"""
    from opendwm_amd.drivers import LatentDecoder
    cfg = small_cfg()
    sd = _bf_sd(CV.make_state_dict(cfg, 1))
    m = _model(cfg, sd, dev)
    dec = LatentDecoder(m)
    assert dec.is_temporal_vae
    B, T, V = 1, 3, 2
    lat = torch.randn(B, T, V, 16, 4, 6, generator=torch.Generator().manual_seed(6)).to(bf16).float()
    z = (lat / cfg["scaling_factor"]).to(bf16).float().permute(0, 2, 3, 1, 4, 5).flatten(0, 1)
    ref = CV.decode(sd, cfg, z)                                               # (b v) c t h w
    want = (ref.unflatten(0, (B, V)).permute(0, 3, 1, 2, 4, 5).flatten(0, 2) / 2 + 0.5).clamp(0, 1)
    got = dec(lat.to(dev))
    e = rel_err(got, want)
    assert got.shape == (B * 9 * V, 3, 32, 48) and e < TOL_MODEL
    one = lat[:, 1:2]
    z1 = (one / cfg["scaling_factor"]).to(bf16).float().permute(0, 2, 3, 1, 4, 5).flatten(0, 1)
    ref1 = CV.decode(sd, cfg, torch.cat([z1, z1 * 0], 2)).chunk(2, dim=2)[0]
    want1 = (ref1.unflatten(0, (B, V)).permute(0, 3, 1, 2, 4, 5).flatten(0, 2) / 2 + 0.5).clamp(0, 1)
    got1 = dec(one.to(dev), diffusion_forcing=True)
    e1 = rel_err(got1, want1)
    _log("latent_decoder_tvae", rel_clip=e, rel_df_frame=e1)
    assert got1.shape == (B * 4 * V, 3, 32, 48) and e1 < TOL_MODEL


def _sharded_decode_worker(rank, world, port, kind, cfg, sd, lat, path):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from opendwm_amd import _lib
    from opendwm_amd.drivers import LatentDecoder
    _lib.load()
    dev = torch.device("cuda:0")
    vae = _decode_model(kind, cfg, sd, dev)
    dec = LatentDecoder(vae, group=dist.group.WORLD)
    out = [dec(lat.to(dev)), dec(lat[:, 1:2].to(dev), diffusion_forcing=True)]
    torch.save([o.cpu() for o in out], f"{path}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


def _decode_model(kind, cfg, sd, dev):
    if kind == "temporal":
        return _model(cfg, sd, dev)
    from opendwm_amd.vae import AutoencoderKL
    vae = AutoencoderKL(**cfg)
    vae.load_state_dict(sd)
    return vae.to(dev).to(bf16).eval()


@pytest.mark.parametrize("kind", ["temporal", "2d"])
def test_latent_decoder_sharded_over_two_ranks(dev, kind):
    """LatentDecoder(group=...): the decode half of the intra-sample sharding - 3 per-view clips (temporal VAE; uneven
    2 + 1 shares, the short one padded) or 9 images (2-D VAE; 5 + 4) decoded on two ranks (gloo, both on the one GPU)
    and all-gathered: every rank returns the whole batch, equal to the single-process decode (5e-3: other GEMM grids)."""
    import tempfile
    import torch.multiprocessing as mp
    from opendwm_amd.drivers import LatentDecoder
    if kind == "temporal":
        cfg = small_cfg()
        sd = _bf_sd(CV.make_state_dict(cfg, 1))
    else:
        from oracle import ctsd_oracle as O
        cfg = dict(block_out_channels=(64, 64, 128, 128), layers_per_block=2, norm_num_groups=16, latent_channels=16)
        sd = _bf_sd(O.make_vae_state_dict(cfg, 0))
    B, T, V = 1, 3, 3
    lat = torch.randn(B, T, V, 16, 8, 8, generator=torch.Generator().manual_seed(8)).to(bf16).float()
    dec = LatentDecoder(_decode_model(kind, cfg, sd, dev))
    single = [dec(lat.to(dev)).cpu(), dec(lat[:, 1:2].to(dev), diffusion_forcing=True).cpu()]
    ctx = mp.get_context("spawn")
    port = 29500 + (os.getpid() + 41) % 2000
    path = os.path.join(tempfile.mkdtemp(), "sharded_decode")
    procs = [ctx.Process(target=_sharded_decode_worker, args=(r, 2, port, kind, cfg, sd, lat, path)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    a, b = torch.load(path + ".0"), torch.load(path + ".1")
    errs = [rel_err(x, s) for x, s in zip(a, single)]
    _log("latent_decoder_sharded", kind=kind, rel_full=errs[0], rel_df_frame=errs[1])
    assert all(x.shape == s.shape for x, s in zip(a, single)) and all(torch.equal(x, y) for x, y in zip(a, b))
    assert max(errs) < 5e-3
