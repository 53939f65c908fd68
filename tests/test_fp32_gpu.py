"""The fp32 accuracy path (BASELINE.json north_star: "within 1e-3 rel fp32" of the reference's fp32 path): every kernel of
the mode against plain fp32 / fp64 PyTorch of the same op, then the MMDiT forward and the guided denoise loop against the
fp32 oracle - small configuration vs the CPU oracle, the 6-layer full-width stack of BASELINE config 3 vs the oracle on the
device.  Tolerance of the mode: 1e-3 relative (Frobenius); the kernels are held to 1e-4."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ctsd_oracle as O          # noqa: E402  (checker only)
from tests.common import rel_err, small_config, small_inputs, to_dev     # noqa: E402

pytestmark = pytest.mark.gpu
f32 = torch.float32
TOL_F32 = 1e-3
TOL_KERNEL_F32 = 1e-4


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


def _rand(shape, dev, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev)


@pytest.mark.parametrize("M,N,K", [(300, 192, 128), (1000, 512, 256), (4096, 1536, 1536), (77, 64, 64)])
def test_gemm_f32_epilogues(dev, M, N, K):
    """dwm_gemm_f32 (two-plane bf16 split of both operands on the bf16 MFMA main loop, fp32 finish) vs fp64 matmul:
    plain + bias + activations, GEGLU, gated residual, row-modulo / per-image residual, blend, per-head q/k RMSNorm"""
    from opendwm_amd import ops
    from opendwm_amd.blocks import geglu_pack
    a, w, b = _rand((M, K), dev, 1), _rand((N, K), dev, 2, K ** -0.5), _rand((N,), dev, 3, 0.1)
    ref = (a.double() @ w.double().t()) + b.double()
    errs = {}
    errs["plain"] = rel_err(ops.gemm(a, w, b), ref)
    errs["nobias"] = rel_err(ops.gemm(a, w, None), a.double() @ w.double().t())
    errs["gelu"] = rel_err(ops.gemm(a, w, b, act=ops.ACT_GELU_TANH), torch.nn.functional.gelu(ref, approximate="tanh"))
    errs["silu"] = rel_err(ops.gemm(a, w, b, act=ops.ACT_SILU), torch.nn.functional.silu(ref))
    if N % 64 == 0:
        h, g = ref.chunk(2, -1)
        errs["geglu"] = rel_err(ops.gemm(a, geglu_pack(w), geglu_pack(b), epilogue=ops.EPI_GEGLU), h * torch.nn.functional.gelu(g))
        nq = N // 64 // 2 * 64 if N >= 128 else 64
        rw = 1 + _rand((nq,), dev, 9, 0.1)
        x = ref.clone()
        xh = x[:, :nq].view(M, -1, 64)
        x[:, :nq] = (xh * torch.rsqrt(xh.pow(2).mean(-1, keepdim=True) + 1e-6)).view(M, nq) * rw.double()
        errs["rmshead"] = rel_err(ops.gemm(a, w, b, epilogue=ops.EPI_RMSHEAD, rms_w=rw, rms_ncols=nq, rms_eps=1e-6), x)
    rpg = 100
    gate, res = _rand(((M + rpg - 1) // rpg, N), dev, 4), _rand((M, N), dev, 5)
    rows = torch.arange(M, device=dev)
    errs["gate_res"] = rel_err(ops.gemm(a, w, b, epilogue=ops.EPI_RESID, gate=gate, rows_per_gate=rpg, res=res),
                               ref * gate.double()[rows // rpg] + res.double())
    tab = _rand((37, N), dev, 6)
    errs["res_mod"] = rel_err(ops.gemm(a, w, b, epilogue=ops.EPI_RESID, res=tab, res_mod=37), ref + tab.double()[rows % 37])
    per = _rand(((M + 49) // 50, N), dev, 7)
    errs["res_per_image"] = rel_err(ops.gemm(a, w, b, epilogue=ops.EPI_RESID, res=per, res_mod=-50), ref + per.double()[rows // 50])
    blend, alpha = _rand((M, N), dev, 8), torch.tensor([0.3, 1.0, 0.88], device=dev)
    rpa = (M + 2) // 3
    al = alpha.double()[rows // rpa][:, None]
    errs["blend"] = rel_err(ops.gemm(a, w, b, epilogue=ops.EPI_RESID, res=res, blend=blend, alpha=alpha, rows_per_alpha=rpa),
                            al * blend.double() + (1 - al) * (ref + res.double()))
    inplace = res.clone()
    ops.gemm(a, w, b, epilogue=ops.EPI_RESID, res=inplace, out=inplace)
    errs["inplace"] = rel_err(inplace, ref + res.double())
    _log("gemm_f32", M=M, N=N, K=K, **{k: f"{v:.2e}" for k, v in errs.items()})
    assert all(v < TOL_KERNEL_F32 for v in errs.values()), errs


def test_layernorm_f32(dev):
    from opendwm_amd import ops
    rows, D, rpm = 600, 1536, 150
    x, mod = _rand((rows, D), dev, 1), _rand((rows // rpm, 6 * D), dev, 2, 0.3)
    w, b = 1 + _rand((D,), dev, 3, 0.1), _rand((D,), dev, 4, 0.1)
    emb = _rand((rows // 50, D), dev, 5, 0.3)
    xd = x.double()
    n = torch.nn.functional.layer_norm(xd, (D,), eps=1e-6)
    r = torch.arange(rows, device=dev)
    sl = lambda i: mod[:, i * D:(i + 1) * D]
    y2 = torch.empty_like(x)
    y = ops.layernorm(x, eps=1e-6, scale=sl(1), shift=sl(0), rows_per_mod=rpm, scale2=sl(4), shift2=sl(3), out2=y2)
    e1 = rel_err(y, n * (1 + sl(1).double()[r // rpm]) + sl(0).double()[r // rpm])
    e2 = rel_err(y2, n * (1 + sl(4).double()[r // rpm]) + sl(3).double()[r // rpm])
    xs = torch.empty_like(x)
    ya = ops.layernorm(x, eps=1e-5, weight=w, bias=b, addvec=emb, rows_per_add=50, xsum=xs)
    xe = xd + emb.double()[r // 50]
    e3 = rel_err(ya, torch.nn.functional.layer_norm(xe, (D,), w.double(), b.double(), eps=1e-5))
    e4 = rel_err(xs, xe)
    _log("layernorm_f32", mod=e1, mod2=e2, affine_add=e3, xsum=e4)
    assert max(e1, e2, e3, e4) < 1e-5


def _attn_ref(q, k, v, rows, heads, mask=None, q1=None, k1=None, v1=None):
    P, L0 = rows.shape
    D = heads * 64

    def gather(x, x1):
        g = x[rows.reshape(-1)].view(P, L0, heads, 64)
        if x1 is not None:
            g = torch.cat([g, x1.view(P, -1, heads, 64)], 1)
        return g.transpose(1, 2)
    Q, K, V = gather(q.double(), None if q1 is None else q1.double()), gather(k.double(), None if k1 is None else k1.double()), \
        gather(v.double(), None if v1 is None else v1.double())
    o = O.sdpa(Q, K, V, None if mask is None else mask[:, None]).transpose(1, 2).reshape(P, -1, D)
    o0 = torch.zeros_like(q, dtype=torch.float64)
    o0[rows.reshape(-1)] = o[:, :L0].reshape(-1, D)
    return o0, None if q1 is None else o[:, L0:].reshape(-1, D)


def test_attention_f32_joint_rowmaps_and_masks(dev):
    """dwm_attention_f32: two-segment (joint) problems, every row map of the VT blocks, group and dense masks, ragged tails"""
    from opendwm_amd import ops
    heads = 3
    D = heads * 64
    errs = {}
    for I, N, Lc in ((3, 100, 10), (2, 448, 154), (4, 33, 0)):
        qkv = _rand((I * N, 3 * D), dev, 1)
        cqkv = _rand((I * Lc, 3 * D), dev, 2) if Lc else None
        out = torch.zeros((I * N, D), dtype=f32, device=dev)
        cout = torch.zeros((I * Lc, D), dtype=f32, device=dev) if Lc else None
        rm = ops.rowmap_identity(I, N)
        kw = dict(q1=cqkv[:, :D], k1=cqkv[:, D:2 * D], v1=cqkv[:, 2 * D:], out1=cout) if Lc else {}
        ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out, rm, heads, **kw)
        r0, r1 = _attn_ref(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], rm.rows().to(dev), heads,
                           **({k: v for k, v in kw.items() if k != "out1"}))
        errs[f"joint_{N}_{Lc}"] = max(rel_err(out, r0), rel_err(cout, r1) if Lc else 0.0)
    B, T, V, h, w = 2, 3, 3, 4, 6
    R = B * T * V * h * w
    qkv = _rand((R, 3 * D), dev, 3)
    for name, mk in (("cv_rowwise", ops.rowmap_crossview_rowwise), ("cv_full", ops.rowmap_crossview_full),
                     ("t_rowwise", ops.rowmap_temporal_rowwise), ("t_pointwise", ops.rowmap_temporal_pointwise),
                     ("t_full", ops.rowmap_temporal_full)):
        rm = mk(B, T, V, h, w)
        out = torch.zeros((R, D), dtype=f32, device=dev)
        gmask = ref_mask = None
        if name == "cv_rowwise":
            gmask = O.ring_crossview_mask(B, V).to(dev)
            gmask[1, 0, 2] = False
            p = torch.arange(rm.n_problems, device=dev)[:, None, None]
            l = torch.arange(rm.L0, device=dev)
            ref_mask = gmask[p // rm.p_per_mask, ((l // rm.group_size) % V)[None, :, None], ((l // rm.group_size) % V)[None, None, :]]
        ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out, rm, heads, group_mask=gmask)
        r0, _ = _attn_ref(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], rm.rows().to(dev), heads, mask=ref_mask)
        errs[name] = rel_err(out, r0)
        if ref_mask is not None:
            out2 = torch.zeros_like(out)
            ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out2, rm, heads, dense_mask=ref_mask)
            errs[name + "_dense"] = rel_err(out2, r0)
    _log("attention_f32", **{k: f"{v:.2e}" for k, v in errs.items()})
    assert all(v < 1e-5 for v in errs.values()), errs


def test_elementwise_f32(dev):
    from opendwm_amd import ops
    x = _rand((4096,), dev, 1, 3.0)
    assert rel_err(ops.silu(x), torch.nn.functional.silu(x.double())) < 1e-6
    t = torch.tensor([0.0, 1.0, 37.5, 999.0], device=dev)
    assert rel_err(ops.timestep_sinusoid(t, 256, dtype=f32), O.timesteps_sinusoid(t.cpu(), 256)) < 1e-4
    img = _rand((3, 16, 8, 12), dev, 2)
    cols = ops.patchify(img, 2, 64, dtype=f32)
    want = torch.nn.functional.unfold(img.double(), 2, stride=2).transpose(1, 2).reshape(-1, 64)
    assert torch.equal(cols.double(), want)
    tok = _rand((3 * 4 * 6, 64), dev, 3)
    back = ops.unpatchify(tok, 3, 16, 4, 6, 2)
    ref = torch.einsum("nhwpqc->nchpwq", tok.view(3, 4, 6, 2, 2, 16)).reshape(3, 16, 8, 12)
    assert torch.equal(back, ref)
    pred, lat = _rand((2, 5000), dev, 4), _rand((5000,), dev, 5)
    lat0, mi = lat.clone(), torch.empty(2, 5000, device=dev)
    ops.cfg_euler_step(pred.reshape(-1), lat, 4.0, -0.03, model_in=mi.view(-1))
    want = lat0.double() - 0.03 * (pred[0].double() + 4.0 * (pred[1].double() - pred[0].double()))
    assert rel_err(lat, want) < 1e-6 and torch.equal(mi[0], lat) and torch.equal(mi[1], lat)


def _fp32_model(cfg, sd, dev):
    from opendwm_amd.dit import DiTCrossviewTemporalConditionModel
    m = DiTCrossviewTemporalConditionModel(**cfg)
    m.load_state_dict(sd)
    m = m.to(dev).eval()                   # fp32 parameters
    m.compute_dtype = f32
    return m


@pytest.mark.parametrize("tt", ["rowwise", "pointwise", "full"])
def test_model_forward_fp32_vs_cpu_oracle(dev, tt):
    """small full-graph configuration, fp32 weights and inputs untouched (no bf16 rounding anywhere): the HIP fp32 path
    against the CPU oracle - north_star's fp32 tolerance, 1e-3; then the same model object in bf16 mode still gives the
    bf16 result (the precision switch is scoped to a forward)"""
    cfg = small_config(temporal_attention_type=tt)
    sd = O.make_state_dict(cfg, 0)
    inp = small_inputs(cfg, 0)
    inp["disable_temporal"] = torch.tensor([False, True])
    ref = O.dit_forward(sd, cfg, **inp)
    m = _fp32_model(cfg, sd, dev)
    di = to_dev(inp, dev)
    out, _, _ = m(di.pop("sample"), di.pop("timestep"), **di)
    e = rel_err(out[0], ref)
    assert out[0].dtype == f32
    m.compute_dtype = torch.bfloat16
    di = to_dev(inp, dev)
    out16, _, _ = m(di.pop("sample"), di.pop("timestep"), **di)
    e16 = rel_err(out16[0], ref)
    _log("model_forward_fp32", temporal=tt, rel_fp32=e, rel_bf16_same_model=e16)
    assert e < TOL_F32, e
    assert out16[0].dtype == torch.bfloat16 and 1e-4 < e16 < 2e-2


def test_gemm_f32_implicit_conv3x3(dev):
    """dwm_gemm_f32 as an implicit 3x3 convolution on a padded token grid (9 taps x 3 plane taps = the 27 tap slots), plain
    and with the in-place residual on the padded grid, against F.conv2d in fp64"""
    import torch.nn.functional as F
    from opendwm_amd import ops
    I, h, w, Cc, N = 3, 6, 10, 128, 192
    g = torch.Generator().manual_seed(5)
    x = torch.randn(I, Cc, h, w, generator=g)
    wt = torch.randn(N, Cc, 3, 3, generator=g) * (9 * Cc) ** -0.5
    b = torch.randn(N, generator=g) * 0.1
    ref = F.conv2d(x.double(), wt.double(), b.double(), padding=1).permute(0, 2, 3, 1).reshape(-1, N)
    grid = ops.PaddedGrid(I, h, w)
    idx = grid.interior_index().to(dev)
    xp = torch.zeros((grid.rows, Cc), dtype=f32, device=dev)
    xp[idx] = x.permute(0, 2, 3, 1).reshape(-1, Cc).to(dev)
    wp = wt.permute(0, 2, 3, 1).reshape(N, 9 * Cc).contiguous().to(dev)
    out = ops.gemm(xp, wp, b.to(dev), a_grid=grid, conv3x3=True)
    e0 = rel_err(out, ref)
    act = ops.gemm(xp, wp, b.to(dev), act=ops.ACT_RELU, a_grid=grid, conv3x3=True)
    e1 = rel_err(act, ref.clamp_min(0))
    # 1x1 convolution back onto the padded grid, residual in place (AdapterResnetBlock.block2)
    w1 = (torch.randn(Cc, N, generator=g) * N ** -0.5).to(dev)
    res = xp.clone()
    ops.gemm(out, w1, None, epilogue=ops.EPI_RESID, res=res, out=res, c_grid=grid)
    want = x.permute(0, 2, 3, 1).reshape(-1, Cc).double() + ref @ w1.double().cpu().T
    border = torch.ones(grid.rows, dtype=torch.bool, device=dev)
    border[idx] = False
    e2 = rel_err(res[idx], want)
    _log("gemm_f32_conv3x3", plain=e0, relu=e1, resid_on_grid=e2)
    assert max(e0, e1, e2) < 1e-4 and torch.count_nonzero(res[border]) == 0


@pytest.mark.parametrize("zero_convs", [True, False])
def test_model_with_layout_adapter_fp32_vs_cpu_oracle(dev, zero_convs):
    """the text+layout model (ImageAdapter residuals + point-wise temporal attention, the headline variant of bench.py) in
    the fp32 mode: fp32 convolutions of the adapter by dwm_gemm_f32, zero convolutions adding into the fp32 hidden state - or,
    use_zero_convs=False (adapters.py:33-36), the level outputs added as they are (the plain fp32 add)"""
    ac = dict(in_channels=6, channels=[128, 128, 128], is_downblocks=[True, False, False], num_res_blocks=2, downscale_factor=8,
              use_zero_convs=zero_convs)
    cfg = small_config(temporal_attention_type="pointwise", condition_image_adapter_config=ac)
    sd = O.make_state_dict(cfg, 0)
    inp = small_inputs(cfg, 0)
    g = torch.Generator().manual_seed(21)
    inp["condition_image_tensor"] = torch.rand(2, 3, 3, 6, 64, 96, generator=g)
    ref = O.dit_forward(sd, cfg, **inp)
    m = _fp32_model(cfg, sd, dev)
    di = to_dev(inp, dev)
    out, _, _ = m(di.pop("sample"), di.pop("timestep"), **di)
    e = rel_err(out[0], ref)
    # and the adapter alone
    feats = O.image_adapter(sd, cfg, inp["condition_image_tensor"])
    from opendwm_amd.blocks import STORE
    STORE.set_precision(f32)
    try:
        mine = m.condition_image_adapter.run(di["condition_image_tensor"])
    finally:
        STORE.set_precision(torch.bfloat16)
    ea = max(rel_err(a, f.flatten(0, -4).permute(0, 2, 3, 1).reshape(-1, f.shape[-3])) for a, f in zip(mine, feats))
    _log("model_forward_fp32_layout", zero_convs=zero_convs, rel_fp32=e, adapter_rel=ea)
    assert out[0].dtype == f32 and e < TOL_F32 and ea < 1e-4, (e, ea)


def test_denoise_fp32_vs_cpu_oracle(dev):
    """four guided FlowMatch-Euler steps in the fp32 mode (fp32 model input, fp32 prediction, fp32 CFG + Euler kernel)"""
    from opendwm_amd.pipeline import CTSDDenoiser
    cfg = small_config()
    sd = O.make_state_dict(cfg, 0)
    inp = small_inputs(cfg, 0)
    cond = {k: v for k, v in inp.items() if k not in ("sample", "timestep")}
    lat = torch.randn(1, 3, 3, 16, 8, 12, generator=torch.Generator().manual_seed(7))
    ref = O.denoise(sd, cfg, lat, cond, steps=4, guidance_scale=4.0)
    m = _fp32_model(cfg, sd, dev)
    out = CTSDDenoiser(m, guidance_scale=4.0, inference_steps=4).run(lat.to(dev), to_dev(cond, dev))
    e = rel_err(out, ref)
    _log("denoise_fp32", steps=4, rel=e)
    assert e < TOL_F32, e


def test_full_width_stack_fp32_vs_oracle_on_device(dev):
    """BASELINE config-3 token geometry (6 views x 16 frames x 32x56 latents, CFG batch 2, d = 1536, 24 heads, 154 text
    tokens) with the first 6 layers of the schedule, fp32 mode against the fp32 oracle on the same device"""
    cfg = O.make_config(num_layers=6, dual_attention_layers=[0, 1, 2, 3, 4, 5], crossview_block_layers=[1, 5],
                        temporal_block_layers=[2, 3])
    gen = torch.Generator().manual_seed(0)
    sd = {n: O.synth_param(n, s, cfg, gen) for n, s in O.param_shapes(cfg).items()}
    m = _fp32_model(cfg, sd, dev)
    inp = O.make_inputs(cfg, 2, 16, 6, 32, 56, seed=0)
    di = to_dev(inp, dev)
    with torch.no_grad():
        ref = O.dit_forward({k: v.to(dev) for k, v in sd.items()}, cfg, **di)
    out, _, _ = m(di.pop("sample"), di.pop("timestep"), **di)
    e = rel_err(out[0], ref)
    _log("full_width_6layers_fp32", rel=e, finite=bool(torch.isfinite(out[0]).all()))
    assert e < TOL_F32, e


# ------------------------------------------------------------------------------------------ SD 2.1 UNet + 2-D VAE (round 4)
def test_glue_kernels_f32(dev):
    """fp32 forms of the token-major glue kernels the UNet / VAE / layout adapter use: GroupNorm(+SiLU) compact, into a padded
    grid and with the (b v) x (t h w) row map of TemporalResnetBlock; nearest 2x upsample into a padded grid; padded-grid
    scatter; row softmax; pixel-unshuffle; 2x2 average pooling - against plain fp64 torch"""
    import torch.nn.functional as F
    from opendwm_amd import ops
    from opendwm_amd.ops import PaddedGrid
    I, h, w, C, G = 3, 8, 12, 320, 32
    x = _rand((I * h * w, C), dev, 1, 2.0) + 0.3
    ga, be = _rand((C,), dev, 2, 0.2) + 1.0, _rand((C,), dev, 3, 0.5)
    img = x.double().view(I, h, w, C).permute(0, 3, 1, 2)
    ref = F.silu(F.group_norm(img, G, ga.double(), be.double(), 1e-5)).permute(0, 2, 3, 1).reshape(I * h * w, C)
    e_gn = rel_err(ops.groupnorm_silu(x, I, h * w, ga, be, G, 1e-5), ref.float())
    grid = PaddedGrid(I, h, w)
    pad = ops.groupnorm_silu(x, I, h * w, ga, be, G, 1e-5, out_grid=grid)
    assert pad.dtype == f32 and pad.shape[0] == grid.rows
    e_gn_pad = rel_err(pad[grid.interior_index().to(dev)], ref.float())
    assert float(pad.abs().sum()) > 0 and torch.count_nonzero(pad.view(I, h + 2, w + 2, C)[:, 0]) == 0      # the border stays zero
    # row-mapped statistics: image = (b, v), pixels = (t, h, w) of rows ordered (b t v)(h w)
    B, T, V, N = 1, 3, 2, 16
    xt = _rand((B * T * V * N, 64), dev, 4)
    g2, b2 = _rand((64,), dev, 5, 0.2) + 1.0, _rand((64,), dev, 6, 0.3)
    vol = xt.double().view(B, T, V, N, 64).permute(0, 2, 4, 1, 3).reshape(B * V, 64, T * N)
    rt = F.group_norm(vol, 8, g2.double(), b2.double(), 1e-5).view(B, V, 64, T, N).permute(0, 3, 1, 4, 2).reshape(-1, 64)
    yt = ops.groupnorm_silu(xt, B * V, T * N, g2, b2, 8, 1e-5, silu=False, img_map=(V, N, T * V * N, N, V * N))
    e_map = rel_err(yt, rt.float())
    # upsample / pad / softmax / unshuffle / avgpool
    up = ops.upsample2_padded(x, I, h, w)
    g2x = PaddedGrid(I, 2 * h, 2 * w)
    want = F.interpolate(img.float(), scale_factor=2, mode="nearest").permute(0, 2, 3, 1).reshape(-1, C)
    assert torch.equal(up[g2x.interior_index().to(dev)], want)
    assert torch.equal(ops.pad_tokens(x, grid)[grid.interior_index().to(dev)], x)
    s = _rand((100, 1024), dev, 7, 3.0)
    e_sm = rel_err(ops.softmax_rows(s, 0.25), torch.softmax(s.double() * 0.25, -1).float())
    px = _rand((2, 6, 16, 24), dev, 8)
    tok = ops.unshuffle_tokens(px, 8, dtype=f32)
    wantu = F.pixel_unshuffle(px, 8).permute(0, 2, 3, 1).reshape(2 * 2 * 3, 6 * 64)
    assert tok.dtype == f32 and torch.equal(tok[:, :384], wantu) and tok.shape[1] == 384
    pooled = ops.avgpool2_tokens(x, I, h, w)
    e_pool = rel_err(pooled, F.avg_pool2d(img, 2).permute(0, 2, 3, 1).reshape(-1, C).float())
    _log("glue_kernels_f32", gn=e_gn, gn_pad=e_gn_pad, gn_mapped=e_map, softmax=e_sm, avgpool=e_pool)
    assert max(e_gn, e_gn_pad, e_map, e_sm, e_pool) < 1e-5


def test_gemm_f32_strided_and_temporal_taps(dev):
    """dwm_gemm_f32 as the stride-2 3x3 convolutions (Downsample2D: symmetric padding 1 in the UNet, (0, 1) padding in the VAE
    encoder) and as the 3-tap temporal convolution of TemporalResnetBlock (Conv3d (3,1,1) over a T-padded row layout)"""
    import torch.nn.functional as F
    from opendwm_amd import ops
    from opendwm_amd.ops import PaddedGrid, TimeGrid
    I, h, w, C, N = 2, 8, 12, 64, 128
    x = _rand((I, C, h, w), dev, 1)
    wt, b = _rand((N, C, 3, 3), dev, 2, (9 * C) ** -0.5), _rand((N,), dev, 3, 0.1)
    grid = PaddedGrid(I, h, w)
    xp = ops.pad_tokens(x.permute(0, 2, 3, 1).reshape(I * h * w, C).contiguous(), grid)
    wm = wt.permute(0, 2, 3, 1).reshape(N, 9 * C).contiguous()
    errs = {}
    for name, s2, ref in (("sym", "sym", F.conv2d(x.double(), wt.double(), b.double(), stride=2, padding=1)),
                          ("asym", True, F.conv2d(F.pad(x.double(), (0, 1, 0, 1)), wt.double(), b.double(), stride=2))):
        y = ops.gemm(xp, wm, b, a_grid=grid, conv3x3=True, stride2=s2)
        errs[name] = rel_err(y, ref.permute(0, 2, 3, 1).reshape(-1, N).float())
    B, T, VN = 2, 4, 24
    tg = TimeGrid(B, T, VN)
    xs = _rand((B, T, VN, C), dev, 4)
    w3, b3 = _rand((N, C, 3), dev, 5, (3 * C) ** -0.5), _rand((N,), dev, 6, 0.1)
    pad = torch.zeros((tg.rows, C), dtype=f32, device=dev)
    pad.view(B, T + 2, VN, C)[:, 1:T + 1] = xs
    y3 = ops.gemm(pad, w3.permute(0, 2, 1).reshape(N, 3 * C).contiguous(), b3, a_grid=tg, conv_taps=tg.tap_shifts())
    ref3 = F.conv1d(xs.double().permute(0, 2, 3, 1).reshape(B * VN, C, T), w3.double(), b3.double(), padding=1)
    errs["temporal"] = rel_err(y3, ref3.view(B, VN, N, T).permute(0, 3, 1, 2).reshape(-1, N).float())
    _log("gemm_f32_strided_and_temporal_taps", **errs)
    assert max(errs.values()) < TOL_KERNEL_F32, errs


@pytest.mark.parametrize("rowwise", [True, False])
def test_unet_forward_fp32_vs_cpu_oracle(dev, rowwise):
    """the SD 2.1 cross-view temporal UNet, whole graph at small width (the configuration of tests/test_unet_gpu.py), fp32
    weights and inputs untouched: `model.compute_dtype = torch.float32` against the CPU oracle at north_star's fp32 tolerance;
    then the same object back in bf16"""
    from oracle import unet_oracle as U
    from opendwm_amd.unet import UNetCrossviewTemporalConditionModel
    cfg = U.make_unet_config(block_out_channels=(128, 256, 512, 512), num_attention_heads=(2, 4, 8, 8), cross_attention_dim=128,
                             projection_class_embeddings_input_dim=11 * 256, enable_rowwise_crossview=rowwise, enable_rowwise_temporal=rowwise)
    sd = U.make_unet_state_dict(cfg, 0)
    inp = U.make_unet_inputs(cfg, 2, 3, 3, 16, 24, text_len=10)
    inp["disable_temporal"] = torch.tensor([False, True])
    if not rowwise:
        inp["crossview_attention_mask"] = None
    ref = U.unet_forward(sd, cfg, **inp)
    m = UNetCrossviewTemporalConditionModel(**cfg)
    m.load_state_dict(sd, strict=True)
    m = m.to(dev).eval()
    m.compute_dtype = f32
    di = to_dev(inp, dev)
    out = m(di.pop("sample"), di.pop("timesteps"), **di)[0][0]
    e = rel_err(out, ref)
    assert out.dtype == f32 and out.shape == ref.shape
    m.compute_dtype = torch.bfloat16
    di = to_dev(inp, dev)
    out16 = m(di.pop("sample"), di.pop("timesteps"), **di)[0][0]
    e16 = rel_err(out16, ref)
    _log("unet_forward_fp32", rowwise=rowwise, rel_fp32=e, rel_bf16_same_model=e16)
    assert e < TOL_F32, e
    assert out16.dtype == torch.bfloat16 and 1e-4 < e16 < 2e-2


@pytest.mark.cost(19, optional=True)
def test_unet_full_width_config0_fp32_vs_oracle_on_device(dev):
    """BASELINE.json configs[0] - the configuration the reference's CPU fp32 denoise is quoted on - at FULL width (SD 2.1: 320 /
    640 / 1280 / 1280 channels, 1.92 B parameters), six views x one frame x 256x256 px (latents [1,1,6,4,32,32], CFG batch 2, 77
    text tokens): the fp32 path against the fp32 oracle evaluated on the device (the CPU needs minutes for it)"""
    from oracle import unet_oracle as U
    from opendwm_amd.unet import UNetCrossviewTemporalConditionModel
    cfg = U.make_unet_config()
    sd = U.make_unet_state_dict(cfg, 0)
    inp = U.make_unet_inputs(cfg, 2, 1, 6, 32, 32, text_len=77)
    m = UNetCrossviewTemporalConditionModel(**cfg)
    m.load_state_dict(sd, strict=True)
    m = m.to(dev).eval()
    m.compute_dtype = f32
    di = to_dev(inp, dev)
    out = m(di.pop("sample"), di.pop("timesteps"), **di)[0][0]
    del m
    torch.cuda.empty_cache()
    with torch.no_grad():
        ref = U.unet_forward({k: v.to(dev) for k, v in sd.items()}, cfg, **to_dev(inp, dev))
    e = rel_err(out, ref)
    _log("unet_full_width_config0_fp32", rel=e, finite=bool(torch.isfinite(out).all()))
    assert out.shape == (2, 1, 6, 4, 32, 32) and out.dtype == f32 and e < TOL_F32, e


@pytest.mark.parametrize("kind", ["sd35", "sd21"])
def test_vae_fp32_vs_cpu_oracle(dev, kind):
    """AutoencoderKL encode + decode with `vae.compute_dtype = torch.float32` at small width (SD 3.5 form; SD 2.1 form with
    quant / post-quant 1x1 convolutions) against the CPU oracle"""
    from opendwm_amd.vae import AutoencoderKL
    vcfg = dict(block_out_channels=(64, 64, 128, 128), layers_per_block=2, norm_num_groups=16, latent_channels=16 if kind == "sd35" else 4,
                use_quant_conv=kind == "sd21", use_post_quant_conv=kind == "sd21")
    sd = O.make_vae_state_dict(vcfg, 0)
    vae = AutoencoderKL(**vcfg)
    vae.load_state_dict(sd)
    vae = vae.to(dev).eval()
    vae.compute_dtype = f32
    g = torch.Generator().manual_seed(3)
    z = torch.randn(2, vcfg["latent_channels"], 8, 16, generator=g)
    x = torch.randn(2, 3, 64, 128, generator=g)
    out = vae.decode(z.to(dev), return_dict=False)[0]
    e_dec = rel_err(out, O.vae_decode(sd, vcfg, z))
    mom = vae.encode(x.to(dev)).latent_dist.parameters
    e_enc = rel_err(mom, O.vae_encode_moments(sd, vcfg, x))
    _log("vae_fp32", kind=kind, rel_decode=e_dec, rel_encode=e_enc)
    assert out.dtype == f32 and e_dec < TOL_F32 and e_enc < TOL_F32, (e_dec, e_enc)
    vae.compute_dtype = torch.bfloat16
    e16 = rel_err(vae.to(torch.bfloat16).decode(z.to(dev), return_dict=False)[0], O.vae_decode(sd, vcfg, z))
    assert 1e-4 < e16 < 2e-2


def test_vae_decode_fp32_full_width_vs_oracle_on_device(dev):
    """the SD 2.1 VAE decoder at its real widths (128 / 256 / 512 / 512, 4 latent channels, post-quant convolution) on the
    latents of BASELINE.json configs[0] (six 32x32 latents -> 256x256 px), fp32 path against the fp32 oracle on the device"""
    from opendwm_amd.vae import AutoencoderKL
    vcfg = dict(block_out_channels=(128, 256, 512, 512), layers_per_block=2, norm_num_groups=32, latent_channels=4,
                use_quant_conv=True, use_post_quant_conv=True)
    sd = O.make_vae_state_dict(vcfg, 0)
    vae = AutoencoderKL(**vcfg)
    vae.load_state_dict(sd)
    vae = vae.to(dev).eval()
    vae.compute_dtype = f32
    z = torch.randn(6, 4, 32, 32, generator=torch.Generator().manual_seed(0)).to(dev)
    out = vae.decode(z, return_dict=False)[0]
    ref = O.vae_decode({k: v.to(dev) for k, v in sd.items()}, vcfg, z)
    e = rel_err(out, ref)
    _log("vae_decode_fp32_full_width", rel=e, shape=list(out.shape))
    assert out.shape == (6, 3, 256, 256) and out.dtype == f32 and e < TOL_F32, e


def test_unet_denoise_loop_fp32_vs_cpu_oracle(dev):
    """the SD 2.1 guided DPM-Solver++(2M) loop (BASELINE.json configs[0] is this loop on the reference's CPU fp32 path) with
    `model.compute_dtype = torch.float32`: fp32 model input, fp32 prediction, fp32 scheduler update (dwm_cfg_multistep_f32)"""
    from oracle import unet_oracle as U
    from opendwm_amd.pipeline import UNetDenoiser
    from opendwm_amd.unet import UNetCrossviewTemporalConditionModel
    cfg = U.make_unet_config(block_out_channels=(128, 256, 512, 512), num_attention_heads=(2, 4, 8, 8), cross_attention_dim=128,
                             projection_class_embeddings_input_dim=11 * 256)
    sd = U.make_unet_state_dict(cfg, 0)
    inp = U.make_unet_inputs(cfg, 2, 2, 3, 16, 24, text_len=10)
    lat = inp.pop("sample")[:1]
    inp.pop("timesteps")
    steps = 4
    ref = U.unet_denoise(sd, cfg, lat, inp, steps, 3.0)
    m = UNetCrossviewTemporalConditionModel(**cfg)
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    m.compute_dtype = f32
    den = UNetDenoiser(m, 3.0, steps)
    out = den.run(lat.to(dev), to_dev(inp, dev))
    e = rel_err(out, ref)
    _log("unet_denoise_loop_fp32", steps=steps, rel=e)
    assert den.model_in.dtype == f32 and e < TOL_F32, e


# ------------------------------------------------------------------------- the variants the fp32 mode refused until round 4
def test_model_explicit_perspective_fp32_vs_cpu_oracle(dev):
    """perspective_modeling_type="explicit" (crossview_temporal_dit.py:11-102, 440-458) in the fp32 mode: the ray features in fp32
    (dwm_ray_features_f32) against the oracle's positional encodings of the executed reference rays, RayEncoder.proj through
    dwm_gemm_f32 inside every VT block, the whole forward against the CPU oracle and the executed reference forward"""
    fx = torch.load(os.path.join(ROOT, "tests", "golden", "reference_forward.pt"))["explicit"]
    cfg = small_config(perspective_modeling_type="explicit")
    sd = O.make_state_dict(cfg, 0)
    inp = small_inputs(cfg, 0)
    inp.pop("added_time_ids")
    cams = {k: fx[k] for k in ("camera_intrinsics_norm", "camera2referego")}
    m = _fp32_model(cfg, sd, dev)
    hh, ww = inp["sample"].shape[-2] // 2, inp["sample"].shape[-1] // 2
    from opendwm_amd.blocks import STORE
    STORE.set_precision(f32)
    try:
        feat = m.rayencoder.features(cams["camera_intrinsics_norm"].to(dev), cams["camera2referego"].to(dev), hh, ww)
    finally:
        STORE.set_precision(torch.bfloat16)
    I = fx["rays_o"].shape[0]
    want = torch.cat([O.positional_encoding(fx["rays_o"].unsqueeze(1), 8).view(I, 1, 1, -1).repeat(1, hh, ww, 1),
                      O.positional_encoding(fx["rays_d"].flatten(1, 2), 4).view(I, hh, ww, -1)], -1).view(I * hh * ww, 72)
    e_feat = (feat[:, :72].cpu() - want).abs().max().item()
    assert feat.dtype == f32 and feat.shape == (I * hh * ww, 128) and torch.count_nonzero(feat[:, 72:]) == 0
    ref = O.dit_forward(sd, cfg, **inp, **cams)
    di = to_dev({**inp, **cams}, dev)
    out, _, _ = m(di.pop("sample"), di.pop("timestep"), **di)
    e, efx = rel_err(out[0], ref), rel_err(out[0], fx["output"])
    _log("explicit_perspective_fp32", feature_max_abs=e_feat, rel_vs_oracle=e, rel_vs_reference_forward=efx)
    # features: sin / cos of arguments up to 128 pi |o| (camera origins of a few metres): fp32 argument reduction, ~1e-4 absolute
    assert out[0].dtype == f32 and e_feat < 5e-4 and e < TOL_F32 and efx < TOL_F32, (e_feat, e, efx)


def _frame_shard_fp32_worker(rank, world, port, cfg, sd, lat, cond, path):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from opendwm_amd.pipeline import CTSDDenoiser
    d = torch.device("cuda:0")
    m = _fp32_model(cfg, sd, d)
    out = CTSDDenoiser(m, guidance_scale=4.0, inference_steps=4, frame_group=dist.group.WORLD).run(lat.to(d), to_dev(cond, d))
    torch.save(out.cpu(), f"{path}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("temporal", ["rowwise", "pointwise"])
def test_frame_shard_two_ranks_fp32_vs_cpu_oracle(dev, temporal):
    """intra-sample frame sharding (opendwm_amd.sharding) in the fp32 mode: the 4 frames of one sample on two ranks (gloo, both on
    the one GPU), fp32 hidden state through the all-to-alls around every temporal block; every rank returns the whole sample,
    within the mode's tolerance of the CPU oracle's unsharded guided loop"""
    import tempfile
    import torch.multiprocessing as mp
    cfg = small_config(temporal_attention_type=temporal)
    sd = O.make_state_dict(cfg, 0)
    inp = small_inputs(cfg, 0, T=4)
    cond = {k: v for k, v in inp.items() if k not in ("sample", "timestep")}
    lat = torch.randn(1, 4, 3, 16, 8, 12, generator=torch.Generator().manual_seed(13))
    ref = O.denoise(sd, cfg, lat, cond, steps=4, guidance_scale=4.0)
    ctx = mp.get_context("spawn")
    port = 29500 + (os.getpid() + 23) % 2000
    path = os.path.join(tempfile.mkdtemp(), "frame_shard_fp32")
    procs = [ctx.Process(target=_frame_shard_fp32_worker, args=(r, 2, port, cfg, sd, lat, cond, path)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    a, b = torch.load(path + ".0"), torch.load(path + ".1")
    e = rel_err(a, ref)
    _log("frame_shard_fp32", temporal=temporal, ranks_equal=bool(torch.equal(a, b)), rel_vs_oracle=e)
    assert a.shape == ref.shape and torch.equal(a, b) and e < TOL_F32, e


# ------------------------------------------------------------------------- the temporal VAE (AutoencoderKLCogVideoX) in fp32
def test_temporal_vae_kernels_f32(dev):
    """the fp32 forms the temporal VAE adds: the 27-tap causal convolution through dwm_gemm_f32 (three groups of 9 taps, Grid3D with
    its two leading context frames), frame mixes, CogVideoXSpatialNorm3D (dwm_groupnorm_spatial_f32)"""
    import torch.nn.functional as F
    from oracle import cogvideox_vae_oracle as CV
    from opendwm_amd import ops
    from opendwm_amd.vae_cogvideox import _Ctx
    # causal 3x3x3 convolution, first call (frame 0 twice) and continuation (cache of the previous call)
    T, B, h, w, Cc, N = 3, 2, 4, 6, 64, 72
    x1, x2 = _rand((B, Cc, T, h, w), dev, 1), _rand((B, Cc, T, h, w), dev, 2)
    wt, b = _rand((N, Cc, 3, 3, 3), dev, 3, (27 * Cc) ** -0.5), _rand((N,), dev, 4)
    sd = {"c.conv.weight": wt.cpu(), "c.conv.bias": b.cpu()}
    cache = CV.ConvCache()
    refs = [CV.causal_conv3d(sd, "c", x.cpu(), cache) for x in (x1, x2)]
    grid = ops.Grid3D(T, B, h, w)
    fr = grid.frame_rows
    wp = wt.permute(0, 2, 3, 4, 1).reshape(N, 27 * Cc).contiguous()
    prev, errs = None, []
    for x, ref in zip((x1, x2), refs):
        buf = ops.pad_tokens(x.permute(2, 0, 3, 4, 1).reshape(-1, Cc).contiguous(), grid)
        if prev is None:
            buf[:fr].copy_(buf[2 * fr:3 * fr])
            buf[fr:2 * fr].copy_(buf[2 * fr:3 * fr])
        else:
            buf[:2 * fr].copy_(prev)
        prev = buf[T * fr:(T + 2) * fr].clone()
        out = ops.gemm(buf, wp, b, a_grid=grid, conv_taps=grid.tap_shifts())
        errs.append(rel_err(out.reshape(T, B, h, w, N).permute(1, 4, 0, 2, 3), ref))
    assert out.dtype == f32
    # frame mix
    xm = _rand((5, 6, 40), dev, 5)
    mix = ops.frame_mix(xm, 240, [0, 1, 3], [0, 2, 4], [1.0, 0.5, 0.5], [0.0, 0.5, 0.5]).view(3, 6, 40)
    e_mix = rel_err(mix, torch.stack([xm[0], 0.5 * (xm[1] + xm[2]), 0.5 * (xm[3] + xm[4])]))
    assert torch.equal(ops.frame_mix(xm, 240, [0, 1, 1, 2, 2], [0] * 5, [1.0] * 5, [0.0] * 5).view(5, 6, 40), xm[[0, 1, 1, 2, 2]])
    # spatial norm, odd clip (separate first frame) at 4x the latent resolution
    Tz, Ts, shift, hz, wz, C2, zc, G = 3, 9, 2, 3, 4, 64, 16, 8
    hh, ww = hz << shift, wz << shift
    fmap, zq = _rand((B, C2, Ts, hh, ww), dev, 6), _rand((B, zc, Tz, hz, wz), dev, 7)
    g = torch.Generator().manual_seed(8)
    nsd = {"n.norm_layer.weight": 1 + 0.1 * torch.randn(C2, generator=g), "n.norm_layer.bias": 0.1 * torch.randn(C2, generator=g),
           "n.conv_y.conv.weight": torch.randn(C2, zc, 1, 1, 1, generator=g) * 0.2, "n.conv_y.conv.bias": 1 + 0.1 * torch.randn(C2, generator=g),
           "n.conv_b.conv.weight": torch.randn(C2, zc, 1, 1, 1, generator=g) * 0.2, "n.conv_b.conv.bias": 0.1 * torch.randn(C2, generator=g)}
    ref = F.silu(CV.norm3d(nsd, "n", fmap.cpu(), zq.cpu(), G, 1e-6, CV.ConvCache()))
    zrows = torch.zeros((Tz * B * hz * wz, 64), dtype=f32, device=dev)
    zrows[:, :zc] = zq.permute(2, 0, 3, 4, 1).reshape(-1, zc)
    wyb = torch.zeros((2 * C2, 64), dtype=f32, device=dev)
    wyb[:C2, :zc] = nsd["n.conv_y.conv.weight"].reshape(C2, zc).to(dev)
    wyb[C2:, :zc] = nsd["n.conv_b.conv.weight"].reshape(C2, zc).to(dev)
    mod = ops.gemm(zrows, wyb, torch.cat([nsd["n.conv_y.conv.bias"], nsd["n.conv_b.conv.bias"]]).to(dev))
    ctx = _Ctx(None, B, dev)
    ctx.Tz = Tz
    g3 = ops.Grid3D(Ts, B, hh, ww)
    out = ops.groupnorm_silu(fmap.permute(2, 0, 3, 4, 1).reshape(-1, C2).contiguous(), B, Ts * hh * ww, nsd["n.norm_layer.weight"].to(dev),
                             nsd["n.norm_layer.bias"].to(dev), G, 1e-6, out_grid=g3, img_map=(B, hh * ww, 0, hh * ww, B * hh * ww),
                             zmap=dict(mod=mod, frames=Ts, videos=B, h=hh, w=ww, shift=shift, zt=ctx.zt(Ts)))
    pad = out[2 * g3.frame_rows:].reshape(Ts * B, hh + 2, ww + 2, C2)
    e_sn = rel_err(pad[:, 1:-1, 1:-1].reshape(Ts, B, hh, ww, C2).permute(1, 4, 0, 2, 3), ref)
    _log("temporal_vae_kernels_f32", conv27_first=errs[0], conv27_cached=errs[1], frame_mix=e_mix, spatial_norm=e_sn)
    assert out.dtype == f32 and torch.count_nonzero(pad[:, 0]) == 0 and torch.count_nonzero(pad[:, :, 0]) == 0
    assert max(errs) < TOL_KERNEL_F32 and e_mix < 1e-6 and e_sn < TOL_KERNEL_F32, (errs, e_mix, e_sn)


def _tvae_fp32(cfg, sd, dev):
    from opendwm_amd.vae_cogvideox import AutoencoderKLCogVideoX
    keep = ("in_channels", "out_channels", "block_out_channels", "latent_channels", "layers_per_block", "norm_eps",
            "norm_num_groups", "temporal_compression_ratio", "scaling_factor", "shift_factor")
    m = AutoencoderKLCogVideoX(**{k: cfg[k] for k in keep})
    m.load_state_dict(sd, strict=True)
    m = m.to(dev).eval()                   # fp32 parameters
    m.compute_dtype = f32
    return m


@pytest.mark.parametrize("frames", [17, 9])
def test_temporal_vae_fp32_vs_cpu_oracle(dev, frames):
    """AutoencoderKLCogVideoX with compute_dtype = float32, fp32 weights and inputs untouched: encode (chunks of 8 frames with the
    remainder first, causal caches across chunks) and decode (chunks of 2 latent frames) against the CPU oracle, north_star's
    1e-3; then the same object in bf16 mode still gives the bf16 result"""
    from oracle import cogvideox_vae_oracle as CV
    cfg = CV.make_cogvideox_config(block_out_channels=(64, 64, 128, 128), layers_per_block=1, norm_num_groups=8)
    sd = CV.make_state_dict(cfg, 0)
    m = _tvae_fp32(cfg, sd, dev)
    x = torch.randn(2, 3, frames, 32, 48, generator=torch.Generator().manual_seed(3))
    mref = CV.encode_moments(sd, cfg, x)
    dist = m.encode(x.to(dev)).latent_dist
    e_enc = rel_err(dist.parameters, mref)
    z = mref[:, :cfg["latent_channels"]].contiguous()
    ref = CV.decode(sd, cfg, z)
    out = m.decode(z.to(dev), return_dict=False)[0]
    e_dec = rel_err(out, ref)
    m.compute_dtype = torch.bfloat16
    out16 = m.to(torch.bfloat16).decode(z.to(dev), return_dict=False)[0]
    e16 = rel_err(out16, ref)
    _log("temporal_vae_fp32", frames=frames, latent_frames=mref.shape[2], rel_encode=e_enc, rel_decode=e_dec, rel_decode_bf16_same_model=e16)
    assert dist.parameters.shape == mref.shape and out.shape == ref.shape and out.dtype == f32
    assert e_enc < TOL_F32 and e_dec < TOL_F32, (e_enc, e_dec)
    assert out16.dtype == torch.bfloat16 and 1e-4 < e16 < 2e-2


def test_temporal_vae_fp32_full_width_vs_oracle_on_device(dev):
    """THUDM/CogVideoX-2b widths (128 / 256 / 256 / 512 channels, 32 groups, 3 + 1 resnets per block): decode of one clip of 5 latent
    frames x 8x14 latents (17 frames of 64x112 px) and the encode of the result in fp32, against the oracle on the device"""
    from oracle import cogvideox_vae_oracle as CV
    cfg = CV.make_cogvideox_config()
    sd = CV.make_state_dict(cfg, 0)
    m = _tvae_fp32(cfg, sd, dev)
    sdd = {k: v.to(dev) for k, v in sd.items()}
    z = torch.randn(1, 16, 5, 8, 14, generator=torch.Generator().manual_seed(0)).to(dev)
    prev = torch.backends.cuda.matmul.allow_tf32, torch.backends.cudnn.allow_tf32
    torch.backends.cuda.matmul.allow_tf32 = torch.backends.cudnn.allow_tf32 = False
    try:
        ref = CV.decode(sdd, cfg, z)
        out = m.decode(z, return_dict=False)[0]
        x = ref.clamp(-1, 1)
        mref = CV.encode_moments(sdd, cfg, x)
        mout = m.encode(x).latent_dist.parameters
    finally:
        torch.backends.cuda.matmul.allow_tf32, torch.backends.cudnn.allow_tf32 = prev
    e_dec, e_enc = rel_err(out, ref), rel_err(mout, mref)
    _log("temporal_vae_fp32_full_width", frames=ref.shape[2], rel_decode=e_dec, rel_encode=e_enc)
    assert out.shape == (1, 3, 17, 64, 112) and e_dec < TOL_F32 and e_enc < TOL_F32, (e_dec, e_enc)
