"""CPU leg (`-m "not gpu"`): the oracle against its golden fixtures and internal
cross-checks, the host-side logic (row maps, weight packing, schedule, FLOP model) and the
C-ABI surface.  No kernel is executed here."""
import ctypes
import os
import re

import einops
import pytest
import torch
import torch.nn.functional as F

from oracle import ctsd_oracle as O
from tests.common import GOLDEN, rel_err, small_config, small_inputs

torch.set_num_threads(4)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------ oracle vs golden
def test_oracle_forward_matches_golden(small_cfg):
    sd = O.make_state_dict(small_cfg, seed=0)
    inp = small_inputs(small_cfg, seed=0)
    tr = {}
    y = O.dit_forward(sd, small_cfg, trace=tr, **inp)
    gold = torch.load(os.path.join(GOLDEN, "dit_small_forward.pt"))
    assert y.shape == (2, 3, 3, 16, 8, 12)
    assert rel_err(y, gold["output"]) < 1e-5
    for k, v in gold["trace"].items():
        assert rel_err(tr[k], v) < 1e-5, k


@pytest.mark.parametrize("tt", ["pointwise", "full"])
def test_oracle_temporal_variants_match_golden(tt):
    cfg = small_config(temporal_attention_type=tt)
    sd = O.make_state_dict(small_config(), seed=0)
    y = O.dit_forward(sd, cfg, **small_inputs(cfg, seed=0))
    gold = torch.load(os.path.join(GOLDEN, f"dit_small_forward_{tt}.pt"))
    assert rel_err(y, gold["output"]) < 1e-5


def test_oracle_denoise_matches_golden(small_cfg):
    sd = O.make_state_dict(small_cfg, seed=0)
    inp = small_inputs(small_cfg, seed=0)
    gold = torch.load(os.path.join(GOLDEN, "denoise_small_2steps.pt"))
    cond = {k: v for k, v in inp.items() if k not in ("sample", "timestep")}
    out = O.denoise(sd, small_cfg, gold["latents_in"], cond, steps=4, guidance_scale=4.0, stop=2)
    assert rel_err(out, gold["latents_out"]) < 1e-5


def test_oracle_denoise_modes_reduce_to_plain(small_cfg):
    """reference_frame_count=0 and take_time large reduce the mode code paths to the plain loop."""
    sd = O.make_state_dict(small_cfg, seed=0)
    inp = small_inputs(small_cfg, seed=0)
    cond = {k: v for k, v in inp.items() if k not in ("sample", "timestep")}
    g = torch.Generator().manual_seed(7)
    lat = torch.randn(1, 3, 3, 16, 8, 12, generator=g)
    a = O.denoise(sd, small_cfg, lat, cond, steps=3, guidance_scale=4.0, stop=1)
    b = O.denoise(sd, small_cfg, lat, cond, steps=3, guidance_scale=4.0, stop=1, image_latents=lat, reference_frame_count=0)
    assert torch.equal(a, b)
    # diffusion forcing, step 0: only frame 0 is in schedule range, every frame is evaluated at timestep index 0
    c = O.denoise(sd, small_cfg, lat, cond, steps=3, guidance_scale=4.0, stop=1, diffusion_forcing=True)
    assert torch.equal(c[:, 1:], lat[:, 1:]) and rel_err(c[:, :1], a[:, :1]) < 1e-5


# ------------------------------------------------------------- oracle internal cross-checks
def test_sdpa_matches_torch_and_fp64():
    g = torch.Generator().manual_seed(1)
    q, k, v = (torch.randn(3, 2, 37, 64, generator=g) for _ in range(3))
    mask = torch.rand(3, 1, 37, 37, generator=g) > 0.3
    mask[..., 0] = True
    ref = F.scaled_dot_product_attention(q, k, v, attn_mask=mask)
    assert rel_err(O.sdpa(q, k, v, mask), ref) < 1e-5
    ref64 = O.sdpa(q.double(), k.double(), v.double(), mask)
    assert rel_err(O.sdpa(q, k, v, mask), ref64) < 1e-5


def test_flash_style_online_softmax_equals_naive():
    """The tiling / online-softmax algebra the HIP kernel uses (64-key tiles, running max in
    the log2 domain, deferred normalisation) restated in torch must equal naive softmax."""
    g = torch.Generator().manual_seed(2)
    L, d = 150, 64
    q, k, v = (torch.randn(L, d, generator=g) for _ in range(3))
    scale_log2 = d ** -0.5 * 1.4426950408889634
    m = torch.full((L,), -1e30)
    l = torch.zeros(L)
    o = torch.zeros(L, d)
    for k0 in range(0, L, 64):
        s = (q @ k[k0:k0 + 64].T) * scale_log2
        m_new = torch.maximum(m, s.max(-1).values)
        alpha = torch.exp2(m - m_new)
        p = torch.exp2(s - m_new[:, None])
        l = l * alpha + p.sum(-1)
        o = o * alpha[:, None] + p @ v[k0:k0 + 64]
        m = m_new
    out = o / l[:, None]
    ref = O.sdpa(q[None, None], k[None, None], v[None, None])[0, 0]
    assert rel_err(out, ref) < 1e-5


def test_layernorm_and_rmsnorm_restatements():
    g = torch.Generator().manual_seed(3)
    x = torch.randn(5, 7, 128, generator=g) * 3 + 1
    mean = x.mean(-1, keepdim=True)
    var = x.var(-1, unbiased=False, keepdim=True)
    assert rel_err(O.layer_norm_noaffine(x), (x - mean) / torch.sqrt(var + 1e-6)) < 1e-5
    w = torch.randn(64, generator=g)
    xx = torch.randn(2, 3, 4, 64, generator=g)
    ref = xx / torch.sqrt((xx * xx).mean(-1, keepdim=True) + 1e-5) * w
    assert rel_err(O.rms_norm(xx, w, 1e-5), ref) < 1e-5


def test_timesteps_sinusoid_layout():
    t = torch.tensor([0.0, 1.0, 500.0])
    e = O.timesteps_sinusoid(t, 8)
    f = torch.exp(-torch.log(torch.tensor(10000.0)) * torch.arange(4) / 4)
    assert torch.allclose(e[:, :4], torch.cos(t[:, None] * f), atol=1e-6)     # flip_sin_to_cos: cos first
    assert torch.allclose(e[:, 4:], torch.sin(t[:, None] * f), atol=1e-6)


def test_patch_embed_equals_im2col_gemm(small_cfg):
    """patchify column order used by the HIP im2col: col = (c*p + py)*p + px."""
    sd = O.make_state_dict(small_cfg, seed=0)
    x = torch.randn(4, 16, 8, 12)
    ref = O.patch_embed(sd, small_cfg, x)
    p = 2
    cols = x.reshape(4, 16, 4, p, 6, p).permute(0, 2, 4, 1, 3, 5).reshape(4 * 24, 64)
    w = sd["pos_embed.proj.weight"].reshape(-1, 64)
    y = cols @ w.T + sd["pos_embed.proj.bias"]
    pos = O.cropped_pos_embed(sd["pos_embed.pos_embed"], 4, 6, small_cfg["pos_embed_max_size"])[0]
    y = y.view(4, 24, -1) + pos
    assert rel_err(y, ref) < 1e-5


def test_vt_block_is_tokenwise_except_attention(small_cfg):
    """The HIP model never materialises the einops rearranges: every op of the VT block but
    the attention is per-token, so permuting tokens commutes with it.  Check on the oracle."""
    sd = O.make_state_dict(small_cfg, seed=0)
    x = torch.randn(6, 9, 128)
    perm = torch.randperm(9)
    a = O.vt_self_attention_block(sd, "temporal_transformer_blocks.0", 2, x)
    b = O.vt_self_attention_block(sd, "temporal_transformer_blocks.0", 2, x[:, perm])
    assert rel_err(b, a[:, perm]) < 1e-5


def test_flow_match_sigmas_properties():
    s = O.flow_match_sigmas(40, shift=3.0)
    assert s.shape == (41,) and s[-1] == 0 and abs(s[0].item() - 1.0) < 1e-6
    assert torch.all(s[:-1] > s[1:])
    # diffusers applies the shift to the training table (giving sigma_max/min) and again to the
    # linspace between them: sigma = 3 t / (1 + 2 t), t in linspace(1, 3e-3/(1+2e-3), n)
    t = torch.linspace(1.0, 0.003 / 1.002, 40)
    assert torch.allclose(s[:-1], 3 * t / (1 + 2 * t), atol=1e-6)


def test_flop_model_matches_survey():
    f = O.flops_per_forward(O.make_config(), 2, 16, 6, 32, 56)
    assert abs(f["total"] / 1e12 - 398.98) < 0.2
    assert abs(f["attention"] / 1e12 - 16.71) < 0.02
    from opendwm_amd.dit import model_flops
    g = model_flops(O.make_config(), 2, 16, 6, 32, 56)
    assert g["total"] == f["total"] and g["attention"] == f["attention"]


# --------------------------------------------------------------------- host-side logic
def _tokens(B, T, V, h, w):
    return torch.arange(B * T * V * h * w).view(B * T * V, h * w, 1)


@pytest.mark.parametrize("shape", [(2, 3, 4, 2, 5), (1, 16, 6, 16, 28)])
def test_rowmaps_equal_reference_rearranges(shape):
    """RowMap.rows() (the kernel's addressing, restated on the host) must enumerate exactly the
    rows the reference's einops.rearrange would copy (crossview_temporal_dit.py:290-361)."""
    from opendwm_amd import ops
    B, T, V, h, w = shape
    x = _tokens(B, T, V, h, w)
    cases = [
        (ops.rowmap_crossview_rowwise, "(bt v) (h w) c -> (bt h) (v w) c", dict(v=V, w=w)),
        (ops.rowmap_crossview_full, "(bt v) (h w) c -> bt (h v w) c", dict(v=V, w=w)),
        (ops.rowmap_temporal_rowwise, "(b t v) (h w) c -> (b v h) (t w) c", dict(b=B, v=V, w=w)),
        (ops.rowmap_temporal_full, "(b t v) hw c -> (b v) (t hw) c", dict(b=B, t=T)),
        (ops.rowmap_temporal_pointwise, "(b t v) hw c -> (b v hw) t c", dict(b=B, t=T)),
    ]
    for mk, pattern, kw in cases:
        rm = mk(B, T, V, h, w)
        ref = einops.rearrange(x, pattern, **kw)[..., 0]
        assert (rm.n_problems, rm.L0) == tuple(ref.shape), pattern
        assert torch.equal(rm.rows(), ref), pattern
    rm = ops.rowmap_identity(B * T * V, h * w)
    assert torch.equal(rm.rows(), x[..., 0])


def test_group_mask_equals_reference_expansion():
    """mode-1 mask semantics of dwm_attention_fwd == the repeat_interleave expansion of
    crossview_temporal_dit.py:301-305."""
    from opendwm_amd import ops
    B, T, V, h, w = 2, 3, 4, 2, 5
    g = torch.Generator().manual_seed(0)
    m = torch.rand(B, V, V, generator=g) > 0.4
    ref = m.repeat_interleave(w, 2).repeat_interleave(w, 1).repeat_interleave(T * h, 0)
    rm = ops.rowmap_crossview_rowwise(B, T, V, h, w)
    p = torch.arange(rm.n_problems)[:, None, None]
    lq = torch.arange(rm.L0)[None, :, None]
    lk = torch.arange(rm.L0)[None, None, :]
    mine = m[p // rm.p_per_mask, (lq // rm.group_size) % V, (lk // rm.group_size) % V]
    assert torch.equal(mine, ref)


def test_geglu_pack_matches_kernel_epilogue_contract():
    """DWM_EPI_GEGLU: within each 64 packed rows, row j < 32 is the value and row j + 32 the gate
    of output column (group*32 + j)."""
    from opendwm_amd.blocks import geglu_pack
    g = torch.Generator().manual_seed(0)
    x = torch.randn(5, 64, generator=g)
    w = torch.randn(256, 64, generator=g)
    b = torch.randn(256, generator=g)
    y = x @ w.T + b
    hv, gate = y.chunk(2, -1)
    ref = hv * F.gelu(gate)
    wp, bp = geglu_pack(w), geglu_pack(b)
    yp = (x @ wp.T + bp).view(5, -1, 2, 32)
    mine = (yp[:, :, 0] * F.gelu(yp[:, :, 1])).reshape(5, -1)
    assert rel_err(mine, ref) < 1e-6


def test_packed_prescaled_rms_weights_are_each_norm_s_own():
    """Attention.packed(): "rms_ps" / "rms_add_ps" = the q RMSNorm weights x head_dim^-1/2 log2(e) (blocks.PRESCALE_Q) | the k weights,
    per stream.  The scaled q weights are temporaries; a conversion cache keyed on tensor identity (STORE.bf's shadow) handed the
    context stream the main stream's q weights when the allocator reused the temporary (round 6, found by the two-rank tests)."""
    from opendwm_amd.blocks import Attention
    bf16 = torch.bfloat16
    a = Attention(128, 2, 64, bias=True, added_kv=True, qk_norm="rms_norm", eps=1e-6)
    g = torch.Generator().manual_seed(3)
    for n in (a.norm_q, a.norm_k, a.norm_added_q, a.norm_added_k):
        n.weight.data = 1 + 0.5 * torch.randn(64, generator=g)
    c = 64 ** -0.5 * 1.4426950408889634
    from opendwm_amd.blocks import STORE
    before = set(STORE._shadow)
    for _ in range(3):                                                # (fresh temporaries each time)
        a._pk = None
        pk = a.packed()
        for key, nq, nk in (("rms", a.norm_q, a.norm_k), ("rms_add", a.norm_added_q, a.norm_added_k)):
            assert torch.equal(pk[key], torch.cat([nq.weight.to(bf16).repeat(2), nk.weight.to(bf16).repeat(2)]))
            assert torch.equal(pk[key + "_ps"], torch.cat([(nq.weight.float() * c).to(bf16).repeat(2), nk.weight.to(bf16).repeat(2)]))
        # ... and no temporary may sit in the identity-keyed cache at all (the stale hit itself needs the allocator's cooperation)
        assert {k for k in STORE._shadow if k not in before} <= {id(p) for p in a.parameters()}
        STORE.bump()


def test_model_state_dict_keys_equal_reference_tree(small_cfg):
    from opendwm_amd.dit import DiTCrossviewTemporalConditionModel
    m = DiTCrossviewTemporalConditionModel(**small_cfg)
    sd = O.make_state_dict(small_cfg, seed=0)
    assert set(m.state_dict().keys()) == set(sd.keys())
    missing, unexpected = m.load_state_dict(sd)
    assert not missing and not unexpected
    for k, v in m.state_dict().items():
        assert v.shape == sd[k].shape, k
    # spot-check the names the reference / diffusers module tree uses
    for k in ("transformer_blocks.0.attn.add_k_proj.bias", "transformer_blocks.0.attn2.norm_q.weight",
              "transformer_blocks.3.norm1_context.linear.weight", "transformer_blocks.0.ff.net.0.proj.weight",
              "crossview_transformer_blocks.0.ff_in.net.2.weight", "temporal_transformer_blocks.1.attn1.to_out.0.bias",
              "view_mixers.0.mix_factor", "time_pos_embeds.0.linear_2.weight", "view_embedding.linear_1.weight",
              "pos_embed.proj.weight", "pos_embed.pos_embed", "norm_out.linear.weight", "proj_out.bias"):
        assert k in sd, k
    assert "transformer_blocks.3.attn.to_add_out.weight" not in sd      # context_pre_only last block


def test_model_pos_embed_buffer_equals_oracle_table(small_cfg):
    from opendwm_amd.dit import sincos_pos_embed_2d
    a = sincos_pos_embed_2d(128, 32, 16)
    b = O.make_pos_embed_table(128, 32, 16)
    assert torch.equal(a, b)


def test_model_rejects_cpu_inputs(small_cfg):
    """No silent CPU / eager fallback: a CPU call must raise."""
    from opendwm_amd.dit import DiTCrossviewTemporalConditionModel
    m = DiTCrossviewTemporalConditionModel(**small_cfg)
    inp = small_inputs(small_cfg)
    with pytest.raises(RuntimeError):
        m(inp.pop("sample"), inp.pop("timestep"), **inp)


def test_schedule_equals_oracle():
    from opendwm_amd.pipeline import FlowMatchEulerSchedule
    s = FlowMatchEulerSchedule(shift=3.0).set_timesteps(40)
    assert torch.allclose(s.sigmas, O.flow_match_sigmas(40, 3.0), atol=1e-7)
    assert torch.allclose(s.timesteps, s.sigmas[:-1] * 1000)


def test_padded_grid_shift_gemm_equals_conv2d():
    """Host restatement of the implicit-GEMM 3x3 convolution (padded token grid + 9 row shifts +
    tap-major weights) the HIP GEMM runs for the ImageAdapter == F.conv2d(padding=1)."""
    from opendwm_amd.ops import PaddedGrid
    g = torch.Generator().manual_seed(0)
    I, h, w, C, N = 3, 4, 6, 8, 5
    x = torch.randn(I, C, h, w, generator=g)
    wt = torch.randn(N, C, 3, 3, generator=g)
    ref = F.conv2d(x, wt, padding=1)                                   # [I, N, h, w]
    grid = PaddedGrid(I, h, w)
    xp = torch.zeros(grid.rows, C)
    idx = grid.interior_index()
    xp[idx] = x.permute(0, 2, 3, 1).reshape(-1, C)                      # token-major interior
    wp = wt.permute(0, 2, 3, 1).reshape(N, 9 * C)                       # [N, (dy, dx, c)]
    out = torch.zeros(grid.pixels, N)
    for t, sh in enumerate(grid.tap_shifts()):
        out += xp[idx + sh] @ wp[:, t * C:(t + 1) * C].T
    assert rel_err(out, ref.permute(0, 2, 3, 1).reshape(-1, N)) < 1e-5


def test_oracle_image_adapter_shapes(small_cfg):
    ac = dict(in_channels=6, channels=[128, 128, 128], is_downblocks=[True, False, False], num_res_blocks=2,
              downscale_factor=8, use_zero_convs=True)
    cfg = small_config(condition_image_adapter_config=ac)
    sd = O.make_state_dict(cfg, 0)
    feats = O.image_adapter(sd, cfg, torch.rand(2, 3, 3, 6, 64, 96))
    assert [tuple(f.shape) for f in feats] == [(2, 3, 3, 128, 4, 6)] * 3
    from opendwm_amd.dit import DiTCrossviewTemporalConditionModel
    m = DiTCrossviewTemporalConditionModel(**cfg)
    assert set(m.state_dict()) == set(sd)
    for k in ("condition_image_adapter.body.0.in_conv.weight", "condition_image_adapter.body.2.resnets.1.block2.bias",
              "condition_image_adapter.zero_convs.1.weight"):
        assert k in sd


def test_vae_from_pretrained_layout_and_keys(tmp_path):
    """The VAE drop-in loads a diffusers-style directory (config.json + safetensors) exactly as
    ctsd.py:953-959 calls it, with the diffusers AutoencoderKL key names."""
    import json
    from safetensors.torch import save_file
    from opendwm_amd.vae import AutoencoderKL
    vcfg = dict(in_channels=3, out_channels=3, latent_channels=16, block_out_channels=[64, 64, 128, 128],
                layers_per_block=2, norm_num_groups=16, scaling_factor=1.5305, shift_factor=0.0609,
                use_quant_conv=False, use_post_quant_conv=False)
    sd = O.make_vae_state_dict(vcfg, 0)
    d = tmp_path / "vae"
    d.mkdir()
    json.dump({**vcfg, "_class_name": "AutoencoderKL", "_diffusers_version": "0.31.0"}, open(d / "config.json", "w"))
    save_file({k: v.contiguous() for k, v in sd.items()}, str(d / "diffusion_pytorch_model.safetensors"))
    vae = AutoencoderKL.from_pretrained(str(tmp_path), subfolder="vae")
    got = vae.state_dict()
    assert set(got) == set(sd) and all(torch.equal(got[k], sd[k]) for k in sd)
    assert vae.config.scaling_factor == 1.5305 and len(vae.config.block_out_channels) == 4
    for k in ("decoder.mid_block.attentions.0.to_q.weight", "decoder.up_blocks.2.resnets.0.conv_shortcut.weight",
              "decoder.up_blocks.0.upsamplers.0.conv.bias", "encoder.down_blocks.1.downsamplers.0.conv.weight",
              "encoder.conv_out.bias", "decoder.conv_norm_out.weight"):
        assert k in sd
    with pytest.raises(RuntimeError):
        vae.decode(torch.zeros(1, 16, 8, 8))          # CPU call must fail loudly


def test_oracle_vae_round_trip_shapes():
    vcfg = dict(block_out_channels=(64, 64, 128, 128), layers_per_block=2, norm_num_groups=16, latent_channels=16)
    sd = O.make_vae_state_dict(vcfg, 0)
    x = torch.rand(2, 3, 32, 48) * 2 - 1
    mom = O.vae_encode_moments(sd, vcfg, x)
    assert mom.shape == (2, 32, 4, 6)
    img = O.vae_decode(sd, vcfg, mom[:, :16])
    assert img.shape == x.shape


# ---------------------------------------------------------------------------- C ABI
def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "dwm_hip.h")).read()
    return sorted(set(re.findall(r"^\s*(?:int|int64_t|const char\*)\s+(dwm_[a-z0-9_]+)\s*\(", hdr, flags=re.M)))


def test_library_builds_and_exports_every_declared_symbol():
    from opendwm_amd import _lib, build
    path = build.build()
    lib = ctypes.CDLL(path)
    declared = _declared_symbols()
    assert len(declared) >= 12
    for name in declared:
        assert hasattr(lib, name), name
    assert set(_lib.SIGNATURES) == set(declared)
    assert lib.dwm_abi_version() == _lib.ABI_VERSION
    lib.dwm_source_hash.restype = ctypes.c_char_p
    assert lib.dwm_source_hash().decode() == build.source_hash()      # the binary is the one of these sources


def test_ctypes_structs_match_header_field_order():
    """Field names of the ctypes mirrors must appear in the header structs in the same order."""
    from opendwm_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "dwm_hip.h")).read()
    for cname, cls in (("dwm_gemm_args", _lib.GemmArgs), ("dwm_attn_args", _lib.AttnArgs),
                       ("dwm_layernorm_args", _lib.LayerNormArgs), ("dwm_gemm_tn_args", _lib.GemmTnArgs)):
        body = hdr[hdr.index(f"typedef struct {cname}"):hdr.index(f"}} {cname};")]
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        pos = -1
        for fname, _ in cls._fields_:
            m = re.search(rf"[\s\*,]{fname}\s*(\[\d+\])?\s*[,;]", body[pos + 1:])
            assert m, (cname, fname)
            pos = pos + 1 + m.start()


def test_entry_points_validate_their_arguments_before_touching_the_device():
    """Argument checks of the C ABI run on the host, before any HIP call: they can be exercised without a GPU (fake, aligned
    device addresses; every call below must return its error code, not launch)."""
    from opendwm_amd import _lib, build
    lib = ctypes.CDLL(build.build())
    lib.dwm_gemm_tn.restype = ctypes.c_int
    lib.dwm_gemm_bf16.restype = ctypes.c_int
    fake = 1 << 20                                             # 16-byte aligned, never dereferenced on these paths

    def tn(**kw):
        g = _lib.GemmTnArgs()
        g.A = g.B = g.out = g.workspace = fake
        g.lda, g.ldb, g.ldo, g.M, g.N, g.C, g.b_rows = 256, 256, 256, 128, 256, 256, 128
        g.workspace_bytes = 1 << 30
        for k, v in kw.items():
            setattr(g, k, v)
        return lib.dwm_gemm_tn(ctypes.byref(g), None)

    assert tn(M=100, b_rows=100) != 0                          # contraction rows must be a multiple of 64
    assert tn(N=252) != 0 and tn(C=12) != 0                    # 16-byte rows
    assert tn(b_rows=64) != 0                                  # no taps: B must have a row for every row of A
    assert tn(ntaps=28) != 0
    assert tn(lda=128) != 0 and tn(ldo=128) != 0               # leading dimensions shorter than the rows
    assert tn(workspace_bytes=1024) != 0                       # not even one K range of partial tiles fits
    assert tn(split_k=64) != 0                                 # more ranges than the rule allows (>= 8 K steps each, <= 32)
    assert tn(A=fake + 2) != 0                                 # misaligned

    def nt(**kw):
        g = _lib.GemmArgs()
        g.A = g.W = g.C = fake
        g.lda, g.ldc, g.M, g.N, g.K = 256, 256, 256, 256, 256
        for k, v in kw.items():
            setattr(g, k, v)
        return lib.dwm_gemm_bf16(ctypes.byref(g), None)

    assert nt(tile=3) != 0 and nt(tile=-1) != 0                # tile configuration: 0 / 1 / 2
    assert nt(K=100) != 0 and nt(N=100) != 0
    assert nt(tile=2, C32=fake, epilogue=_lib.EPI_RESID, ldc32=256) != 0     # the fp32 residual stream has no 256 x 128 form
    assert nt(epilogue=99) != 0


def test_train_pair_construction_matches_oracle():
    """Host-side part of the train step (no GPU needed): sigma table, timestep / noisy-latent construction."""
    from opendwm_amd import pipeline as P
    sig = P.flow_match_train_sigmas(1000, 3.0)
    assert torch.equal(sig, O.flow_match_train_sigmas(3.0, 1000))
    assert sig.shape == (1000,) and sig[0] == 1.0 and abs(sig[-1].item() - 3e-3 / (1 + 2e-3)) < 1e-9
    assert torch.all(sig[1:] < sig[:-1])
    g = torch.Generator().manual_seed(3)
    idx = P.sample_timestep_indices((4096,), g)
    assert idx.min() >= 0 and idx.max() <= 999 and 400 < idx.float().mean() < 600      # logit-normal(0, 1): centred

    class _M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.zeros(4))
    tr = P.CTSDTrainer.__new__(P.CTSDTrainer)
    tr.sigmas, tr.num_train_timesteps, tr.weighting_scheme = sig, 1000, "logit_normal"
    lat = torch.randn(2, 3, 2, 4, 4, 6)
    idx = torch.tensor([10, 700])
    noise = torch.randn(lat.shape)
    noisy, ts, sg, _ = tr.make_training_pair(lat, timestep_indices=idx, noise=noise)
    assert ts.shape == (2, 3, 2) and torch.allclose(ts[:, 0, 0], sig[idx] * 1000)
    assert torch.allclose(noisy, sig[idx].view(2, 1, 1, 1, 1, 1) * noise + (1 - sig[idx].view(2, 1, 1, 1, 1, 1)) * lat)


# ------------------------------------------------------------------ SD 2.1 UNet (row a10)
def _unet_small():
    from tests.golden.make_golden import unet_small_config
    return unet_small_config()


def test_unet_oracle_matches_golden():
    from oracle import unet_oracle as U
    cfg = _unet_small()
    sd = U.make_unet_state_dict(cfg, 0)
    inp = U.make_unet_inputs(cfg, 2, 2, 3, 8, 16, text_len=10)
    gold = torch.load(os.path.join(GOLDEN, "unet_small.pt"))
    y = U.unet_forward(sd, cfg, **inp)
    assert y.shape == (2, 2, 3, 4, 8, 16) and rel_err(y, gold["output"]) < 1e-5
    cond = {k: v for k, v in inp.items() if k not in ("sample", "timesteps")}
    out = U.unet_denoise(sd, cfg, inp["sample"][:1], cond, steps=4, guidance_scale=3.0, stop=2)
    assert rel_err(out, gold["denoise_2steps"]) < 1e-5


def test_unet_oracle_pieces_equal_torch_modules():
    """the restated diffusers blocks against torch.nn modules wired the documented way (same weights)"""
    from oracle import unet_oracle as U
    g = torch.Generator().manual_seed(0)
    C, Co, E = 64, 128, 96
    x, temb = torch.randn(3, C, 6, 8, generator=g), torch.randn(3, E, generator=g)
    n1, c1, tp = torch.nn.GroupNorm(32, C, eps=1e-5), torch.nn.Conv2d(C, Co, 3, padding=1), torch.nn.Linear(E, Co)
    n2, c2, cs = torch.nn.GroupNorm(32, Co, eps=1e-5), torch.nn.Conv2d(Co, Co, 3, padding=1), torch.nn.Conv2d(C, Co, 1)
    sd = {}
    for name, m in (("norm1", n1), ("conv1", c1), ("time_emb_proj", tp), ("norm2", n2), ("conv2", c2), ("conv_shortcut", cs)):
        for k, v in m.state_dict().items():
            sd[f"r.{name}.{k}"] = v
    silu = torch.nn.functional.silu
    h = c1(silu(n1(x))) + tp(silu(temb))[:, :, None, None]
    ref = cs(x) + c2(silu(n2(h)))
    assert rel_err(U.resnet_block_2d(sd, "r", x, temb, 1e-5), ref) < 1e-6
    # temporal resnet: Conv3d (3,1,1) over T, temb [N, T, E]
    x5, t5 = torch.randn(2, Co, 5, 4, 6, generator=g), torch.randn(2, 5, E, generator=g)
    m1, k1, tq = torch.nn.GroupNorm(32, Co, eps=1e-5), torch.nn.Conv3d(Co, Co, (3, 1, 1), padding=(1, 0, 0)), torch.nn.Linear(E, Co)
    m2, k2 = torch.nn.GroupNorm(32, Co, eps=1e-5), torch.nn.Conv3d(Co, Co, (3, 1, 1), padding=(1, 0, 0))
    sd = {}
    for name, m in (("norm1", m1), ("conv1", k1), ("time_emb_proj", tq), ("norm2", m2), ("conv2", k2)):
        for k, v in m.state_dict().items():
            sd[f"t.{name}.{k}"] = v
    h = k1(silu(m1(x5))) + tq(silu(t5)).permute(0, 2, 1)[:, :, :, None, None]
    assert rel_err(U.temporal_resnet_block(sd, "t", x5, t5, 1e-5), x5 + k2(silu(m2(h)))) < 1e-6


def test_unet_model_state_dict_keys_equal_oracle_tree():
    """the HIP model's module tree carries exactly the reference key names / shapes the oracle uses; SD 2.1 checkpoints are
    renamed the reference's way (crossview_temporal_unet.py:358-373)"""
    from oracle import unet_oracle as U
    from opendwm_amd.unet import UNetCrossviewTemporalConditionModel
    cfg = _unet_small()
    m = UNetCrossviewTemporalConditionModel(**cfg)
    shapes = U.unet_param_shapes(cfg)
    sd = m.state_dict()
    assert set(sd) == set(shapes)
    assert all(tuple(sd[k].shape) == tuple(shapes[k]) for k in shapes)
    conv = UNetCrossviewTemporalConditionModel.try_to_convert_state_dict(
        {"down_blocks.0.resnets.1.conv1.weight": 1, "down_blocks.0.attentions.0.norm.weight": 2})
    assert set(conv) == {"down_blocks.0.resnets.1.spatial_res_block.conv1.weight", "down_blocks.0.attentions.0.norm.weight"}
    full = U.make_unet_config()
    n = sum(torch.Size(s).numel() for s in U.unet_param_shapes(full).values())
    assert 1.9e9 < n < 1.95e9
    # module protocol the pipeline uses on the model (ctsd.py:867-875, 1462)
    assert m.depth_net is None and m.gradient_checkpointing is False
    m.enable_gradient_checkpointing()
    assert m.gradient_checkpointing is True
    with pytest.raises(RuntimeError):             # no CPU / PyTorch fallback, in train mode either
        m.train()(torch.zeros(1, 1, 3, 4, 8, 16), torch.zeros(1, 1, 3), encoder_hidden_states=torch.zeros(1, 1, 3, 10, cfg["cross_attention_dim"]))


def test_dpm_solver_tables_and_last_step():
    from oracle import unet_oracle as U
    ts, sig = U.dpm_solver_tables(50)
    assert ts[0] == 999 and ts[-1] == 20 and sig[-1] == 0 and torch.all(sig[1:] < sig[:-1])
    kx, ko, A, B, Cc = U.dpm_solver_coefficients(sig, 49, "v_prediction")
    assert (A, B, Cc) == (0.0, 1.0, 0.0)
    # a constant x0 prediction is a fixed point of the data-prediction solver: x' - x0 = A (x - x0) for every order
    for i in (0, 5, 30):
        kx, ko, A, B, Cc = U.dpm_solver_coefficients(sig, i, "epsilon")
        a_t = 1.0 / (float(sig[i + 1]) ** 2 + 1) ** 0.5
        a_s = 1.0 / (float(sig[i]) ** 2 + 1) ** 0.5
        # exactness for x = alpha x0 + sigma eps with the true (constant) x0: the update lands on the same form at t
        x0, eps = 0.7, -0.3
        x = a_s * x0 + float(sig[i]) * a_s * eps
        xn = A * x + (B + Cc) * x0
        assert abs(xn - (a_t * x0 + float(sig[i + 1]) * a_t * eps)) < 1e-9


def test_split_weight_planes_and_tap_groups():
    """host side of dwm_gemm_f32's weight operand (opendwm_amd.ops.split_weight): hi + lo planes reproduce the fp32 weight to 2^-16,
    the per-tap layout is [hi_t | lo_t | hi_t], and more than 9 taps (the 27 of a causal 3x3x3 convolution) are laid out as groups of
    9 taps, each group a contiguous [N, 9 * 3C] matrix - what the kernel's one-main-loop-launch-per-group walk expects"""
    from opendwm_amd import ops
    g = torch.Generator().manual_seed(0)
    N, Cc = 24, 64
    w = torch.randn(N, 27 * Cc, generator=g)
    hi = w.to(torch.bfloat16)
    lo = (w - hi.float()).to(torch.bfloat16)
    assert ((hi.float() + lo.float()) - w).abs().max() <= w.abs().max() * 2.0 ** -16
    ws1 = ops.split_weight(w[:, :Cc].contiguous(), 1, cache=False)                     # plain: [hi | lo | hi]
    assert ws1.shape == (N, 3 * Cc) and torch.equal(ws1[:, :Cc], hi[:, :Cc]) and torch.equal(ws1[:, Cc:2 * Cc], lo[:, :Cc]) \
        and torch.equal(ws1[:, 2 * Cc:], hi[:, :Cc])
    ws9 = ops.split_weight(w[:, :9 * Cc].contiguous(), 9, cache=False).view(N, 9, 3, Cc)
    assert torch.equal(ws9[:, :, 0], hi.view(N, 27, Cc)[:, :9]) and torch.equal(ws9[:, :, 1], lo.view(N, 27, Cc)[:, :9]) \
        and torch.equal(ws9[:, :, 2], ws9[:, :, 0])
    ws27 = ops.split_weight(w, 27, cache=False)
    assert ws27.shape == (N, 3 * 27 * Cc) and ws27.is_contiguous()
    flat = ws27.reshape(-1)
    per_group = N * 9 * 3 * Cc
    for grp in range(3):
        blk = flat[grp * per_group:(grp + 1) * per_group].view(N, 9, 3, Cc)
        assert torch.equal(blk[:, :, 0], hi.view(N, 27, Cc)[:, 9 * grp:9 * grp + 9])
        assert torch.equal(blk[:, :, 1], lo.view(N, 27, Cc)[:, 9 * grp:9 * grp + 9])
        assert torch.equal(blk[:, :, 2], blk[:, :, 0])
