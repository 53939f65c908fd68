"""GPU leg (`-m gpu`): parity of the HIP path (through the C ABI) against the oracle /
plain fp32 torch references, on seeded inputs.  Tolerances follow BASELINE.json's
north_star: <= 2e-2 relative (Frobenius) for the bf16 path against the fp32 reference;
individual kernels are held to a much tighter bound (bf16 output rounding, 2^-9).
Every measured error is appended to gpurun_out/gpu_parity.log."""
import json
import os

import pytest
import torch
import torch.nn.functional as F

from oracle import ctsd_oracle as O
from tests.common import GOLDEN, rel_err, small_config, small_inputs, to_dev

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
bf16 = torch.bfloat16
TOL_KERNEL = 6e-3       # one bf16 rounding of the output (2^-9 max, ~1.1e-3 rms) plus fp32 accumulation order
TOL_MODEL = 2e-2        # BASELINE.json north_star, bf16


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
    _lib.load()          # fail loudly if libdwm_hip.so is missing: there is no fallback path
    return torch.device("cuda:0")


def _rand(shape, dev, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dev).to(bf16)


# ---------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (512, 768, 128), (300, 64, 192), (1000, 1536, 1536),
                                   (192, 4608, 1536), (4096, 6144, 1536), (77, 8, 64)])
def test_gemm_plain(dev, M, N, K):
    from opendwm_amd import ops
    a, w, b = _rand((M, K), dev, 1), _rand((N, K), dev, 2, K ** -0.5), _rand((N,), dev, 3)
    ref = a.float() @ w.float().T + b.float()
    out = ops.gemm(a, w, b)
    e = rel_err(out, ref)
    # asymmetric operands: a transposed / permuted C write cannot pass this
    _log("gemm_plain", M=M, N=N, K=K, rel=e)
    assert e < TOL_KERNEL
    out2 = ops.gemm(a, w, None, act=ops.ACT_GELU_TANH)
    assert rel_err(out2, F.gelu(a.float() @ w.float().T, approximate="tanh")) < TOL_KERNEL
    out3 = ops.gemm(a, w, b, act=ops.ACT_SILU)
    assert rel_err(out3, F.silu(ref)) < TOL_KERNEL


def test_gemm_strided_a_and_identity(dev):
    """A as a column slice of a wider buffer (lda > K) and A = I with asymmetric W
    (transpose-detecting, cdna guide §3)."""
    from opendwm_amd import ops
    big = _rand((512, 512), dev, 4)
    a = big[:, 128:256]
    w = _rand((320, 128), dev, 5)
    assert rel_err(ops.gemm(a, w), a.float() @ w.float().T) < TOL_KERNEL
    eye = torch.eye(256, device=dev, dtype=bf16)
    w2 = (torch.arange(256 * 256, device=dev).view(256, 256) % 251).to(bf16)
    assert torch.equal(ops.gemm(eye, w2).float(), w2.float().T.contiguous())


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (700, 12288, 1536), (1000, 1024, 128)])
def test_gemm_geglu(dev, M, N, K):
    from opendwm_amd import ops
    from opendwm_amd.blocks import geglu_pack
    a, w, b = _rand((M, K), dev, 1), _rand((N, K), dev, 2, K ** -0.5), _rand((N,), dev, 3)
    y = a.float() @ w.float().T + b.float()
    hv, g = y.chunk(2, -1)
    ref = hv * F.gelu(g)
    out = ops.gemm(a, geglu_pack(w), geglu_pack(b), epilogue=ops.EPI_GEGLU)
    e = rel_err(out, ref)
    _log("gemm_geglu", M=M, N=N, K=K, rel=e)
    assert out.shape == (M, N // 2) and e < TOL_KERNEL


@pytest.mark.parametrize("M,N,K,rpg", [(448 * 4, 1536, 1536, 448), (600, 256, 128, 100), (154 * 3, 1536, 6144, 154)])
def test_gemm_resid(dev, M, N, K, rpg):
    from opendwm_amd import ops
    a, w, b = _rand((M, K), dev, 1), _rand((N, K), dev, 2, K ** -0.5), _rand((N,), dev, 3)
    groups = (M + rpg - 1) // rpg
    gate, res, blend = _rand((groups, N), dev, 4), _rand((M, N), dev, 5), _rand((M, N), dev, 6)
    alpha = torch.rand(groups, device=dev)
    y = a.float() @ w.float().T + b.float()
    rows = torch.arange(M, device=dev) // rpg
    ref1 = res.float() + gate.float()[rows] * y
    out1 = ops.gemm(a, w, b, epilogue=ops.EPI_RESID, gate=gate, rows_per_gate=rpg, res=res)
    e1 = rel_err(out1, ref1)
    al = alpha[rows][:, None]
    ref2 = al * blend.float() + (1 - al) * (res.float() + y)
    out2 = ops.gemm(a, w, b, epilogue=ops.EPI_RESID, res=res, blend=blend, alpha=alpha, rows_per_alpha=rpg)
    e2 = rel_err(out2, ref2)
    # in-place forms used by the model: out aliases res / blend
    r2 = res.clone()
    ops.gemm(a, w, b, epilogue=ops.EPI_RESID, gate=gate, rows_per_gate=rpg, res=r2, out=r2)
    bl = blend.clone()
    ops.gemm(a, w, b, epilogue=ops.EPI_RESID, res=res, blend=bl, alpha=alpha, rows_per_alpha=rpg, out=bl)
    # row-modulo residual (pos-embed add of the patch embedding)
    pos = _rand((rpg, N), dev, 7)
    out3 = ops.gemm(a, w, b, epilogue=ops.EPI_RESID, res=pos, res_mod=rpg)
    ref3 = y + pos.float()[torch.arange(M, device=dev) % rpg]
    # plain residual add, and the one-row-per-group residual (res_mod < 0: the time-embedding term of the UNet's resnets) - with
    # the gated and the blended form above the four operand sets that have compile-time forms of the kernel (RS)
    out4 = ops.gemm(a, w, b, epilogue=ops.EPI_RESID, res=res)
    e4 = rel_err(out4, res.float() + y)
    out5 = ops.gemm(a, w, b, epilogue=ops.EPI_RESID, res=gate, res_mod=-rpg)
    e5 = rel_err(out5, y + gate.float()[rows])
    # ... and the same values from the run-time form (an activation keeps a call out of the FAST kernels)
    out6 = ops.gemm(a, w, b, epilogue=ops.EPI_RESID, res=gate, res_mod=-rpg, act=ops.ACT_RELU)
    e6 = rel_err(out6, torch.relu(y) + gate.float()[rows])
    _log("gemm_resid", M=M, N=N, K=K, rel_gate=e1, rel_blend=e2, rel_mod=rel_err(out3, ref3), rel_plain=e4, rel_row_per_group=e5,
         rel_row_per_group_general=e6)
    assert e1 < TOL_KERNEL and e2 < TOL_KERNEL and rel_err(out3, ref3) < TOL_KERNEL and max(e4, e5, e6) < TOL_KERNEL
    assert torch.equal(r2, out1) and torch.equal(bl, out2)


@pytest.mark.parametrize("M,heads,K,biased", [(602, 24, 1536, True), (300, 2, 128, False)])
def test_gemm_rmshead(dev, M, heads, K, biased):
    from opendwm_amd import ops
    D = heads * 64
    a, w = _rand((M, K), dev, 1), _rand((3 * D, K), dev, 2, K ** -0.5)
    b = _rand((3 * D,), dev, 3) if biased else None
    wq, wk = _rand((64,), dev, 4) * 0.2 + 1, _rand((64,), dev, 5) * 0.2 + 1
    y = a.float() @ w.float().T + (b.float() if biased else 0)
    q, k, v = y.view(M, 3, heads, 64).unbind(1)
    ref = torch.stack([O.rms_norm(q, wq.float(), 1e-6), O.rms_norm(k, wk.float(), 1e-6), v], 1).reshape(M, 3 * D)
    rms = torch.cat([wq.repeat(heads), wk.repeat(heads)]).contiguous()
    out = ops.gemm(a, w, b, epilogue=ops.EPI_RMSHEAD, rms_w=rms, rms_ncols=2 * D, rms_eps=1e-6)
    e = rel_err(out, ref)
    _log("gemm_rmshead", M=M, heads=heads, rel=e)
    assert e < TOL_KERNEL
    # stand-alone kernel agrees
    y16 = (a.float() @ w.float().T + (b.float() if biased else 0)).to(bf16)
    qk = y16[:, :2 * D].contiguous()
    ops.rmsnorm_heads_(qk, rms, 1e-6)
    assert rel_err(qk, ref[:, :2 * D]) < TOL_KERNEL


@pytest.mark.parametrize("M,N,K", [(256, 128, 64), (700, 320, 128), (1000, 1536, 1536), (2048, 640, 2880), (77, 8, 64), (4096, 4608, 1536),
                                   (513, 1160, 6144)])
def test_gemm_tile_256x128_equals_256x256(dev, M, N, K):
    """Tile configuration 2 (256 x 128 x 32, two 4-wave workgroups per CU) accumulates every output element in the same
    order as the 256 x 256 x 64 tile (the same MFMA K steps one after the other), so all epilogues must agree BIT FOR BIT;
    the plain form is also held against the fp32 product."""
    from opendwm_amd import ops as ops_
    from opendwm_amd.blocks import geglu_pack
    import functools, types
    # split_k = 1: small grids with a long K would otherwise take the split-K path (always 256 x 256 tiles) in both calls
    ops = types.SimpleNamespace(**{k: getattr(ops_, k) for k in dir(ops_) if k.isupper()}, gemm=functools.partial(ops_.gemm, split_k=1))
    a, w, b = _rand((M, K), dev, 1), _rand((N, K), dev, 2, K ** -0.5), _rand((N,), dev, 3)
    ref = a.float() @ w.float().T + b.float()
    o2 = ops.gemm(a, w, b, tile=2)
    e = rel_err(o2, ref)
    _log("gemm_tile2_plain", M=M, N=N, K=K, rel=e)
    assert e < TOL_KERNEL
    assert torch.equal(o2, ops.gemm(a, w, b, tile=1))
    for act in (ops.ACT_GELU_TANH, ops.ACT_SILU):
        assert torch.equal(ops.gemm(a, w, b, act=act, tile=2), ops.gemm(a, w, b, act=act, tile=1))
    rpg = 100
    groups = (M + rpg - 1) // rpg
    gate, res, blend = _rand((groups, N), dev, 4), _rand((M, N), dev, 5), _rand((M, N), dev, 6)
    alpha = torch.rand(groups, device=dev)
    pos = _rand((rpg, N), dev, 7)
    for kw in (dict(gate=gate, rows_per_gate=rpg, res=res), dict(res=res, blend=blend, alpha=alpha, rows_per_alpha=rpg),
               dict(res=pos, res_mod=rpg), dict(res=res, act=ops.ACT_SILU)):
        o1 = ops.gemm(a, w, b, epilogue=ops.EPI_RESID, tile=1, **kw)
        assert torch.equal(ops.gemm(a, w, b, epilogue=ops.EPI_RESID, tile=2, **kw), o1), kw.keys()
    r2 = res.clone()                                # in place over the residual
    ops.gemm(a, w, b, epilogue=ops.EPI_RESID, gate=gate, rows_per_gate=rpg, res=r2, out=r2, tile=2)
    assert torch.equal(r2, ops.gemm(a, w, b, epilogue=ops.EPI_RESID, gate=gate, rows_per_gate=rpg, res=res, tile=1))
    if N % 128 == 0:
        wg, bg = geglu_pack(w), geglu_pack(b)
        assert torch.equal(ops.gemm(a, wg, bg, epilogue=ops.EPI_GEGLU, tile=2), ops.gemm(a, wg, bg, epilogue=ops.EPI_GEGLU, tile=1))
    if N % 192 == 0:
        D = N // 3
        rms = (_rand((2 * D,), dev, 8) * 0.2 + 1).contiguous()
        kw = dict(epilogue=ops.EPI_RMSHEAD, rms_w=rms, rms_ncols=2 * D, rms_eps=1e-6)
        assert torch.equal(ops.gemm(a, w, b, tile=2, **kw), ops.gemm(a, w, b, tile=1, **kw))


@pytest.mark.parametrize("I,h,w,C,N", [(3, 16, 28, 128, 320), (2, 4, 6, 64, 192), (7, 9, 5, 320, 640)])
def test_gemm_tile_256x128_implicit_conv(dev, I, h, w, C, N):
    """the implicit 3x3 convolution (tap-shifted LDS-DMA sources, K steps of 32 inside a tap) and the padded output grid on
    the 256 x 128 tile (split_k = 1: the small test grids would otherwise take the split-K path, which keeps the 256 x 256 tile)"""
    from opendwm_amd import ops
    grid = ops.PaddedGrid(I, h, w)
    x = _rand((I, C, h, w), dev, 1)
    wt, b = _rand((N, C, 3, 3), dev, 2, (9 * C) ** -0.5), _rand((N,), dev, 3)
    ref = F.silu(F.conv2d(x.float(), wt.float(), b.float(), padding=1)).permute(0, 2, 3, 1).reshape(-1, N)
    idx = grid.interior_index().to(dev)
    xp = torch.zeros((grid.rows, C), dtype=bf16, device=dev)
    xp[idx] = x.permute(0, 2, 3, 1).reshape(-1, C)
    wp = wt.permute(0, 2, 3, 1).reshape(N, 9 * C).contiguous()
    o2 = ops.gemm(xp, wp, b, act=ops.ACT_SILU, a_grid=grid, conv3x3=True, tile=2, split_k=1)
    e = rel_err(o2, ref)
    _log("gemm_tile2_conv3x3", I=I, h=h, w=w, C=C, N=N, rel=e)
    assert e < TOL_KERNEL
    assert torch.equal(o2, ops.gemm(xp, wp, b, act=ops.ACT_SILU, a_grid=grid, conv3x3=True, tile=1, split_k=1))
    w1 = _rand((C, N), dev, 4, N ** -0.5)
    r1, r2 = xp.clone(), xp.clone()
    ops.gemm(o2, w1, None, epilogue=ops.EPI_RESID, res=r1, out=r1, c_grid=grid, tile=1, split_k=1)
    ops.gemm(o2, w1, None, epilogue=ops.EPI_RESID, res=r2, out=r2, c_grid=grid, tile=2)
    assert torch.equal(r1, r2)


@pytest.mark.parametrize("M,N,K,split", [(300, 1280, 4096, 0), (72, 1280, 5120, 0), (512, 512, 2048, 5), (1536, 1536, 8192, 0),
                                         (77, 8, 1024, 2)])
def test_gemm_split_k(dev, M, N, K, split):
    """Split-K path (small tile grid, long K): fp32 partial tiles + ordered reduction + epilogue in the finishing
    kernel.  Same references as the single-pass path, every RESID form incl. the in-place ones, and the result must
    not depend on scheduling (bit-equal across runs)."""
    from opendwm_amd import ops
    a, w, b = _rand((M, K), dev, 1), _rand((N, K), dev, 2, K ** -0.5), _rand((N,), dev, 3)
    y = a.float() @ w.float().T + b.float()
    out = ops.gemm(a, w, b, split_k=split)
    one = ops.gemm(a, w, b, split_k=1)
    e, e1 = rel_err(out, y), rel_err(one, y)
    assert e < TOL_KERNEL and e1 < TOL_KERNEL
    assert torch.equal(out, ops.gemm(a, w, b, split_k=split))
    assert rel_err(ops.gemm(a, w, b, act=ops.ACT_GELU_TANH, split_k=split), F.gelu(y, approximate="tanh")) < TOL_KERNEL
    rpg = 100
    groups = (M + rpg - 1) // rpg
    gate, res, blend = _rand((groups, N), dev, 4), _rand((M, N), dev, 5), _rand((M, N), dev, 6)
    alpha = torch.rand(groups, device=dev)
    rows = torch.arange(M, device=dev) // rpg
    al = alpha[rows][:, None]
    r2 = res.clone()
    ops.gemm(a, w, b, epilogue=ops.EPI_RESID, gate=gate, rows_per_gate=rpg, res=r2, out=r2, split_k=split)
    e2 = rel_err(r2, res.float() + gate.float()[rows] * y)
    bl = blend.clone()
    ops.gemm(a, w, b, epilogue=ops.EPI_RESID, res=res, blend=bl, alpha=alpha, rows_per_alpha=rpg, out=bl, split_k=split)
    e3 = rel_err(bl, al * blend.float() + (1 - al) * (res.float() + y))
    per = _rand((groups, N), dev, 7)
    e4 = rel_err(ops.gemm(a, w, b, epilogue=ops.EPI_RESID, res=per, res_mod=-rpg, split_k=split), y + per.float()[rows])
    _log("gemm_split_k", M=M, N=N, K=K, split=split, rel=e, rel_single_pass=e1, rel_gate=e2, rel_blend=e3, rel_per_image=e4)
    assert e2 < TOL_KERNEL and e3 < TOL_KERNEL and e4 < TOL_KERNEL
    if split > 1:
        from opendwm_amd.blocks import geglu_pack
        with pytest.raises(RuntimeError):
            ops.gemm(a, geglu_pack(w) if N % 64 == 0 else w, None, epilogue=ops.EPI_GEGLU if N % 64 == 0 else ops.EPI_RMSHEAD,
                     split_k=split)


def test_gemm_split_k_implicit_conv(dev):
    """the lowest UNet level: few pixels, K = 9 taps x C; padded output grid with an in-place residual"""
    from opendwm_amd import ops
    I, h, w, C, N = 4, 4, 7, 1280, 256
    grid = ops.PaddedGrid(I, h, w)
    x = _rand((I, C, h, w), dev, 1)
    wt, b = _rand((N, C, 3, 3), dev, 2, (9 * C) ** -0.5), _rand((N,), dev, 3)
    ref = F.conv2d(x.float(), wt.float(), b.float(), padding=1).permute(0, 2, 3, 1).reshape(-1, N)
    idx = grid.interior_index().to(dev)
    xp = torch.zeros((grid.rows, C), dtype=bf16, device=dev)
    xp[idx] = x.permute(0, 2, 3, 1).reshape(-1, C)
    wp = wt.permute(0, 2, 3, 1).reshape(N, 9 * C).contiguous()
    outs = [ops.gemm(xp, wp, b, a_grid=grid, conv3x3=True, split_k=sk) for sk in (1, 0, 6)]
    errs = [rel_err(o, ref) for o in outs]
    res = torch.zeros((grid.rows, N), dtype=bf16, device=dev)
    res[idx] = _rand((I * h * w, N), dev, 4)
    want = res[idx].float() + ref
    ops.gemm(xp, wp, b, a_grid=grid, conv3x3=True, epilogue=ops.EPI_RESID, res=res, out=res, c_grid=grid, split_k=6)
    border = torch.ones(grid.rows, dtype=torch.bool, device=dev)
    border[idx] = False
    e2 = rel_err(res[idx], want)
    _log("gemm_split_k_conv", rel_single=errs[0], rel_auto=errs[1], rel_6=errs[2], rel_padded_out=e2)
    assert max(errs) < TOL_KERNEL and e2 < TOL_KERNEL and torch.count_nonzero(res[border]) == 0


def test_gemm_rejects_bad_arguments(dev):
    from opendwm_amd import ops
    a, w = _rand((64, 100), dev, 1), _rand((64, 100), dev, 2)
    with pytest.raises(RuntimeError):
        ops.gemm(a, w)                      # K % 64 != 0
    with pytest.raises(RuntimeError):
        ops.gemm(a.float(), w)              # dtype
    with pytest.raises(RuntimeError):
        ops.gemm(a.cpu(), w.cpu())          # no CPU path


# ----------------------------------------------------------------------------- attention
def test_tr_read_probe(dev):
    """Hardware semantics of ds_read_b64_tr_b16 as the attention kernel assumes them: within
    each 16-lane group the 16 addressed 8-byte rows form a 4x16 matrix M[u >> 2][4 (u & 3) + e]
    and lane t receives column t: result[j] = M[j][t]."""
    from opendwm_amd import ops
    lane = torch.arange(64)
    u, g = lane & 15, lane >> 4
    # dense case: group g reads 128 contiguous bytes at g*128
    offs = (g * 128 + u * 8).to(dev)
    out = ops.tr_probe(offs).cpu().long()
    exp = torch.stack([g * 64 + j * 16 + u for j in range(4)], 1)
    _log("tr_probe_dense", ok=bool(torch.equal(out, exp)), got=out[:20].tolist())
    assert torch.equal(out, exp)
    # strided rows (row stride 256 B), as the V tile uses them
    offs = (g * 1024 + (u >> 2) * 256 + (u & 3) * 8).to(dev)
    out = ops.tr_probe(offs).cpu().long()
    exp = torch.stack([g * 512 + j * 128 + u for j in range(4)], 1)
    _log("tr_probe_strided", ok=bool(torch.equal(out, exp)), got=out[:20].tolist())
    assert torch.equal(out, exp)


def _attn_ref(q, k, v, rows, heads, mask=None, q1=None, k1=None, v1=None):
    """q,k,v [R, heads*64] fp32; rows [P, L0] gather; returns (o0 scattered [R, D], o1)."""
    P, L0 = rows.shape
    D = heads * 64
    def gather(x, x1):
        g = x[rows.reshape(-1)].view(P, L0, heads, 64)
        if x1 is not None:
            g = torch.cat([g, x1.view(P, -1, heads, 64)], 1)
        return g.transpose(1, 2)
    Q, K, V = gather(q, q1), gather(k, k1), gather(v, v1)
    m = None if mask is None else mask[:, None]
    o = O.sdpa(Q, K, V, m).transpose(1, 2).reshape(P, -1, D)
    o0 = torch.zeros_like(q)
    o0[rows.reshape(-1)] = o[:, :L0].reshape(-1, D)
    o1 = None if q1 is None else o[:, L0:].reshape(-1, D)
    return o0, o1


# auto; bits 0-3: queries per wave of the tiled kernel (1: 32, 2: 64) / compute waves of the resident kernel (the others are pure
# loader waves); the same with the online-softmax fallback forced (bit 4); several heads per workgroup / item (bits 8-11: the head
# loop of the resident kernel - the next head's copy by loader waves behind the computing waves' progress); the tiled kernel (bit 5)
ATTN_VARIANTS = [0, 1, 2, 16 | 1, 16 | 8, (3 << 8) | 12, (2 << 8) | 5, 32 | 1, 32 | 2]


def _variant_for(variant, heads):
    """heads per workgroup (bits 8-11) must divide the head count: fall back to all heads (or none) where it does not"""
    hs = (variant >> 8) & 15
    if hs and heads % hs:
        variant = (variant & ~0xF00) | ((heads if heads <= 15 else 0) << 8)
    return variant


@pytest.mark.parametrize("variant", ATTN_VARIANTS)
@pytest.mark.parametrize("I,N,Lc,heads", [(3, 448, 154, 24), (2, 100, 0, 2), (2, 64, 10, 2), (5, 16, 0, 2), (1, 300, 3, 3)])
def test_attention_joint(dev, variant, I, N, Lc, heads):
    from opendwm_amd import ops
    D = heads * 64
    qkv = _rand((I * N, 3 * D), dev, 1)
    cqkv = _rand((I * Lc, 3 * D), dev, 2) if Lc else None
    out = torch.zeros((I * N, D), dtype=bf16, device=dev)
    cout = torch.zeros((I * Lc, D), dtype=bf16, device=dev) if Lc else None
    rm = ops.rowmap_identity(I, N)
    kw = {}
    if Lc:
        kw = dict(q1=cqkv[:, :D], k1=cqkv[:, D:2 * D], v1=cqkv[:, 2 * D:], out1=cout)
    ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out, rm, heads, variant=_variant_for(variant, heads), **kw)
    f, cf = qkv.float(), (cqkv.float() if Lc else None)
    r0, r1 = _attn_ref(f[:, :D], f[:, D:2 * D], f[:, 2 * D:], rm.rows().to(dev), heads,
                       q1=cf[:, :D] if Lc else None, k1=cf[:, D:2 * D] if Lc else None,
                       v1=cf[:, 2 * D:] if Lc else None)
    e0 = rel_err(out, r0)
    e1 = rel_err(cout, r1) if Lc else 0.0
    _log("attention_joint", variant=variant, I=I, N=N, Lc=Lc, heads=heads, rel0=e0, rel1=e1)
    assert e0 < TOL_KERNEL and e1 < TOL_KERNEL


@pytest.mark.parametrize("scale", [1.0, 8.0], ids=["unit_scores", "huge_scores_fallback"])
@pytest.mark.parametrize("I,N,Lc,heads", [(2, 448, 154, 6), (2, 448, 0, 6), (2, 608, 0, 3), (2, 97, 0, 4), (1, 64, 0, 2), (2, 200, 33, 2),
                                          (1, 575, 0, 2), (3, 33, 32, 3)])
def test_attention_resident_forms(dev, scale, I, N, Lc, heads):
    """attn_res_kernel (K / V of one head resident in LDS; unmasked self-attention, 64 <= L <= 608): one and all heads per
    item (head loop: the next head's copy in two parts - refill point inside the last round, rest after it - and L2 touches),
    one and two segments, sequence lengths that end in a ragged 32-key step / a full 64-key step / the LDS limit, two rounds of
    query tiles per wave (L = 602 / 608: 19 tiles), the maximum-free fast path and the online-softmax fallback - forced (bit 4)
    and taken by itself when the scores leave the safe range (inputs x 8: log2-domain scores of several hundred, where 2^s
    overflows) - against the fp32 reference"""
    from opendwm_amd import ops
    D = heads * 64
    qkv = _rand((I * N, 3 * D), dev, 11, scale)
    cqkv = _rand((I * Lc, 3 * D), dev, 12, scale) if Lc else None
    rm = ops.rowmap_identity(I, N)
    kw, f, cf = {}, qkv.float(), (cqkv.float() if Lc else None)
    r0, r1 = _attn_ref(f[:, :D], f[:, D:2 * D], f[:, 2 * D:], rm.rows().to(dev), heads,
                       q1=cf[:, :D] if Lc else None, k1=cf[:, D:2 * D] if Lc else None, v1=cf[:, 2 * D:] if Lc else None)
    errs = {}
    for variant in (0, 12, 8, 3, (heads << 8) | 7, (heads << 8) | 12, 16 | (heads << 8) | 8, 16):
        out = torch.full((I * N, D), float("nan"), dtype=bf16, device=dev)
        cout = torch.full((I * Lc, D), float("nan"), dtype=bf16, device=dev) if Lc else None
        if Lc:
            kw = dict(q1=cqkv[:, :D], k1=cqkv[:, D:2 * D], v1=cqkv[:, 2 * D:], out1=cout)
        ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out, rm, heads, variant=variant, **kw)
        errs[variant] = max(rel_err(out, r0), rel_err(cout, r1) if Lc else 0.0)
    _log("attention_resident_forms", scale=scale, I=I, N=N, Lc=Lc, heads=heads, **{str(k): v for k, v in errs.items()})
    # huge scores: scale * log2(e) is folded into the bf16 Q fragments (as in every kernel of this file), 2^-9 relative on
    # log2-domain scores of several hundred moves a near-one-hot softmax by percents; the point of that case is the fallback
    assert all(e < (TOL_KERNEL if scale == 1.0 else 3e-2) for e in errs.values()), errs


@pytest.mark.parametrize("I,N,Lc,heads,hs", [(150, 256, 40, 4, 2), (3, 448, 154, 24, 6), (40, 448, 0, 12, 1), (70, 230, 0, 8, 2)])
def test_attention_resident_persistent_workgroups_across_item_seams(dev, I, N, Lc, heads, hs):
    """attn_res_kernel as a persistent kernel: more items than CUs (150 x 2 = 300 items, 40 x 12 = 480), so a workgroup walks
    several (problem, head group) items - table rebuilds, the copy pipeline across item seams, both wave geometries - against
    the reference on the first and the LAST problems (the ones a workgroup reaches after its first item), and repeated
    launches bit-identical.  Both segments of the inputs and of the output live in ONE allocation each, as the model passes them: the
    default kernel for these lengths (round 6: attn_stream_kernel) has one form for segments within +-16 GiB and one for pairs
    further apart (bit-identical: tests/test_round6_gpu.py; include/dwm_hip.h, dwm_attn_stream_launches).  Variants: the library's choice, the 12-wave resident kernel (bit 13), its 8-compute-wave geometry."""
    from opendwm_amd import ops
    D = heads * 64
    qc = _rand((I * (N + Lc), 3 * D), dev, 21)
    qkv, cqkv = qc[:I * N], (qc[I * N:] if Lc else None)
    rm = ops.rowmap_identity(I, N)
    outs = {}
    for variant in ((hs << 8), (hs << 8) | (1 << 13), (hs << 8) | 8):
        runs = []
        for _ in range(2):
            both = torch.full((I * (N + Lc), D), float("nan"), dtype=bf16, device=dev)
            out, cout = both[:I * N], (both[I * N:] if Lc else None)
            kw = dict(q1=cqkv[:, :D], k1=cqkv[:, D:2 * D], v1=cqkv[:, 2 * D:], out1=cout) if Lc else {}
            ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out, rm, heads, variant=variant, **kw)
            runs.append((out, cout))
        assert torch.equal(runs[0][0], runs[1][0]) and (not Lc or torch.equal(runs[0][1], runs[1][1]))
        outs[variant] = runs[0]
    errs = []
    for p0 in (0, I - 2):                       # two problems at the front, two at the back
        f = qkv[p0 * N:(p0 + 2) * N].float()
        cf = cqkv[p0 * Lc:(p0 + 2) * Lc].float() if Lc else None
        r0, r1 = _attn_ref(f[:, :D], f[:, D:2 * D], f[:, 2 * D:], ops.rowmap_identity(2, N).rows().to(dev), heads,
                           q1=cf[:, :D] if Lc else None, k1=cf[:, D:2 * D] if Lc else None, v1=cf[:, 2 * D:] if Lc else None)
        for variant, (out, cout) in outs.items():
            errs.append(max(rel_err(out[p0 * N:(p0 + 2) * N], r0), rel_err(cout[p0 * Lc:(p0 + 2) * Lc], r1) if Lc else 0.0))
    _log("attention_resident_item_seams", I=I, N=N, Lc=Lc, heads=heads, hs=hs, rel=max(errs))
    assert max(errs) < TOL_KERNEL


def test_attention_resident_temporal_rowmap_multihead(dev):
    """the resident kernel through a strided row map (row-wise temporal attention: L = frames x row width, token rows far apart),
    24 heads in groups of 6 per workgroup, bit-equal between the two wave geometries' fast paths is NOT required - both against
    the reference"""
    from opendwm_amd import ops
    B, T, V, h, w, heads = 1, 16, 2, 3, 28, 24
    D = heads * 64
    rm = ops.rowmap_temporal_rowwise(B, T, V, h, w)
    R = B * T * V * h * w
    qkv = _rand((R, 3 * D), dev, 13)
    f = qkv.float()
    ref, _ = _attn_ref(f[:, :D], f[:, D:2 * D], f[:, 2 * D:], rm.rows().to(dev), heads)
    errs = {}
    for variant in ((6 << 8) | 12, (6 << 8) | 8, (4 << 8) | 5):
        out = torch.full((R, D), float("nan"), dtype=bf16, device=dev)
        ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out, rm, heads, variant=variant)
        errs[variant] = rel_err(out, ref)
    _log("attention_resident_temporal_rowmap", L=rm.L0, **{str(k): v for k, v in errs.items()})
    assert all(e < TOL_KERNEL for e in errs.values()), errs


@pytest.mark.parametrize("V,w,heads", [(6, 28, 3), (6, 32, 2), (6, 16, 8), (4, 8, 2), (8, 11, 2), (3, 28, 2), (6, 28, 24)])
def test_attention_group_forms(dev, V, w, heads):
    """row-wise cross-view attention whose problems are V whole groups of w <= 32 tokens: the shared form (one workgroup per
    (problem, head group), K / V of a head copied to LDS once; V = 4 / 6 / 8), the per-wave form (variant bit 7) and the
    tiled kernel (bit 5) against the masked reference, with a mask that differs per batch entry and per view"""
    from opendwm_amd import ops
    B, T, h = 2, 2, 3
    D = heads * 64
    rm = ops.rowmap_crossview_rowwise(B, T, V, h, w)
    R = B * T * V * h * w
    qkv = _rand((R, 3 * D), dev, 3)
    gmask = O.ring_crossview_mask(B, V).to(dev)
    gmask[1, 1, V - 1] = True
    gmask[0, 0, 1] = False
    p = torch.arange(rm.n_problems, device=dev)[:, None, None]
    l = torch.arange(rm.L0, device=dev)
    ref_mask = gmask[p // rm.p_per_mask, ((l // rm.group_size) % V)[None, :, None], ((l // rm.group_size) % V)[None, None, :]]
    f = qkv.float()
    r0, _ = _attn_ref(f[:, :D], f[:, D:2 * D], f[:, 2 * D:], rm.rows().to(dev), heads, mask=ref_mask)
    errs = {}
    for name, variant in (("default", 0), ("per_wave", 128), ("tiled", 32)):
        out = torch.zeros((R, D), dtype=bf16, device=dev)
        ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out, rm, heads, group_mask=gmask, variant=variant)
        errs[name] = rel_err(out, r0)
    _log("attention_group_forms", V=V, w=w, heads=heads, **errs)
    assert all(e < TOL_KERNEL for e in errs.values()), errs


@pytest.mark.parametrize("variant", ATTN_VARIANTS)
@pytest.mark.parametrize("kind", ["crossview_rowwise", "crossview_full", "temporal_rowwise", "temporal_full",
                                  "temporal_pointwise"])
def test_attention_rowmaps_and_masks(dev, variant, kind):
    from opendwm_amd import ops
    B, T, V, h, w, heads = 2, 5, 6, 3, 7, 2
    D = heads * 64
    rm = getattr(ops, "rowmap_" + kind)(B, T, V, h, w)
    R = B * T * V * h * w
    qkv = _rand((R, 3 * D), dev, 3)
    out = torch.zeros((R, D), dtype=bf16, device=dev)
    gmask = None
    ref_mask = None
    if kind.startswith("crossview"):
        gmask = O.ring_crossview_mask(B, V).to(dev)
        gmask[1, 2, 5] = True                                  # make the two batch entries differ
        p = torch.arange(rm.n_problems, device=dev)[:, None, None]
        l = torch.arange(rm.L0, device=dev)
        ref_mask = gmask[p // rm.p_per_mask, ((l // rm.group_size) % V)[None, :, None], ((l // rm.group_size) % V)[None, None, :]]
    ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out, rm, heads, group_mask=gmask, variant=_variant_for(variant, heads))
    f = qkv.float()
    r0, _ = _attn_ref(f[:, :D], f[:, D:2 * D], f[:, 2 * D:], rm.rows().to(dev), heads, mask=ref_mask)
    e = rel_err(out, r0)
    _log("attention_rowmap", kind=kind, variant=variant, rel=e)
    assert e < TOL_KERNEL
    if ref_mask is not None:                                   # the same through the dense-mask mode
        out2 = torch.zeros_like(out)
        ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out2, rm, heads, dense_mask=ref_mask, variant=_variant_for(variant, heads))
        assert rel_err(out2, r0) < TOL_KERNEL


@pytest.mark.parametrize("L", [1, 3, 6, 8, 9, 16, 17, 31, 32])
@pytest.mark.parametrize("P,heads", [(13, 2), (7, 24), (50, 5)])
def test_attention_short_sequences_packed_kernel(dev, L, P, heads):
    """L <= 32, one segment, no mask -> `attn_small_kernel`: 32 / SL problems share one 32 x 32 MFMA tile (SL = 8, 16, 32
    token slots) behind a block-diagonal validity mask.  Problem counts that leave the last tile / last workgroup partly
    empty, every slot size and its ragged fill, three heads-per-wave splits; against fp32 SDPA and against the tiled
    kernel (variant bit 5) on the same inputs."""
    from opendwm_amd import ops
    D = heads * 64
    qkv = _rand((P * L, 3 * D), dev, 100 + L)
    rm = ops.rowmap_identity(P, L)
    out = torch.zeros((P * L, D), dtype=bf16, device=dev)
    ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out, rm, heads)
    tiled = torch.zeros_like(out)
    ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], tiled, rm, heads, variant=32)
    f = qkv.float()
    r0, _ = _attn_ref(f[:, :D], f[:, D:2 * D], f[:, 2 * D:], rm.rows().to(dev), heads)
    e, et = rel_err(out, r0), rel_err(tiled, r0)
    _log("attention_short", L=L, P=P, heads=heads, rel=e, rel_tiled=et)
    assert e < TOL_KERNEL and et < TOL_KERNEL


def test_attention_pointwise_temporal_packed_kernel_rowmap(dev):
    """the point-wise temporal row map (tokens of one (b, v, h, w) across frames, crossview_temporal_dit.py:352-361) at
    T = 16 through the packed kernel: rows of one problem are V*N apart, two problems per tile, q/k/v as column slices of
    one fused [rows, 3D] buffer."""
    from opendwm_amd import ops
    B, T, V, h, w, heads = 2, 16, 3, 3, 5, 24
    D = heads * 64
    rm = ops.rowmap_temporal_pointwise(B, T, V, h, w)
    R = B * T * V * h * w
    qkv = _rand((R, 3 * D), dev, 7)
    out = torch.zeros((R, D), dtype=bf16, device=dev)
    ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out, rm, heads)
    f = qkv.float()
    r0, _ = _attn_ref(f[:, :D], f[:, D:2 * D], f[:, 2 * D:], rm.rows().to(dev), heads)
    e = rel_err(out, r0)
    _log("attention_pointwise_packed", rel=e)
    assert e < TOL_KERNEL


def test_attention_online_softmax_spike(dev):
    """Force the running-max rescale: one key in a LATE tile dominates one query (cdna guide
    §5.4 rule 26), and large-magnitude scores."""
    from opendwm_amd import ops
    heads, L = 1, 300
    qkv = _rand((L, 192), dev, 9)
    qkv[7, :64] = 6.0
    qkv[250, 64:128] = 6.0           # q7 . k250 = 64*36 -> scaled 288: every earlier tile's max is tiny
    out = torch.zeros((L, 64), dtype=bf16, device=dev)
    rm = ops.rowmap_identity(1, L)
    for variant in ATTN_VARIANTS:
        ops.attention(qkv[:, :64], qkv[:, 64:128], qkv[:, 128:], out, rm, heads, variant=_variant_for(variant, heads))
        f = qkv.float()
        ref, _ = _attn_ref(f[:, :64], f[:, 64:128], f[:, 128:], rm.rows().to(dev), heads)
        e = rel_err(out, ref)
        _log("attention_spike", variant=variant, rel=e, row7=rel_err(out[7], ref[7]))
        assert e < TOL_KERNEL and rel_err(out[7], ref[7]) < TOL_KERNEL and torch.isfinite(out.float()).all()


# ------------------------------------------------------------------------ norms / glue
@pytest.mark.parametrize("rows,D", [(448 * 3, 1536), (100, 128), (77, 512)])
def test_layernorm_family(dev, rows, D):
    from opendwm_amd import ops
    x = _rand((rows, D), dev, 1, 2.0) + 0.5
    xf = x.float()
    n = F.layer_norm(xf, (D,), None, None, 1e-6)
    rpm = 16
    G = (rows + rpm - 1) // rpm
    mod = _rand((G, 4 * D), dev, 2, 0.5)
    ridx = torch.arange(rows, device=dev) // rpm
    sc, sh, sc2, sh2 = (mod[:, i * D:(i + 1) * D] for i in range(4))
    y2 = torch.empty_like(x)
    y = ops.layernorm(x, eps=1e-6, scale=sc, shift=sh, rows_per_mod=rpm, scale2=sc2, shift2=sh2, out2=y2)
    e1 = rel_err(y, n * (1 + sc.float()[ridx]) + sh.float()[ridx])
    e2 = rel_err(y2, n * (1 + sc2.float()[ridx]) + sh2.float()[ridx])
    w, b = _rand((D,), dev, 3) * 0.2 + 1, _rand((D,), dev, 4)
    e3 = rel_err(ops.layernorm(x, eps=1e-5, weight=w, bias=b), F.layer_norm(xf, (D,), w.float(), b.float(), 1e-5))
    add = _rand((G, D), dev, 5)
    xs = torch.empty_like(x)
    y4 = ops.layernorm(x, eps=1e-5, weight=w, bias=b, addvec=add, rows_per_add=rpm, xsum=xs)
    s = (xf + add.float()[ridx]).to(bf16)
    e4 = rel_err(xs, s.float())
    e5 = rel_err(y4, F.layer_norm(s.float(), (D,), w.float(), b.float(), 1e-5))
    _log("layernorm", rows=rows, D=D, e=[e1, e2, e3, e4, e5])
    assert max(e1, e2, e3, e5) < TOL_KERNEL and e4 < 1e-6


def test_elementwise_kernels(dev):
    from opendwm_amd import ops
    x = _rand((192, 1536), dev, 1, 3.0)
    assert rel_err(ops.silu(x), F.silu(x.float())) < TOL_KERNEL
    t = torch.tensor([0.0, 1.0, 17.0, 500.0, 999.0, 5.0, 3.25, -0.75], device=dev)
    for C in (256, 1536):
        got = ops.timestep_sinusoid(t, C)
        ref = O.timesteps_sinusoid(t, C)
        _log("sinusoid", C=C, maxabs=(got.float() - ref).abs().max().item())
        assert (got.float() - ref).abs().max() < 1e-2        # bf16 storage of values in [-1, 1]
    lat = torch.randn(6, 16, 8, 12, device=dev)
    cols = ops.patchify(lat, 2)
    ref = lat.reshape(6, 16, 4, 2, 6, 2).permute(0, 2, 4, 1, 3, 5).reshape(6 * 24, 64)
    assert torch.equal(cols, ref.to(bf16))
    assert torch.equal(ops.patchify(lat.to(bf16), 2), ref.to(bf16))
    y = _rand((6 * 24, 64), dev, 2)
    up = ops.unpatchify(y, 6, 16, 4, 6, 2)
    refu = torch.einsum("nhwpqc->nchpwq", y.view(6, 4, 6, 2, 2, 16)).reshape(6, 16, 8, 12)
    assert torch.equal(up, refu)
    # patchify o unpatchify round trip on the channel-permuted layout
    pred = _rand((2, 4096), dev, 3)
    latents = torch.randn(4096, device=dev)
    l0 = latents.clone()
    mi = torch.empty((2, 4096), dtype=bf16, device=dev)
    ops.cfg_euler_step(pred, latents, 4.0, -0.03, model_in=mi)
    u, c = pred.float()
    ref = l0 + (-0.03) * (u + 4.0 * (c - u))
    assert torch.allclose(latents, ref, atol=1e-6)
    assert torch.equal(mi[0], ref.to(bf16)) and torch.equal(mi[1], mi[0])
    assert torch.equal(ops.cast_bf16(l0), l0.to(bf16))


# ------------------------------------------------------------------ layout ImageAdapter
def test_layout_kernels(dev):
    from opendwm_amd import ops
    x = torch.rand(5, 6, 64, 96, device=dev)
    t = ops.unshuffle_tokens(x, 8)
    ref = F.pixel_unshuffle(x, 8).permute(0, 2, 3, 1).reshape(5 * 8 * 12, 384)
    assert torch.equal(t, ref.to(bf16))
    t2 = ops.unshuffle_tokens(x[:, :3].contiguous(), 8, 256)            # K padded 192 -> 256 with zeros
    assert torch.equal(t2[:, :192], F.pixel_unshuffle(x[:, :3], 8).permute(0, 2, 3, 1).reshape(-1, 192).to(bf16))
    assert torch.count_nonzero(t2[:, 192:]) == 0
    p = ops.avgpool2_tokens(t, 5, 8, 12)
    refp = F.avg_pool2d(t.float().view(5, 8, 12, 384).permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1).reshape(-1, 384)
    assert rel_err(p, refp) < TOL_KERNEL
    a, b = _rand((100, 256), dev, 1), _rand((100, 256), dev, 2)
    assert torch.equal(ops.add_(a.clone(), b), (a.float() + b.float()).to(bf16))


@pytest.mark.parametrize("I,h,w,C,N", [(3, 16, 28, 128, 128), (2, 4, 6, 64, 192), (7, 5, 3, 256, 64)])
def test_gemm_implicit_conv3x3(dev, I, h, w, C, N):
    """3x3 conv (pad 1) as implicit GEMM over the padded token grid == F.conv2d, plus the padded
    output path (c_grid) with an in-place residual."""
    from opendwm_amd import ops
    grid = ops.PaddedGrid(I, h, w)
    x = _rand((I, C, h, w), dev, 1)
    wt, b = _rand((N, C, 3, 3), dev, 2, (9 * C) ** -0.5), _rand((N,), dev, 3)
    ref = F.relu(F.conv2d(x.float(), wt.float(), b.float(), padding=1)).permute(0, 2, 3, 1).reshape(-1, N)
    idx = grid.interior_index().to(dev)
    xp = torch.zeros((grid.rows, C), dtype=bf16, device=dev)
    xp[idx] = x.permute(0, 2, 3, 1).reshape(-1, C)
    wp = wt.permute(0, 2, 3, 1).reshape(N, 9 * C).contiguous()
    out = ops.gemm(xp, wp, b, act=ops.ACT_RELU, a_grid=grid, conv3x3=True)
    e = rel_err(out, ref)
    # 1x1 conv from compact rows into the padded grid, accumulating onto a padded residual in place
    w1 = _rand((C, N), dev, 4, N ** -0.5)
    res = xp.clone()
    ops.gemm(out, w1, None, epilogue=ops.EPI_RESID, res=res, out=res, c_grid=grid)
    ref2 = x.float().permute(0, 2, 3, 1).reshape(-1, C) + out.float() @ w1.float().T
    e2 = rel_err(res[idx], ref2)
    border = torch.ones(grid.rows, dtype=torch.bool, device=dev)
    border[idx] = False
    _log("gemm_conv3x3", I=I, h=h, w=w, C=C, N=N, rel=e, rel_padded_out=e2)
    assert e < TOL_KERNEL and e2 < TOL_KERNEL and torch.count_nonzero(res[border]) == 0


def _adapter_cfg():
    return dict(in_channels=6, channels=[128, 128, 128], is_downblocks=[True, False, False], num_res_blocks=2,
                downscale_factor=8, use_zero_convs=True)


def test_image_adapter_vs_oracle(dev):
    cfg = small_config(condition_image_adapter_config=_adapter_cfg())
    sd = _bf16_round_sd(O.make_state_dict(cfg, 0))
    m = _hip_model(cfg, sd, dev)
    g = torch.Generator().manual_seed(5)
    img = torch.rand(2, 3, 3, 6, 64, 96, generator=g).to(bf16).float()
    ref = O.image_adapter(sd, cfg, img)
    got = m.condition_image_adapter(img.to(dev))
    for i, (a, b) in enumerate(zip(got, ref)):
        assert a.shape == b.shape
        e = rel_err(a, b)
        _log("image_adapter", level=i, rel=e)
        assert e < TOL_MODEL


def test_model_with_layout_adapter_vs_oracle(dev):
    cfg = small_config(condition_image_adapter_config=_adapter_cfg(), temporal_attention_type="pointwise")
    sd = _bf16_round_sd(O.make_state_dict(cfg, 0))
    m = _hip_model(cfg, sd, dev)
    inp = small_inputs(cfg, 0)
    g = torch.Generator().manual_seed(5)
    inp["condition_image_tensor"] = torch.rand(2, 3, 3, 6, 64, 96, generator=g)
    inp = {k: (v.to(bf16).float() if v.is_floating_point() and k not in ("timestep", "added_time_ids") else v)
           for k, v in inp.items()}
    ref = O.dit_forward(sd, cfg, **inp)
    di = to_dev(inp, dev)
    out, _, _ = m(di.pop("sample"), di.pop("timestep"), **di)
    e = rel_err(out[0], ref)
    _log("model_layout_adapter", rel=e)
    assert e < TOL_MODEL
    di = to_dev(inp, dev)                       # second call hits the adapter cache: identical result
    out2, _, _ = m(di.pop("sample"), di.pop("timestep"), **di)
    assert torch.equal(out2[0], out[0])


# --------------------------------------------------------------------------------- VAE
@pytest.mark.parametrize("I,h,w,C,G", [(3, 8, 12, 128, 32), (2, 16, 16, 64, 8), (5, 4, 4, 512, 32)])
def test_groupnorm_silu_and_upsample(dev, I, h, w, C, G):
    from opendwm_amd import ops
    P = h * w
    x = _rand((I * P, C), dev, 1, 2.0) + 0.3
    ga, be = _rand((C,), dev, 2) * 0.2 + 1, _rand((C,), dev, 3)
    xn = x.float().view(I, P, C).permute(0, 2, 1)
    ref = F.silu(F.group_norm(xn, G, ga.float(), be.float(), 1e-6)).permute(0, 2, 1).reshape(I * P, C)
    e1 = rel_err(ops.groupnorm_silu(x, I, P, ga, be, G, 1e-6), ref)
    grid = ops.PaddedGrid(I, h, w)
    yp = ops.groupnorm_silu(x, I, P, ga, be, G, 1e-6, out_grid=grid)
    idx = grid.interior_index().to(dev)
    e2 = rel_err(yp[idx], ref)
    border = torch.ones(grid.rows, dtype=torch.bool, device=dev)
    border[idx] = False
    ref3 = F.group_norm(xn, G, ga.float(), be.float(), 1e-6).permute(0, 2, 1).reshape(I * P, C)
    e3 = rel_err(ops.groupnorm_silu(x, I, P, ga, be, G, 1e-6, silu=False), ref3)
    up = ops.upsample2_padded(x, I, h, w)
    g2 = ops.PaddedGrid(I, 2 * h, 2 * w)
    refu = F.interpolate(x.float().view(I, h, w, C).permute(0, 3, 1, 2), scale_factor=2.0, mode="nearest")
    ok_up = torch.equal(up[g2.interior_index().to(dev)], refu.permute(0, 2, 3, 1).reshape(-1, C).to(bf16))
    _log("groupnorm", I=I, C=C, G=G, e=[e1, e2, e3], upsample_exact=ok_up)
    assert max(e1, e2, e3) < TOL_KERNEL and torch.count_nonzero(yp[border]) == 0 and ok_up


def test_softmax_rows(dev):
    from opendwm_amd import ops
    for L in (64, 448, 1792):
        x = _rand((300, L), dev, 1, 3.0)
        ref = torch.softmax(x.float() * 0.2, -1)
        e = rel_err(ops.softmax_rows(x, 0.2), ref)
        assert e < TOL_KERNEL, (L, e)


def test_vae_decode_vs_oracle(dev):
    """AutoencoderKL.decode (ctsd.py:1634-1640) at reduced width: every block type of the SD 3.5 VAE
    decoder (mid attention, channel-changing resnets with conv_shortcut, 3 nearest-2x upsamplers)."""
    from opendwm_amd.vae import AutoencoderKL
    vcfg = dict(block_out_channels=(64, 64, 128, 128), layers_per_block=2, norm_num_groups=16, latent_channels=16)
    sd = _bf16_round_sd(O.make_vae_state_dict(vcfg, 0))
    vae = AutoencoderKL(**vcfg)
    vae.load_state_dict(sd)
    vae = vae.to(dev).to(bf16).eval()
    g = torch.Generator().manual_seed(3)
    z = torch.randn(3, 16, 8, 8, generator=g).to(bf16).float()
    ref = O.vae_decode(sd, vcfg, z)
    out = vae.decode(z.to(dev), return_dict=False, chunk=2)[0]
    e = rel_err(out, ref)
    _log("vae_decode", rel=e, shape=list(out.shape))
    assert out.shape == (3, 3, 64, 64) and e < TOL_MODEL


def test_vae_full_width_decode_vs_oracle_on_device(dev):
    """the SD 3.5 VAE decoder at its real widths (128 / 256 / 512 / 512, 32 groups) on two 32x56 latents (256x448 px: the
    images of BASELINE config 3), against the fp32 oracle evaluated on the device"""
    from opendwm_amd.vae import AutoencoderKL
    vcfg = dict(block_out_channels=(128, 256, 512, 512), layers_per_block=2, norm_num_groups=32, latent_channels=16)
    sd = _bf16_round_sd(O.make_vae_state_dict(vcfg, 0))
    vae = AutoencoderKL(**vcfg)
    vae.load_state_dict(sd)
    vae = vae.to(dev).to(bf16).eval()
    z = torch.randn(2, 16, 32, 56, generator=torch.Generator().manual_seed(0)).to(bf16).float().to(dev)
    ref = O.vae_decode({k: v.to(dev) for k, v in sd.items()}, vcfg, z)
    out = vae.decode(z, return_dict=False)[0]
    e = rel_err(out, ref)
    _log("vae_decode_full_width", rel=e, shape=list(out.shape))
    assert out.shape == (2, 3, 256, 448) and e < TOL_MODEL


def test_vae_encode_vs_oracle(dev):
    from opendwm_amd.vae import AutoencoderKL
    vcfg = dict(block_out_channels=(64, 64, 128, 128), layers_per_block=2, norm_num_groups=16, latent_channels=16)
    sd = _bf16_round_sd(O.make_vae_state_dict(vcfg, 0))
    vae = AutoencoderKL(**vcfg)
    missing, unexpected = vae.load_state_dict(sd)
    assert not missing and not unexpected
    vae = vae.to(dev).to(bf16).eval()
    g = torch.Generator().manual_seed(4)
    x = (torch.rand(3, 3, 64, 64, generator=g) * 2 - 1).to(bf16).float()
    ref = O.vae_encode_moments(sd, vcfg, x)
    dist = vae.encode(x.to(dev), chunk=2).latent_dist
    e = rel_err(dist.parameters, ref)
    _log("vae_encode", rel=e, shape=list(dist.parameters.shape))
    assert dist.parameters.shape == (3, 32, 8, 8) and e < TOL_MODEL
    assert torch.equal(dist.mode(), dist.mean) and dist.sample().shape == (3, 16, 8, 8)
    # encode -> decode round trip runs end to end at the pipeline's scaling (ctsd.py:1216-1218,1636-1637)
    lat = (dist.mode() - vae.config.shift_factor) * vae.config.scaling_factor
    img = vae.decode(lat / vae.config.scaling_factor + vae.config.shift_factor)[0]
    assert img.shape == (3, 3, 64, 64) and torch.isfinite(img.float()).all()


# ----------------------------------------------------------------------- blocks / model
def _bf16_round_sd(sd):
    return {k: v.to(bf16).float() for k, v in sd.items()}


def _hip_model(cfg, sd, dev):
    from opendwm_amd.dit import DiTCrossviewTemporalConditionModel
    m = DiTCrossviewTemporalConditionModel(**cfg)
    m.load_state_dict(sd)
    return m.to(dev).to(bf16).eval()


def test_vt_block_vs_oracle(dev, small_cfg):
    from opendwm_amd import ops
    sd = _bf16_round_sd(O.make_state_dict(small_cfg, 0))
    m = _hip_model(small_cfg, sd, dev)
    blk = m.temporal_transformer_blocks[0]
    x = _rand((6, 40, 128), dev, 1)
    mask = torch.rand(6, 40, 40, device=dev) > 0.3
    mask[:, :, 0] = True
    for mk in (None, mask):
        ref = O.vt_self_attention_block(sd, "temporal_transformer_blocks.0", 2, x.float().cpu(),
                                        None if mk is None else mk.cpu())
        got = blk(x, mk)
        e = rel_err(got, ref)
        d = rel_err(got.float().cpu() - x.float().cpu(), ref - x.float().cpu())
        _log("vt_block", masked=mk is not None, rel=e, rel_delta=d)
        assert e < TOL_MODEL and d < 2 * TOL_MODEL


def test_joint_block_vs_oracle(dev, small_cfg):
    sd = _bf16_round_sd(O.make_state_dict(small_cfg, 0))
    m = _hip_model(small_cfg, sd, dev)
    I, N, Lc, D = 5, 24, 10, 128
    for i in (0, 2, 3):       # dual, plain, context-pre-only
        h, c, temb = _rand((I, N, D), dev, 1), _rand((I, Lc, D), dev, 2), _rand((I, D), dev, 3, 0.5)
        rc, rh = O.joint_transformer_block(sd, f"transformer_blocks.{i}", small_cfg, i,
                                           h.float().cpu(), c.float().cpu(), temb.float().cpu())
        from opendwm_amd import ops
        h2, c2 = h.reshape(I * N, D).clone(), c.reshape(I * Lc, D).clone()
        gc, gh = m.transformer_blocks[i].run(h2, c2, ops.silu(temb), I)
        eh = rel_err(gh.view(I, N, D).float().cpu() - h.float().cpu(), rh - h.float().cpu())
        ec = 0.0 if rc is None else rel_err(gc.view(I, Lc, D).float().cpu() - c.float().cpu(), rc - c.float().cpu())
        _log("joint_block", layer=i, rel_delta_h=eh, rel_delta_c=ec, rel_h=rel_err(gh.view(I, N, D), rh))
        assert eh < 2 * TOL_MODEL and ec < 2 * TOL_MODEL and rel_err(gh.view(I, N, D), rh) < TOL_MODEL
        assert (gc is None) == (rc is None)


@pytest.mark.parametrize("tt,gold", [("rowwise", "dit_small_forward.pt"), ("pointwise", "dit_small_forward_pointwise.pt"),
                                     ("full", "dit_small_forward_full.pt")])
def test_model_forward_vs_golden(dev, tt, gold):
    """HIP forward against the committed oracle fixture (fp32 weights) and against the oracle
    re-run on bf16-rounded weights (same inputs)."""
    cfg = small_config(temporal_attention_type=tt)
    sd32 = O.make_state_dict(small_config(), 0)
    sd = _bf16_round_sd(sd32)
    m = _hip_model(cfg, sd, dev)
    inp = small_inputs(cfg, 0)
    inp16 = {k: (v.to(bf16).float() if v.is_floating_point() and k != "timestep" and k != "added_time_ids" else v)
             for k, v in inp.items()}
    ref = O.dit_forward(sd, cfg, **inp16)
    di = to_dev(inp16, dev)
    out, a, b = m(di.pop("sample"), di.pop("timestep"), **di)
    assert isinstance(out, (list, tuple)) and out[0].shape == (2, 3, 3, 16, 8, 12) and out[0].dtype == bf16
    e = rel_err(out[0], ref)
    eg = rel_err(out[0], torch.load(os.path.join(GOLDEN, gold))["output"])
    _log("model_forward", temporal=tt, rel_vs_oracle=e, rel_vs_golden=eg)
    assert e < TOL_MODEL and eg < TOL_MODEL
    # determinism + batch independence (size-independent property): the CFG halves do not mix
    di = to_dev(inp16, dev)
    out2, _, _ = m(di.pop("sample"), di.pop("timestep"), **di)
    assert torch.equal(out2[0], out[0])
    half = {k: v[1:] for k, v in to_dev(inp16, dev).items()}
    out3, _, _ = m(half.pop("sample"), half.pop("timestep"), **half)
    assert torch.equal(out3[0], out[0][1:])


def test_model_5d_inputs_and_flags(dev, small_cfg):
    """disable_crossview / disable_temporal switch the mixers to alpha = 1 per batch entry
    (crossview_temporal.py:44-47); result must match the oracle."""
    sd = _bf16_round_sd(O.make_state_dict(small_cfg, 0))
    m = _hip_model(small_cfg, sd, dev)
    inp = small_inputs(small_cfg, 0)
    inp = {k: (v.to(bf16).float() if v.is_floating_point() and k not in ("timestep", "added_time_ids") else v)
           for k, v in inp.items()}
    inp["disable_crossview"] = torch.tensor([True, False])
    inp["disable_temporal"] = torch.tensor([False, True])
    ref = O.dit_forward(sd, small_cfg, **inp)
    di = to_dev(inp, dev)
    out, _, _ = m(di.pop("sample"), di.pop("timestep"), **di)
    e = rel_err(out[0], ref)
    _log("model_flags", rel=e)
    assert e < TOL_MODEL


def test_denoise_two_steps_vs_golden(dev, small_cfg):
    from opendwm_amd.pipeline import CTSDDenoiser
    sd = _bf16_round_sd(O.make_state_dict(small_cfg, 0))
    m = _hip_model(small_cfg, sd, dev)
    gold = torch.load(os.path.join(GOLDEN, "denoise_small_2steps.pt"))
    inp = small_inputs(small_cfg, 0)
    cond = to_dev({k: v for k, v in inp.items() if k not in ("sample", "timestep")}, dev)
    den = CTSDDenoiser(m, guidance_scale=4.0, inference_steps=4)
    out = den.run(gold["latents_in"].to(dev), cond, stop=2)
    e = rel_err(out, gold["latents_out"])
    _log("denoise_2steps", rel=e)
    assert out.dtype == torch.float32 and e < TOL_MODEL


@pytest.mark.parametrize("mode", ["reference_frames", "diffusion_forcing"])
def test_denoise_modes_vs_oracle(dev, small_cfg, mode):
    """Reference-frame injection and diffusion forcing of inference_pipeline (ctsd.py:1498-1572)."""
    from opendwm_amd.pipeline import CTSDDenoiser
    sd = _bf16_round_sd(O.make_state_dict(small_cfg, 0))
    m = _hip_model(small_cfg, sd, dev)
    inp = small_inputs(small_cfg, 0)
    cond = {k: v for k, v in inp.items() if k not in ("sample", "timestep")}
    g = torch.Generator().manual_seed(11)
    lat = torch.randn(1, 3, 3, 16, 8, 12, generator=g)
    img = torch.randn(1, 3, 3, 16, 8, 12, generator=g)
    if mode == "reference_frames":
        kw = dict(image_latents=img, reference_frame_count=1)
        steps, stop = 4, 2
    else:
        kw = dict(image_latents=img, diffusion_forcing=True, take_time=0)
        steps, stop = 6, 3
    ref = O.denoise(sd, small_cfg, lat, cond, steps=steps, guidance_scale=4.0, stop=stop, **kw)
    den = CTSDDenoiser(m, guidance_scale=4.0, inference_steps=steps)
    kwd = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in kw.items()}
    out = den.run(lat.to(dev), to_dev(cond, dev), stop=stop, **kwd)
    e = rel_err(out, ref)
    _log("denoise_mode", mode=mode, rel=e)
    assert e < TOL_MODEL
    if mode == "reference_frames":
        assert torch.equal(out[:, :1].cpu(), img[:, :1])


@pytest.mark.parametrize("mode", ["full", "reference_frames", "diffusion_forcing"])
def test_denoise_step_graph_replay_equals_eager(dev, small_cfg, mode):
    """Whole-step HIP graph (CTSDDenoiser.enable_graph): captured once, replayed per step with refreshed timestep /
    sigma-step buffers - bit-identical latents to launching the same kernels eagerly."""
    from opendwm_amd.pipeline import CTSDDenoiser
    sd = _bf16_round_sd(O.make_state_dict(small_cfg, 0))
    m = _hip_model(small_cfg, sd, dev)
    inp = small_inputs(small_cfg, 0)
    cond = to_dev({k: v for k, v in inp.items() if k not in ("sample", "timestep")}, dev)
    g = torch.Generator().manual_seed(12)
    lat = torch.randn(1, 3, 3, 16, 8, 12, generator=g).to(dev)
    img = torch.randn(1, 3, 3, 16, 8, 12, generator=g).to(dev)
    kw = {"full": {}, "reference_frames": dict(image_latents=img, reference_frame_count=1),
          "diffusion_forcing": dict(image_latents=img, diffusion_forcing=True, take_time=0)}[mode]
    steps = 6
    eager = CTSDDenoiser(m, guidance_scale=4.0, inference_steps=steps).run(lat, cond, stop=4, **kw)
    den = CTSDDenoiser(m, guidance_scale=4.0, inference_steps=steps).enable_graph()
    graphed = den.run(lat, cond, stop=4, **kw)
    assert den._graph is not None
    _log("denoise_graph", mode=mode, equal=bool(torch.equal(eager, graphed)), rel=rel_err(graphed, eager))
    assert torch.equal(eager, graphed)
    again = den.run(lat, cond, stop=4, **kw)                     # a second prepare() re-captures on fresh buffers
    assert torch.equal(again, eager)


def _cfg_split_worker(rank, world, port, cfg, sd, lat, cond, path):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from opendwm_amd.pipeline import CTSDDenoiser
    dev = torch.device("cuda:0")
    m = _hip_model(cfg, sd, dev)
    den = CTSDDenoiser(m, guidance_scale=4.0, inference_steps=4, cfg_group=dist.group.WORLD)
    out = den.run(lat.to(dev), to_dev(cond, dev), stop=3)
    torch.save(out.cpu(), f"{path}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_cfg_split_two_ranks(dev, small_cfg):
    """Classifier-free-guidance split: two ranks run the unconditional / conditional half of one sample
    and exchange the prediction halves with one all-gather per step (gloo here, both ranks on the one GPU; RCCL in
    production).  Both ranks hold bit-identical latents, equal to the single-process run up to bf16 round-off (the
    half-size batch picks other GEMM tile grids)."""
    import tempfile
    import torch.multiprocessing as mp
    from opendwm_amd.pipeline import CTSDDenoiser
    sd = _bf16_round_sd(O.make_state_dict(small_cfg, 0))
    inp = small_inputs(small_cfg, 0)
    cond = {k: v for k, v in inp.items() if k not in ("sample", "timestep")}
    lat = torch.randn(1, 3, 3, 16, 8, 12, generator=torch.Generator().manual_seed(13))
    single = CTSDDenoiser(_hip_model(small_cfg, sd, dev), guidance_scale=4.0, inference_steps=4).run(lat.to(dev), to_dev(cond, dev), stop=3).cpu()
    ctx = mp.get_context("spawn")
    port = 29500 + (os.getpid() + 7) % 2000
    path = os.path.join(tempfile.mkdtemp(), "cfg_split")
    procs = [ctx.Process(target=_cfg_split_worker, args=(r, 2, port, small_cfg, sd, lat, cond, path)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    a, b = torch.load(path + ".0"), torch.load(path + ".1")
    e = rel_err(a, single)
    _log("cfg_split", ranks_equal=bool(torch.equal(a, b)), rel_vs_single=e)
    assert torch.equal(a, b) and e < 5e-3


def _frame_shard_worker(rank, world, port, cfg, sd, lat, cond, kw, path):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from opendwm_amd.pipeline import CTSDDenoiser
    dev = torch.device("cuda:0")
    m = _hip_model(cfg, sd, dev)
    den = CTSDDenoiser(m, guidance_scale=4.0, inference_steps=4, frame_group=dist.group.WORLD)
    out = den.run(lat.to(dev), to_dev(cond, dev), stop=3, **{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in kw.items()})
    torch.save(out.cpu(), f"{path}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("temporal,mode", [("rowwise", "full"), ("pointwise", "diffusion_forcing"), ("rowwise", "reference_frames")])
def test_frame_shard_two_ranks(dev, small_cfg, temporal, mode):
    """Intra-sample sharding (opendwm_amd.sharding): the 4 frames of one sample on two ranks, one
    all-to-all before and after every temporal block (gloo here, both ranks on the one GPU; RCCL in production), per-frame
    timesteps / reference frames / scheduler update on the rank that owns the frame.  Every rank returns the whole
    sample, equal to the single-process run up to bf16 round-off (half-size GEMM grids; 5e-3 rel)."""
    import tempfile
    import torch.multiprocessing as mp
    from opendwm_amd.pipeline import CTSDDenoiser
    cfg = dict(small_cfg, temporal_attention_type=temporal)
    sd = _bf16_round_sd(O.make_state_dict(cfg, 0))
    inp = small_inputs(cfg, 0, T=4)
    cond = {k: v for k, v in inp.items() if k not in ("sample", "timestep")}
    lat = torch.randn(1, 4, 3, 16, 8, 12, generator=torch.Generator().manual_seed(13))
    img = torch.randn(1, 4, 3, 16, 8, 12, generator=torch.Generator().manual_seed(14))
    kw = {"full": {}, "reference_frames": dict(image_latents=img, reference_frame_count=1),
          "diffusion_forcing": dict(image_latents=img, diffusion_forcing=True, take_time=0)}[mode]
    single = CTSDDenoiser(_hip_model(cfg, sd, dev), guidance_scale=4.0, inference_steps=4).run(
        lat.to(dev), to_dev(cond, dev), stop=3, **{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in kw.items()}).cpu()
    ctx = mp.get_context("spawn")
    port = 29500 + (os.getpid() + 11) % 2000
    path = os.path.join(tempfile.mkdtemp(), "frame_shard")
    procs = [ctx.Process(target=_frame_shard_worker, args=(r, 2, port, cfg, sd, lat, cond, kw, path)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    a, b = torch.load(path + ".0"), torch.load(path + ".1")
    e = rel_err(a, single)
    _log("frame_shard", temporal=temporal, mode=mode, ranks_equal=bool(torch.equal(a, b)), rel_vs_single=e)
    assert a.shape == single.shape and torch.equal(a, b) and e < 5e-3


def _random_cameras(B, T, V, seed):
    """normalised pinhole intrinsics + camera -> reference-ego rigid transforms, [B, T, V, 3, 3] / [B, T, V, 4, 4]"""
    g = torch.Generator().manual_seed(seed)
    K = torch.zeros(B, T, V, 3, 3)
    K[..., 0, 0] = 0.8 + 0.4 * torch.rand(B, T, V, generator=g)
    K[..., 1, 1] = 1.2 + 0.4 * torch.rand(B, T, V, generator=g)
    K[..., 0, 2] = 0.45 + 0.1 * torch.rand(B, T, V, generator=g)
    K[..., 1, 2] = 0.45 + 0.1 * torch.rand(B, T, V, generator=g)
    K[..., 2, 2] = 1.0
    M = torch.zeros(B, T, V, 4, 4)
    q, _ = torch.linalg.qr(torch.randn(B, T, V, 3, 3, generator=g))
    M[..., :3, :3] = q
    M[..., :3, 3] = 0.5 * torch.randn(B, T, V, 3, generator=g)
    M[..., 3, 3] = 1.0
    return K, M


@pytest.mark.parametrize("temporal", ["rowwise", "pointwise"])
def test_frame_shard_two_ranks_explicit_perspective(dev, temporal):
    """Frame sharding with perspective_modeling_type="explicit" (examples/ctsd_unimlvg_6views_video_generation.json;
    crossview_temporal_dit.py:440-458): the temporal blocks of a rank run on all frames of its token rows, so their per-token ray
    embedding is built from the gathered camera matrices of every frame, restricted to those rows.  Equal to the single-process
    run up to bf16 round-off, both ranks bit-identical."""
    import tempfile
    import torch.multiprocessing as mp
    from opendwm_amd.pipeline import CTSDDenoiser
    cfg = small_config(perspective_modeling_type="explicit", temporal_attention_type=temporal)
    sd = _bf16_round_sd(O.make_state_dict(cfg, 0))
    inp = small_inputs(cfg, 0, T=4)
    inp.pop("added_time_ids")
    cond = {k: v for k, v in inp.items() if k not in ("sample", "timestep")}
    K, M = _random_cameras(1, 4, 3, 21)
    cond["camera_intrinsics_norm"] = torch.cat([K, K])                     # CFG-doubled, as every other condition
    cond["camera2referego"] = torch.cat([M, M])
    lat = torch.randn(1, 4, 3, 16, 8, 12, generator=torch.Generator().manual_seed(13))
    single = CTSDDenoiser(_hip_model(cfg, sd, dev), guidance_scale=4.0, inference_steps=4).run(lat.to(dev), to_dev(cond, dev), stop=3).cpu()
    ctx = mp.get_context("spawn")
    port = 29500 + (os.getpid() + 17) % 2000
    path = os.path.join(tempfile.mkdtemp(), "frame_shard_explicit")
    procs = [ctx.Process(target=_frame_shard_worker, args=(r, 2, port, cfg, sd, lat, cond, {}, path)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    a, b = torch.load(path + ".0"), torch.load(path + ".1")
    e = rel_err(a, single)
    _log("frame_shard_explicit", temporal=temporal, ranks_equal=bool(torch.equal(a, b)), rel_vs_single=e)
    assert a.shape == single.shape and torch.equal(a, b) and e < 5e-3


@pytest.mark.parametrize("name", ["crossview_rowwise_0", "crossview_rowwise_1", "crossview_full_0", "temporal_full", "temporal_rowwise", "temporal_pointwise"])
def test_attention_rowmaps_and_mixer_vs_reference_fixture(dev, name):
    """Golden vectors produced by the REFERENCE's own forward_crossview / forward_temporal_block_and_mix_result code
    (tests/golden/make_reference_fixtures.py): einops rearranges, [B,V,V] -> token mask expansion and AlphaBlender, with a
    plain SDPA block (2 heads x 64, identity projections).  Here: the HIP attention kernel with the row maps / group mask
    (no rearranged copy is ever made) followed by the AlphaBlender GEMM epilogue."""
    from opendwm_amd import ops
    from opendwm_amd.blocks import AlphaBlender
    fx = torch.load(os.path.join(GOLDEN, "reference_blocks.pt"))
    s, c = fx["shape"], fx["blocks"][name]
    B, T, V, h, w, C = s["B"], s["T"], s["V"], s["h"], s["w"], s["C"]
    kind = "_".join(name.split("_")[:2])
    rm = {"crossview_rowwise": ops.rowmap_crossview_rowwise, "crossview_full": ops.rowmap_crossview_full, "temporal_full": ops.rowmap_temporal_full,
          "temporal_rowwise": ops.rowmap_temporal_rowwise, "temporal_pointwise": ops.rowmap_temporal_pointwise}[kind](B, T, V, h, w)
    hidden = fx["hidden"].to(bf16)
    x = (hidden.float() + (fx["view_emb"] if name.startswith("crossview") else fx["seq_emb"])).to(bf16)
    tok = x.reshape(-1, C).to(dev)
    out = torch.empty_like(tok)
    gm = fx["mask"].to(dev) if kind == "crossview_rowwise" else None
    ops.attention(tok, tok, tok, out, rm, 2, group_mask=gm)
    mixer = AlphaBlender(2.0, merge_strategy="learned_with_images").to(dev)
    alpha = mixer.get_alpha(c["disable"].to(dev), B)
    eye = torch.eye(C, dtype=bf16, device=dev)
    mixed = ops.gemm(out, eye, None, epilogue=ops.EPI_RESID, blend=hidden.reshape(-1, C).to(dev), alpha=alpha,
                     rows_per_alpha=T * V * h * w)
    # reference run in fp32 on fp32 inputs; bf16 inputs here: compare against the fixture recomputed from the rounded inputs
    a = torch.where(c["disable"], torch.ones(1), torch.sigmoid(fx["mix_factor"]))
    xin = x.float().reshape(-1, C)[rm.rows()].view(rm.n_problems, rm.L0, C)
    q = xin.view(rm.n_problems, rm.L0, 2, 64).transpose(1, 2)
    ref_blk = F.scaled_dot_product_attention(q, q, q, attn_mask=None if c["block_mask"] is None else c["block_mask"][:, None])
    ref_blk = ref_blk.transpose(1, 2).reshape(rm.n_problems, rm.L0, C)
    back = torch.empty(B * T * V * h * w, C)
    back[rm.rows().reshape(-1)] = ref_blk.reshape(-1, C)
    want = a.view(B, 1, 1) * hidden.float().view(B, -1, C) + (1 - a.view(B, 1, 1)) * back.view(B, -1, C)
    e = rel_err(mixed, want.reshape(-1, C))
    drift = rel_err(want.reshape(-1, C), c["out"].reshape(-1, C))      # only the bf16 rounding of the inputs
    _log("reference_fixture_block", case=name, rel=e, input_rounding=drift)
    assert e < TOL_KERNEL * 2 and drift < 1e-2


def test_model_forward_vs_reference_forward_fixture(dev, small_cfg):
    """tests/golden/reference_forward.pt: the REAL DiTCrossviewTemporalConditionModel.forward / VTSelfAttentionBlock.forward
    run with oracle leaf modules (make_reference_forward_fixture.py).  6-D inputs with a disable_temporal flag, and the
    5-D form [B, T, C, H, W]: the reference returns it with the inserted view axis still in place (tuple form)."""
    fxf = torch.load(os.path.join(GOLDEN, "reference_forward.pt"))
    sd = _bf16_round_sd(O.make_state_dict(small_cfg, 0))
    m = _hip_model(small_cfg, sd, dev)
    inp = small_inputs(small_cfg, 0)
    inp["disable_temporal"] = fxf["rowwise"]["disable_temporal"]
    di = to_dev(inp, dev)
    out, _, _ = m(di.pop("sample"), di.pop("timestep"), **di)
    e6 = rel_err(out[0], fxf["rowwise"]["output"])
    inp1 = small_inputs(small_cfg, 0, V=1)
    five = {k: (v.squeeze(2) if torch.is_tensor(v) and v.dim() >= 3 and k != "crossview_attention_mask" else v) for k, v in inp1.items()}
    five["disable_temporal"] = torch.zeros(2, 1, dtype=torch.bool)
    d5 = to_dev(five, dev)
    out5, _, _ = m(d5.pop("sample"), d5.pop("timestep"), **d5)
    e5 = rel_err(out5[0], fxf["five_dim"]["output"])
    d5 = to_dev(five, dev)
    as_dict = m(d5.pop("sample"), d5.pop("timestep"), return_dict=True, **d5)
    _log("reference_forward_fixture", rel_6d=e6, rel_5d=e5, shape_5d=list(out5[0].shape))
    # fixture weights are fp32, the model holds them rounded to bf16: the bound is the usual bf16 one
    assert e6 < TOL_MODEL and e5 < TOL_MODEL
    assert out5[0].dim() == 6 and out5[0].shape[2] == 1 and as_dict["noise_pred"].dim() == 5


def test_full_width_block_stack_vs_oracle_on_device(dev):
    """BASELINE config-3 token geometry (6 views x 16 frames x 32x56 latents, CFG batch 2,
    d = 1536, 24 heads, 154 text tokens) with the first 6 layers of the schedule (dual blocks,
    cross-view after 1 and 5, temporal after 2 and 3): HIP bf16 vs the oracle run in fp32 on
    the same device."""
    cfg = O.make_config(num_layers=6, dual_attention_layers=[0, 1, 2, 3, 4, 5], crossview_block_layers=[1, 5],
                        temporal_block_layers=[2, 3])
    gen = torch.Generator().manual_seed(0)
    sd = {}
    for name, shape in O.param_shapes(cfg).items():
        sd[name] = O.synth_param(name, shape, cfg, gen).to(bf16)
    m = _hip_model(cfg, sd, dev)
    sd_dev = {k: v.to(dev).float() for k, v in sd.items()}
    inp = O.make_inputs(cfg, 2, 16, 6, 32, 56, seed=0)
    inp = {k: (v.to(bf16).float() if v.is_floating_point() and k not in ("timestep", "added_time_ids") else v)
           for k, v in inp.items()}
    di = to_dev(inp, dev)
    ref = O.dit_forward(sd_dev, cfg, **di)
    out, _, _ = m(di.pop("sample"), di.pop("timestep"), **di)
    e = rel_err(out[0], ref)
    _log("full_width_6layers", rel=e)
    assert torch.isfinite(out[0].float()).all() and e < TOL_MODEL


def test_vae_sd21_quant_convs(dev):
    """SD 2.1 VAE: 4 latent channels, quant_conv on the moments, post_quant_conv before the decoder (diffusers AutoencoderKL
    with use_quant_conv / use_post_quant_conv, scaling 0.18215)"""
    from opendwm_amd.vae import AutoencoderKL
    vcfg = dict(block_out_channels=(64, 64, 128, 128), layers_per_block=2, norm_num_groups=16, latent_channels=4,
                use_quant_conv=True, use_post_quant_conv=True, scaling_factor=0.18215, shift_factor=None)
    sd = _bf16_round_sd(O.make_vae_state_dict(vcfg, 0))
    vae = AutoencoderKL(**vcfg)
    missing, unexpected = vae.load_state_dict(sd)
    assert not missing and not unexpected
    vae = vae.to(dev).to(bf16).eval()
    g = torch.Generator().manual_seed(4)
    x = (torch.rand(2, 3, 64, 64, generator=g) * 2 - 1).to(bf16).float()
    ref = O.vae_encode_moments(sd, vcfg, x)
    dist = vae.encode(x.to(dev)).latent_dist
    e1 = rel_err(dist.parameters, ref)
    z = torch.randn(2, 4, 8, 8, generator=g).to(bf16).float()
    e2 = rel_err(vae.decode(z.to(dev))[0], O.vae_decode(sd, vcfg, z))
    _log("vae_sd21", encode=e1, decode=e2)
    assert dist.parameters.shape == (2, 8, 8, 8) and e1 < TOL_MODEL and e2 < TOL_MODEL


# ---------------------------------------------------------------- tensor-timestep schedulers (temporal_independent.py)
def test_schedulers_vs_reference_fixture(dev):
    """opendwm_amd.schedulers (HIP kernels dwm_frame_affine / dwm_cfg_ddim_step) against the vectors of the EXECUTED
    reference DDPMScheduler.add_noise / get_velocity and DDIMScheduler.step (tests/golden/reference_schedulers.pt), all
    prediction types, eta > 0, clipping, per-sample / per-frame / per-view timesteps; then the fused CFG form against the
    same step applied to the guided prediction."""
    from opendwm_amd.schedulers import DDIMScheduler, DDPMScheduler
    fx = torch.load(os.path.join(GOLDEN, "reference_schedulers.pt"))
    d = fx["ddpm"]
    sch = DDPMScheduler()
    worst = 0.0
    for name, c in d["cases"].items():
        noisy = sch.add_noise(d["x0"].to(dev), d["noise"].to(dev), c["timesteps"].to(dev))
        vel = sch.get_velocity(d["x0"].to(dev), d["noise"].to(dev), c["timesteps"])
        worst = max(worst, (noisy.cpu() - c["noisy"]).abs().max().item(), (vel.cpu() - c["velocity"]).abs().max().item())
    assert worst < 2e-6, worst
    dd = fx["ddim"]
    errs = {}
    for name, c in dd["cases"].items():
        kw = c["kw"]
        s = DDIMScheduler(prediction_type=kw["prediction_type"], clip_sample=kw.get("clip_sample", False),
                          clip_sample_range=kw.get("clip_sample_range", 1.0), set_alpha_to_one=kw.get("set_alpha_to_one", False))
        s.set_timesteps(c["num_inference_steps"])
        sample = dd["sample"].to(dev)
        keep = sample.clone()
        prev, x0 = s.step(dd["model_output"].to(dev), c["timesteps"].to(dev), sample, eta=kw.get("eta", 0.0),
                          use_clipped_model_output=kw.get("use_clipped", False),
                          variance_noise=dd["variance_noise"].to(dev) if kw.get("eta", 0.0) > 0 else None, return_dict=False)
        assert torch.equal(sample, keep)                                   # the caller's sample is not modified
        errs[name] = max((prev.cpu() - c["prev_sample"]).abs().max().item(), (x0.cpu() - c["pred_original_sample"]).abs().max().item())
    _log("schedulers_vs_reference", ddpm_max_abs=worst, ddim_max_abs=errs)
    assert all(v < 3e-5 for v in errs.values()), errs
    # fused classifier-free guidance (ctsd.py:1548-1552) + step + next bf16 model input
    c = dd["cases"]["v_eta0"]
    s = DDIMScheduler()
    s.set_timesteps(50)
    u, cnd = dd["model_output"].to(dev).to(bf16), dd["variance_noise"].to(dev).to(bf16)
    guided = u.float() + 3.0 * (cnd.float() - u.float())
    want, _ = s.step(guided, c["timesteps"].to(dev), dd["sample"].to(dev), return_dict=False)
    model_in = torch.empty(2, *dd["sample"].shape, dtype=bf16, device=dev)
    got, _ = s.step(torch.stack([u, cnd]).contiguous(), c["timesteps"].to(dev), dd["sample"].to(dev), return_dict=False, guidance_scale=3.0,
                    model_in=model_in)
    assert (got - want).abs().max().item() < 1e-5
    assert torch.equal(model_in[0], got.to(bf16)) and torch.equal(model_in[1], got.to(bf16))


# ---------------------------------------------------------------- explicit perspective modelling (RayEncoder / get_rays)
def test_explicit_perspective_forward_vs_oracle_and_reference_fixture(dev):
    """perspective_modeling_type="explicit" (examples/ctsd_unimlvg_6views_video_generation.json, configs/ctsd/unimlvg/*;
    crossview_temporal_dit.py:11-102, 156-159, 440-458): the ray-feature kernel against the oracle's positional encodings of
    the executed reference rays, and the whole forward against the oracle (bf16-rounded weights) and against the vector of
    the executed reference forward (fp32 weights)."""
    fx = torch.load(os.path.join(GOLDEN, "reference_forward.pt"))["explicit"]
    cfg = small_config(perspective_modeling_type="explicit")
    sd32 = O.make_state_dict(cfg, 0)
    sd = _bf16_round_sd(sd32)
    m = _hip_model(cfg, sd, dev)
    inp = small_inputs(cfg, 0)
    inp.pop("added_time_ids")
    cams = {k: fx[k] for k in ("camera_intrinsics_norm", "camera2referego")}
    hh, ww = inp["sample"].shape[-2] // 2, inp["sample"].shape[-1] // 2
    feat = m.rayencoder.features(cams["camera_intrinsics_norm"].to(dev), cams["camera2referego"].to(dev), hh, ww)
    I = fx["rays_o"].shape[0]
    want = torch.cat([O.positional_encoding(fx["rays_o"].unsqueeze(1), 8).view(I, 1, 1, -1).repeat(1, hh, ww, 1),
                      O.positional_encoding(fx["rays_d"].flatten(1, 2), 4).view(I, hh, ww, -1)], -1).view(I * hh * ww, 72)
    e_feat = (feat[:, :72].float().cpu() - want).abs().max().item()
    assert feat.shape == (I * hh * ww, 128) and torch.count_nonzero(feat[:, 72:]) == 0
    ref = O.dit_forward(sd, cfg, **inp, **cams)
    di = to_dev({**inp, **cams}, dev)
    out, _, _ = m(di.pop("sample"), di.pop("timestep"), **di)
    e, efx = rel_err(out[0], ref), rel_err(out[0], fx["output"])
    _log("explicit_perspective", feature_max_abs=e_feat, rel_vs_oracle=e, rel_vs_reference_forward=efx)
    assert e_feat < 5e-3           # bf16 features of sin / cos in [-1, 1]: 2^-9 rounding + fp32 argument reduction at 128 pi |o|
    assert e < TOL_MODEL and efx < TOL_MODEL
    with pytest.raises(RuntimeError):
        di = to_dev(inp, dev)
        m(di.pop("sample"), di.pop("timestep"), **di)                     # camera matrices are required in this mode
