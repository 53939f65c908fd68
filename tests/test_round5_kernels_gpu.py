"""GPU leg: kernels and host restructurings that were written at the end of round 4 without a GPU at hand and validated + measured
by the first call of round 5 (profiles/r5a_*), now defaults:

  * the GENERAL form of the 4-wave GEMM kernels (gemm_bf16_4w.hip, template parameter GEN): ragged M / N, the A row map and the taps
    of an implicit convolution (dense and stride 2), the per-image residual row - a battery in a subprocess with DWM_GEMM4W=1 (every
    covered launch; the variable is read once per process), against fp64 products / F.conv2d AND against the 8-wave kernels' output of
    the same calls (DWM_GEMM4W=0); `dwm_gemm4w_launches_general` must count exactly the covered calls.
  * `DiTCrossviewTemporalConditionModel.stack_modulation` (dit.py): the AdaLN modulation rows of all joint blocks and norm_out from
    ONE stacked GEMM per forward (host-side restructuring over validated kernels) - small model against the oracle and against
    the per-block forward, three temporal types, and the stack rebuilt after a state-dict load.
  * attn_stream_kernel (attention_stream.hip, round 6; variant bit 12): the resident attention kernel with ONE wave per SIMD and up to
    five query tiles per wave, V double-buffered in LDS, K fragments from global memory, Q in AGPRs (round 5's attn_res4_kernel with its
    head seam removed) - every remainder class of its tile schedule, both softmax paths, item seams (one and several heads per item), a
    strided row map, Q with the softmax scale folded in by the producer (variant bit 15) - against the fp32 reference and against
    attn_res_kernel.
(attn_res2_kernel, the 8-wave / two-tiles-per-wave form of round 4, measured 20 % SLOWER than attn_res_kernel - 611 against 769
TFLOP/s at L = 602, profiles/r5a_microbench_attn_res2.log - and was deleted.)
"""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tests.common import rel_err                      # noqa: E402
from tests.test_hip_gpu import TOL_KERNEL, _attn_ref, _log, _rand      # noqa: E402

TOL, TOL32 = 6e-3, 2e-5


def _gemm_general_battery():
    """runs in a subprocess (DWM_GEMM4W=1, or 0 for the 8-wave reference outputs); prints one JSON line + saves the outputs"""
    import torch.nn.functional as F
    from opendwm_amd import _lib, ops
    from opendwm_amd.blocks import geglu_pack
    f32 = torch.float32
    dev = torch.device("cuda:0")
    lib = _lib.load()
    out, keep = {}, {}

    def rnd(shape, seed, scale=1.0, dtype=bf16):
        g = torch.Generator().manual_seed(seed)
        return (torch.randn(*shape, generator=g) * scale).to(dev).to(dtype)

    def rel(a, b):
        a, b = a.double(), b.double()
        return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()

    gen = lambda: int(lib.dwm_gemm4w_launches_general())
    n0 = gen()
    # PLAIN, ragged M and / or N (N % 8 == 0), several K lengths incl. two steps
    for M, N, K in [(300, 256, 128), (256, 320, 128), (300, 328, 192), (1000, 512, 256), (29568 // 8, 1536, 1536), (257, 8, 128), (1, 264, 640)]:
        a, w, b = rnd((M, K), 1), rnd((N, K), 2, K ** -0.5), rnd((N,), 3)
        y = a.double() @ w.double().T + b.double()
        got = ops.gemm(a, w, b, act=ops.ACT_SILU, split_k=1)
        out[f"plain_{M}x{N}x{K}"] = rel(got, F.silu(y))
        keep[f"plain_{M}x{N}x{K}"] = got
    out["n_plain"] = gen() - n0
    n0 = gen()
    # GEGLU / RMSHEAD with ragged M and N % 64 == 0 but not % 256
    for M, N, K in [(300, 384, 128), (462, 1536 + 128, 256)]:
        a, w, b = rnd((M, K), 4), rnd((N, K), 5, K ** -0.5), rnd((N,), 6)
        y = a.double() @ w.double().T + b.double()
        got = ops.gemm(a, geglu_pack(w), geglu_pack(b), epilogue=ops.EPI_GEGLU)
        out[f"geglu_{M}x{N}x{K}"] = rel(got, y[:, :N // 2] * F.gelu(y[:, N // 2:]))
        keep[f"geglu_{M}x{N}x{K}"] = got
    for M, heads, K in [(300, 2, 128), (154 * 3, 6, 384)]:
        D = heads * 64
        a, w, b = rnd((M, K), 7), rnd((3 * D, K), 8, K ** -0.5), rnd((3 * D,), 9)
        rms = (1.0 + 0.1 * torch.randn(2 * D, generator=torch.Generator().manual_seed(10))).to(dev).to(bf16)
        y = a.double() @ w.double().T + b.double()
        qk = y[:, :2 * D].view(M, 2 * heads, 64)
        qk = qk * torch.rsqrt(qk.pow(2).mean(-1, keepdim=True) + 1e-6) * rms.double().view(2 * heads, 64)
        got = ops.gemm(a, w, b, epilogue=ops.EPI_RMSHEAD, rms_w=rms, rms_ncols=2 * D, rms_eps=1e-6)
        out[f"rmshead_{M}x{heads}x{K}"] = rel(got, torch.cat([qk.reshape(M, 2 * D), y[:, 2 * D:]], 1))
        keep[f"rmshead_{M}x{heads}x{K}"] = got
    out["n_geglu_rmshead"] = gen() - n0
    n0 = gen()
    # RESID on the bf16 and on the fp32 stream with ragged M (the context stream: 154 tokens per image), in place
    for M, N, K, rpg in [(154 * 3, 1536, 1536, 154), (300, 320, 128, 100)]:
        a, w, b = rnd((M, K), 11), rnd((N, K), 12, K ** -0.5), rnd((N,), 13)
        groups = (M + rpg - 1) // rpg
        gate = rnd((groups, N), 14)
        alpha = torch.rand(groups, generator=torch.Generator().manual_seed(15)).to(dev)
        rows = torch.arange(M, device=dev) // rpg
        y = a.double() @ w.double().T + b.double()
        al = alpha.double()[rows][:, None]
        for name, dt, tol in (("bf16", bf16, TOL), ("fp32", f32, TOL32)):
            res, blend = rnd((M, N), 16, dtype=dt), rnd((M, N), 17, dtype=dt)
            kw = (lambda t: dict(out32=t, mirror=False, split_k=1)) if dt == f32 else (lambda t: dict(out=t, split_k=1))
            r1 = res.clone()
            ops.gemm(a, w, b, epilogue=ops.EPI_RESID, gate=gate, rows_per_gate=rpg, res=r1, **kw(r1))
            bl = blend.clone()
            ops.gemm(a, w, b, epilogue=ops.EPI_RESID, res=res, blend=bl, alpha=alpha, rows_per_alpha=rpg, **kw(bl))
            r3 = res.clone()
            ops.gemm(a, w, None, epilogue=ops.EPI_RESID, res=r3, **kw(r3))
            out[f"resid_{name}_{M}x{N}x{K}"] = {
                "gate": rel(r1, res.double() + gate.double()[rows] * y),
                "blend": rel(bl, al * blend.double() + (1 - al) * (res.double() + y)),
                "plain": rel(r3, res.double() + a.double() @ w.double().T), "tol": tol}
            keep[f"resid_{name}_{M}x{N}x{K}"] = torch.stack([r1.float(), bl.float(), r3.float()])
    out["n_resid"] = gen() - n0
    n0 = gen()
    # the residual row per image (res_mod < 0), full tiles and ragged
    for M, N, K, per in [(512, 256, 128, 64), (300, 320, 192, 50)]:
        a, w, b = rnd((M, K), 18), rnd((N, K), 19, K ** -0.5), rnd((N,), 20)
        rowres = rnd(((M + per - 1) // per, N), 21)
        rows = torch.arange(M, device=dev) // per
        got = ops.gemm(a, w, b, epilogue=ops.EPI_RESID, res=rowres, res_mod=-per, split_k=1)
        out[f"rowres_{M}x{N}x{K}"] = rel(got, a.double() @ w.double().T + b.double() + rowres.double()[rows])
        keep[f"rowres_{M}x{N}x{K}"] = got
    out["n_rowres"] = gen() - n0
    n0 = gen()
    # implicit 3x3 convolution over the padded token grid (A row map + 9 taps), dense and stride 2, compact output
    for I, h, w_, C, N in [(3, 16, 28, 128, 320), (2, 4, 6, 64, 192), (7, 9, 5, 320, 640), (2, 8, 12, 192, 256)]:
        grid = ops.PaddedGrid(I, h, w_)
        x = rnd((I, C, h, w_), 22)
        wt, b = rnd((N, C, 3, 3), 23, (9 * C) ** -0.5), rnd((N,), 24)
        idx = grid.interior_index().to(dev)
        xp = torch.zeros((grid.rows, C), dtype=bf16, device=dev)
        xp[idx] = x.permute(0, 2, 3, 1).reshape(-1, C)
        wp = wt.permute(0, 2, 3, 1).reshape(N, 9 * C).contiguous()
        got = ops.gemm(xp, wp, b, act=ops.ACT_RELU, a_grid=grid, conv3x3=True, split_k=1)
        ref = F.relu(F.conv2d(x.float(), wt.float(), b.float(), padding=1)).permute(0, 2, 3, 1).reshape(-1, N)
        out[f"conv3x3_{I}x{h}x{w_}x{C}x{N}"] = rel(got, ref)
        keep[f"conv3x3_{I}x{h}x{w_}x{C}x{N}"] = got
        if h % 2 == 0 and w_ % 2 == 0:
            got2 = ops.gemm(xp, wp, b, a_grid=grid, conv3x3=True, stride2="sym", split_k=1)
            ref2 = F.conv2d(x.float(), wt.float(), b.float(), padding=1, stride=2).permute(0, 2, 3, 1).reshape(-1, N)
            out[f"conv3x3s2_{I}x{h}x{w_}x{C}x{N}"] = rel(got2, ref2)
            keep[f"conv3x3s2_{I}x{h}x{w_}x{C}x{N}"] = got2
    out["n_conv"] = gen() - n0
    out["n_total_4w"] = int(lib.dwm_gemm4w_launches())
    torch.cuda.synchronize()
    torch.save({k: v.cpu() for k, v in keep.items()}, os.environ["DWM_BATTERY_OUT"])
    print("GEMM4WGEN " + json.dumps(out))


def _run_battery(mode, path):
    env = dict(os.environ, DWM_BATTERY_OUT=path)
    env.pop("DWM_GEMM4W", None)
    if mode is not None:
        env["DWM_GEMM4W"] = mode
    r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("GEMM4WGEN ")]
    assert len(line) == 1, r.stdout[-2000:]
    return json.loads(line[0][10:])


pytestmark = [pytest.mark.gpu]
bf16 = torch.bfloat16
R4 = 1 << 12                                           # variant bit 12: attn_stream_kernel (one wave per SIMD, 2..5 query tiles per wave; also the default)
K12 = 1 << 13                                          # variant bit 13: keep the 12-wave attn_res_kernel where the streaming form would serve
PRE = 1 << 15                                          # variant bit 15: Q arrives with scale * log2(e) folded in


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("the gpu-marked tests need a HIP device (torch.cuda.is_available() is False)")
    from opendwm_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _rand2(I, N, Lc, D, dev, seed, scale=1.0):
    """qkv [I * N, 3 D] and cqkv [I * Lc, 3 D] as the two parts of ONE allocation (see _run)"""
    qc = _rand((I * (N + Lc), 3 * D), dev, seed, scale)
    return qc[:I * N], (qc[I * N:] if Lc else None)


def _served():
    from opendwm_amd import _lib
    return int(_lib.load().dwm_attn_stream_launches())


def _run(ops, qkv, cqkv, I, N, Lc, heads, rm, variant):
    # (both output segments in one allocation, as blocks.JointTransformerBlock passes them: attn_stream_kernel's folded-offset form; the
    #  form for segments more than +-16 GiB apart is tests/test_round6_gpu.py's - include/dwm_hip.h, dwm_attn_stream_launches)
    D = heads * 64
    both = torch.full((qkv.shape[0] + I * Lc, D), float("nan"), dtype=bf16, device=qkv.device)
    out = both[:qkv.shape[0]]
    cout = both[qkv.shape[0]:] if Lc else None
    kw = dict(q1=cqkv[:, :D], k1=cqkv[:, D:2 * D], v1=cqkv[:, 2 * D:], out1=cout) if Lc else {}
    ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out, rm, heads, variant=variant, **kw)
    return out, cout


# attn_stream_kernel (variant bit 12) serves unmasked self-attention with 8..19 query tiles of 32 (225 <= L <= 608); wave w of its 4 takes nqt / 4
# (+ 1 for w < nqt % 4) adjacent tiles.  Sequence lengths: every tile count 8..19 (each split 2..5 tiles per wave, ragged and full last
# tiles / key steps, one and two segments) + three lengths below the range (they stay on attn_res_kernel: same answers expected)
@pytest.mark.parametrize("scale", [1.0, 8.0], ids=["unit_scores", "huge_scores_fallback"])
@pytest.mark.parametrize("I,N,Lc,heads", [(2, 448, 154, 6), (2, 448, 0, 6), (2, 608, 0, 3), (2, 97, 0, 4), (1, 64, 0, 2), (2, 200, 33, 2),
                                          (1, 575, 0, 2), (2, 256, 0, 2), (1, 288, 0, 2), (2, 290, 0, 2), (2, 352, 0, 3), (1, 384, 0, 2),
                                          (1, 400, 0, 2), (1, 416, 1, 2), (1, 480, 0, 2), (1, 512, 0, 2), (1, 513, 30, 2), (1, 225, 0, 2)])
def test_attention_one_wave_per_simd_forms(dev, scale, I, N, Lc, heads):
    from opendwm_amd import ops
    D = heads * 64
    qkv, cqkv = _rand2(I, N, Lc, D, dev, 11, scale)
    rm = ops.rowmap_identity(I, N)
    f, cf = qkv.float(), (cqkv.float() if Lc else None)
    r0, r1 = _attn_ref(f[:, :D], f[:, D:2 * D], f[:, 2 * D:], rm.rows().to(dev), heads,
                       q1=cf[:, :D] if Lc else None, k1=cf[:, D:2 * D] if Lc else None, v1=cf[:, 2 * D:] if Lc else None)
    base, cbase = _run(ops, qkv, cqkv, I, N, Lc, heads, rm, K12)
    errs, same = {"old": max(rel_err(base, r0), rel_err(cbase, r1) if Lc else 0.0)}, {}
    covered = 8 <= (N + Lc + 31) // 32 <= 20                     # the streaming kernel's range: 8..20 query tiles
    for variant in (R4, R4 | (heads << 8), R4 | (1 << 8), R4 | 16, R4 | 16 | (heads << 8)):
        n0 = _served()
        out, cout = _run(ops, qkv, cqkv, I, N, Lc, heads, rm, variant)
        assert _served() - n0 == (1 if covered else 0), (variant, covered)
        errs[variant] = max(rel_err(out, r0), rel_err(cout, r1) if Lc else 0.0)
        same[variant] = bool(torch.equal(out, base) and (not Lc or torch.equal(cout, cbase)))
    _log("attention_one_wave_per_simd_forms", scale=scale, I=I, N=N, Lc=Lc, heads=heads, **{str(k): v for k, v in errs.items()},
         bit_equal_to_12_wave_kernel={str(k): v for k, v in same.items()})
    assert all(e < (TOL_KERNEL if scale == 1.0 else 3e-2) for e in errs.values()), errs


@pytest.mark.parametrize("I,N,Lc,heads,hs", [(150, 256, 40, 4, 2), (3, 448, 154, 24, 6), (40, 448, 0, 12, 1), (70, 230, 0, 8, 2), (300, 448, 154, 2, 1)])
def test_attention_one_wave_per_simd_across_item_seams(dev, I, N, Lc, heads, hs):
    """persistent workgroups walking several (problem, head group) items: table rebuilds, the Q prefetch and the image copies
    across head and item seams; repeated launches bit-identical"""
    from opendwm_amd import ops
    D = heads * 64
    served = _served
    qkv, cqkv = _rand2(I, N, Lc, D, dev, 21)                      # both input segments in one allocation (see _run)
    rm = ops.rowmap_identity(I, N)
    n0 = served()
    a = _run(ops, qkv, cqkv, I, N, Lc, heads, rm, R4 | (hs << 8))
    b = _run(ops, qkv, cqkv, I, N, Lc, heads, rm, R4 | (hs << 8))
    n1 = served()
    d = _run(ops, qkv, cqkv, I, N, Lc, heads, rm, K12 | (hs << 8))
    assert (n1 - n0, served() - n1) == (2, 0)                     # the streaming kernel served a and b, the 12-wave kernel d
    # two launches of one kernel: bit-identical.  (Round 6 saw launches differ by single bf16 roundings in ~2e-5 of the elements: with
    # the segments in SEPARATE allocations one launch's pair lay within +-16 GiB and the other's did not - which the kernel's first
    # form left to the 12-wave kernel.  profiles/README.md, round 6.)
    rep = max(rel_err(a[0], b[0]), rel_err(a[1], b[1]) if Lc else 0.0)
    if rep > 0:                                                   # where: problem / head / row histogram of the differing elements
        for nm, x, y, rows in (("seg0", a[0], b[0], N), ("seg1", a[1], b[1], Lc)) if Lc else (("seg0", a[0], b[0], N),):
            ne = (x != y).view(I, rows, heads, 64)
            idx = ne.nonzero()
            if idx.numel():
                c3 = _run(ops, qkv, cqkv, I, N, Lc, heads, rm, R4 | (hs << 8))
                z = (c3[0] if nm == "seg0" else c3[1]).view(I, rows, heads, 64)
                _log("attention_item_seams_two_launches_differ", seg=nm, I=I, N=N, Lc=Lc, heads=heads, hs=hs, elements=int(ne.sum()),
                     problems=idx[:, 0].unique().tolist()[:40], heads_hit=idx[:, 2].unique().tolist(), rows_hit=idx[:, 1].unique().tolist()[:64],
                     dims_hit=idx[:, 3].unique().numel(), max_abs=float((x.float() - y.float()).abs().max()),
                     third_equals_first=bool(torch.equal(z, x.view(I, rows, heads, 64))), third_equals_second=bool(torch.equal(z, y.view(I, rows, heads, 64))))
    assert rep == 0.0, rep
    errs = []
    for p0 in (0, I - 2):
        f = qkv[p0 * N:(p0 + 2) * N].float()
        cf = cqkv[p0 * Lc:(p0 + 2) * Lc].float() if Lc else None
        r0, r1 = _attn_ref(f[:, :D], f[:, D:2 * D], f[:, 2 * D:], ops.rowmap_identity(2, N).rows().to(dev), heads,
                           q1=cf[:, :D] if Lc else None, k1=cf[:, D:2 * D] if Lc else None, v1=cf[:, 2 * D:] if Lc else None)
        errs.append(max(rel_err(a[0][p0 * N:(p0 + 2) * N], r0), rel_err(a[1][p0 * Lc:(p0 + 2) * Lc], r1) if Lc else 0.0))
    # against the 12-wave kernel over ALL problems (the reference above covers four of them)
    whole = max(rel_err(a[0], d[0]), rel_err(a[1], d[1]) if Lc else 0.0)
    _log("attention_one_wave_per_simd_item_seams", I=I, N=N, Lc=Lc, heads=heads, hs=hs, rel=max(errs), rel_to_12_wave_all_problems=whole, rel_between_two_launches=rep,
         bit_equal_to_12_wave_kernel=bool(torch.equal(a[0], d[0])))
    assert max(errs) < TOL_KERNEL and whole < TOL_KERNEL


@pytest.mark.parametrize("I,N,Lc,heads", [(3, 448, 154, 4), (2, 300, 0, 2)])
def test_attention_prescaled_q(dev, I, N, Lc, heads):
    """variant bit 15: the producer folded scale * log2(e) into Q (one rounding instead of two) - every kernel then takes the scores as
    log2-domain; same answers as the scaled form within the rounding of Q"""
    from opendwm_amd import ops
    D = heads * 64
    qkv, cqkv = _rand2(I, N, Lc, D, dev, 31)
    rm = ops.rowmap_identity(I, N)
    f, cf = qkv.float(), (cqkv.float() if Lc else None)
    r0, r1 = _attn_ref(f[:, :D], f[:, D:2 * D], f[:, 2 * D:], rm.rows().to(dev), heads,
                       q1=cf[:, :D] if Lc else None, k1=cf[:, D:2 * D] if Lc else None, v1=cf[:, 2 * D:] if Lc else None)
    c = 0.125 * 1.4426950408889634
    qkv2, cqkv2 = qkv.clone(), (cqkv.clone() if Lc else None)
    qkv2[:, :D] = (f[:, :D] * c).to(bf16)
    if Lc:
        cqkv2[:, :D] = (cf[:, :D] * c).to(bf16)
    errs = {}
    for variant in (PRE, PRE | K12, PRE | 32, PRE | R4 | 16):
        out, cout = _run(ops, qkv2, cqkv2, I, N, Lc, heads, rm, variant)
        errs[variant] = max(rel_err(out, r0), rel_err(cout, r1) if Lc else 0.0)
    _log("attention_prescaled_q", I=I, N=N, Lc=Lc, heads=heads, **{str(k): v for k, v in errs.items()})
    assert all(e < TOL_KERNEL for e in errs.values()), errs


def test_attention_one_wave_per_simd_temporal_rowmap_multihead(dev):
    """through a strided row map (row-wise temporal attention: L = frames x row width), 24 heads in groups of 6"""
    from opendwm_amd import ops
    B, T, V, h, w, heads = 1, 16, 2, 3, 28, 24
    D = heads * 64
    rm = ops.rowmap_temporal_rowwise(B, T, V, h, w)
    R = B * T * V * h * w
    qkv = _rand((R, 3 * D), dev, 13)
    f = qkv.float()
    ref, _ = _attn_ref(f[:, :D], f[:, D:2 * D], f[:, 2 * D:], rm.rows().to(dev), heads)
    errs = {}
    for variant in (R4 | (6 << 8), R4 | (4 << 8), R4 | (1 << 8), R4, 0, K12):
        out = torch.full((R, D), float("nan"), dtype=bf16, device=dev)
        ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out, rm, heads, variant=variant)
        errs[variant] = rel_err(out, ref)
    _log("attention_one_wave_per_simd_temporal_rowmap", L=rm.L0, **{str(k): v for k, v in errs.items()})
    assert all(e < TOL_KERNEL for e in errs.values()), errs


def test_four_wave_gemm_general_form(dev, tmp_path):
    gen = _run_battery("1", str(tmp_path / "gen.pt"))
    base = _run_battery("0", str(tmp_path / "base.pt"))           # the same calls on the 8-wave kernels
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "gpu_parity.log"), "a") as f:
        f.write(json.dumps({"test": "gemm4w_general_form", **gen}) + "\n")
    # every call of the battery is one the general form covers; none ran it without the switch
    assert gen["n_plain"] == 7 and gen["n_geglu_rmshead"] == 4 and gen["n_resid"] == 2 * 2 * 3 and gen["n_rowres"] == 2, gen
    assert gen["n_conv"] == 4 + 3, gen
    assert all(base[k] == 0 for k in base if k.startswith("n_")), base
    for k, v in gen.items():
        if k.startswith("n_"):
            continue
        if isinstance(v, dict):
            tol = v.pop("tol", TOL)
            assert all(e < tol for e in v.values()), (k, v)
        else:
            assert v < TOL, (k, v)
    # against the 8-wave kernels on the same inputs: the same products in a different summation order
    a, b = torch.load(str(tmp_path / "gen.pt")), torch.load(str(tmp_path / "base.pt"))
    worst = {k: rel_err(a[k], b[k]) for k in a}
    _log("gemm4w_general_vs_8wave", **{k: v for k, v in sorted(worst.items(), key=lambda kv: -kv[1])[:5]})
    assert all(v < TOL for v in worst.values()), worst


@pytest.mark.parametrize("tt", ["rowwise", "pointwise", "full"])
def test_stacked_modulation_forward(dev, tt):
    from oracle import ctsd_oracle as O
    from tests.common import small_config, small_inputs, to_dev
    from tests.test_hip_gpu import TOL_MODEL, _bf16_round_sd, _hip_model
    cfg = small_config(temporal_attention_type=tt)
    sd = _bf16_round_sd(O.make_state_dict(small_config(), 0))
    m = _hip_model(cfg, sd, dev)
    inp = small_inputs(cfg, 0)
    inp16 = {k: (v.to(bf16).float() if v.is_floating_point() and k != "timestep" and k != "added_time_ids" else v) for k, v in inp.items()}
    ref = O.dit_forward(sd, cfg, **inp16)

    def run():
        di = to_dev(inp16, dev)
        return m(di.pop("sample"), di.pop("timestep"), **di)[0][0]
    m.stack_modulation = False
    base = run()
    m.stack_modulation = True
    got = run()
    assert torch.equal(got, run())                                   # deterministic
    e, d = rel_err(got, ref), rel_err(got, base)
    # another state dict: the stacked copy must follow (packed tensors are rebuilt after load_state_dict)
    sd2 = _bf16_round_sd(O.make_state_dict(small_config(), 1))
    m.load_state_dict(sd2)
    m.to(dev).to(bf16)
    e2 = rel_err(run(), O.dit_forward(sd2, cfg, **inp16))
    _log("stacked_modulation_forward", temporal=tt, rel_vs_oracle=e, rel_vs_default=d, bit_equal_to_default=bool(torch.equal(got, base)),
         rel_vs_oracle_after_reload=e2)
    assert e < TOL_MODEL and e2 < TOL_MODEL and d < TOL_MODEL


if __name__ == "__main__":
    _gemm_general_battery()
