"""GPU leg, OFF by default: kernels written at the end of round 4 WITHOUT a GPU at hand (the round's GPU budget was spent).  They
are opt-in in the product (an environment variable or a variant bit each; the defaults are the kernels the whole suite has run
on), and their tests are opt-in here: DWM_TEST_UNVALIDATED=1 runs them (scripts/calls/r5_a.sh does, as the first call of the
next round).  A case moves into the regular files once it has passed on hardware.

  * attn_res2_kernel (attention.hip; dwm_attn_args.variant bit 6 / DWM_ATTN_RES2=1): the resident attention kernel with two
    query tiles per wave - every shape class the resident kernel's own tests hold, against the fp32 reference AND against the
    default kernel's output (the same per-tile arithmetic in the same order: expected bit-equal, logged, not asserted).
"""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tests.common import rel_err                      # noqa: E402
from tests.test_hip_gpu import TOL_KERNEL, _attn_ref, _log, _rand      # noqa: E402

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not os.environ.get("DWM_TEST_UNVALIDATED"),
                                 reason="kernels written without a GPU at hand: DWM_TEST_UNVALIDATED=1 runs their tests")]
bf16 = torch.bfloat16
RES2 = 64                                              # variant bit 6


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("the gpu-marked tests need a HIP device (torch.cuda.is_available() is False)")
    from opendwm_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _run(ops, qkv, cqkv, I, N, Lc, heads, rm, variant):
    D = heads * 64
    out = torch.full((qkv.shape[0], D), float("nan"), dtype=bf16, device=qkv.device)
    cout = torch.full((I * Lc, D), float("nan"), dtype=bf16, device=qkv.device) if Lc else None
    kw = dict(q1=cqkv[:, :D], k1=cqkv[:, D:2 * D], v1=cqkv[:, 2 * D:], out1=cout) if Lc else {}
    ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out, rm, heads, variant=variant, **kw)
    return out, cout


# sequence lengths: every remainder class of the tile schedule (nqt = tiles of 32 queries; full rounds of 16, then rem / 8 +
# (wave < rem % 8) tiles per wave): nqt = 2, 3 (singles only), 4 (97 queries: ragged last tile), 8 (one single per wave), 9-15
# (pairs and singles mixed), 14 (L = 448: the dual / temporal case), 16 (exactly one full round), 17, 18, 19 (L = 602 / 608: a
# full round + singles: the joint case), and two segments
@pytest.mark.parametrize("scale", [1.0, 8.0], ids=["unit_scores", "huge_scores_fallback"])
@pytest.mark.parametrize("I,N,Lc,heads", [(2, 448, 154, 6), (2, 448, 0, 6), (2, 608, 0, 3), (2, 97, 0, 4), (1, 64, 0, 2), (2, 200, 33, 2),
                                          (1, 575, 0, 2), (3, 33, 32, 3), (2, 256, 0, 2), (2, 290, 0, 2), (1, 480, 0, 2), (1, 512, 0, 2),
                                          (1, 513, 30, 2), (2, 352, 0, 3), (1, 416, 1, 2)])
def test_attention_paired_resident_forms(dev, scale, I, N, Lc, heads):
    from opendwm_amd import ops
    D = heads * 64
    qkv = _rand((I * N, 3 * D), dev, 11, scale)
    cqkv = _rand((I * Lc, 3 * D), dev, 12, scale) if Lc else None
    rm = ops.rowmap_identity(I, N)
    f, cf = qkv.float(), (cqkv.float() if Lc else None)
    r0, r1 = _attn_ref(f[:, :D], f[:, D:2 * D], f[:, 2 * D:], rm.rows().to(dev), heads,
                       q1=cf[:, :D] if Lc else None, k1=cf[:, D:2 * D] if Lc else None, v1=cf[:, 2 * D:] if Lc else None)
    base, cbase = _run(ops, qkv, cqkv, I, N, Lc, heads, rm, 0)
    errs, same = {}, {}
    for variant in (RES2, RES2 | (heads << 8), RES2 | 16, RES2 | 16 | (heads << 8)):
        out, cout = _run(ops, qkv, cqkv, I, N, Lc, heads, rm, variant)
        errs[variant] = max(rel_err(out, r0), rel_err(cout, r1) if Lc else 0.0)
        same[variant] = bool(torch.equal(out, base) and (not Lc or torch.equal(cout, cbase)))
    _log("attention_paired_resident_forms", scale=scale, I=I, N=N, Lc=Lc, heads=heads, **{str(k): v for k, v in errs.items()},
         bit_equal_to_default={str(k): v for k, v in same.items()})
    assert all(e < (TOL_KERNEL if scale == 1.0 else 3e-2) for e in errs.values()), errs


@pytest.mark.parametrize("I,N,Lc,heads,hs", [(150, 256, 40, 4, 2), (3, 448, 154, 24, 6), (40, 448, 0, 12, 1), (70, 230, 0, 8, 2)])
def test_attention_paired_resident_across_item_seams(dev, I, N, Lc, heads, hs):
    """persistent workgroups walking several (problem, head group) items: table rebuilds, the Q prefetch and the copy pipeline
    across head and item seams; repeated launches bit-identical"""
    from opendwm_amd import ops
    D = heads * 64
    qkv = _rand((I * N, 3 * D), dev, 21)
    cqkv = _rand((I * Lc, 3 * D), dev, 22) if Lc else None
    rm = ops.rowmap_identity(I, N)
    a = _run(ops, qkv, cqkv, I, N, Lc, heads, rm, RES2 | (hs << 8))
    b = _run(ops, qkv, cqkv, I, N, Lc, heads, rm, RES2 | (hs << 8))
    d = _run(ops, qkv, cqkv, I, N, Lc, heads, rm, hs << 8)
    assert torch.equal(a[0], b[0]) and (not Lc or torch.equal(a[1], b[1]))
    errs = []
    for p0 in (0, I - 2):
        f = qkv[p0 * N:(p0 + 2) * N].float()
        cf = cqkv[p0 * Lc:(p0 + 2) * Lc].float() if Lc else None
        r0, r1 = _attn_ref(f[:, :D], f[:, D:2 * D], f[:, 2 * D:], ops.rowmap_identity(2, N).rows().to(dev), heads,
                           q1=cf[:, :D] if Lc else None, k1=cf[:, D:2 * D] if Lc else None, v1=cf[:, 2 * D:] if Lc else None)
        errs.append(max(rel_err(a[0][p0 * N:(p0 + 2) * N], r0), rel_err(a[1][p0 * Lc:(p0 + 2) * Lc], r1) if Lc else 0.0))
    # against the default kernel over ALL problems (the reference above covers four of them)
    whole = max(rel_err(a[0], d[0]), rel_err(a[1], d[1]) if Lc else 0.0)
    _log("attention_paired_resident_item_seams", I=I, N=N, Lc=Lc, heads=heads, hs=hs, rel=max(errs), rel_to_default_all_problems=whole,
         bit_equal_to_default=bool(torch.equal(a[0], d[0])))
    assert max(errs) < TOL_KERNEL and whole < TOL_KERNEL


def test_attention_paired_resident_temporal_rowmap_multihead(dev):
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
    for variant in (RES2 | (6 << 8), RES2 | (4 << 8), RES2):
        out = torch.full((R, D), float("nan"), dtype=bf16, device=dev)
        ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out, rm, heads, variant=variant)
        errs[variant] = rel_err(out, ref)
    _log("attention_paired_resident_temporal_rowmap", L=rm.L0, **{str(k): v for k, v in errs.items()})
    assert all(e < TOL_KERNEL for e in errs.values()), errs


def test_attention_paired_resident_rejects_wave_override(dev):
    from opendwm_amd import ops
    heads, N = 2, 128
    qkv = _rand((N, 3 * heads * 64), dev, 3)
    out = torch.zeros((N, heads * 64), dtype=bf16, device=dev)
    with pytest.raises(RuntimeError):
        ops.attention(qkv[:, :128], qkv[:, 128:256], qkv[:, 256:], out, ops.rowmap_identity(1, N), heads, variant=RES2 | 8)
