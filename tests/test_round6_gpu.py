"""GPU leg, round 6:
  * dwm_block_permute (elementwise.hip): the pack / unpack of the frame-shard all-to-all (opendwm_amd/sharding.py) as ONE HIP launch per
    direction - against the torch permutes it replaced, both directions, bf16 and fp32, block sizes down to one 16-byte chunk;
  * FrameShard.frames_to_rows / rows_to_frames on the device through a one-rank group: identity, through the HIP kernel.
(The streaming attention kernel of round 6 - attention_stream.hip - is covered by tests/test_round5_kernels_gpu.py, whose
one-wave-per-SIMD cases it took over, and, as the default for 225 <= L <= 608, by every model-level test.)"""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = [pytest.mark.gpu]


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("the gpu-marked tests need a HIP device (torch.cuda.is_available() is False)")
    from opendwm_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("B,Tl,V,R,hl,width,D", [(1, 2, 6, 8, 2, 56, 1536), (2, 3, 2, 4, 2, 5, 8), (1, 1, 1, 2, 1, 1, 8), (2, 4, 3, 2, 8, 28, 192)])
def test_block_permute_matches_torch_permutes(dev, dtype, B, Tl, V, R, hl, width, D):
    from opendwm_amd.sharding import _permute_blocks
    blk = hl * width * D
    g = torch.Generator().manual_seed(1)
    h = torch.randn(B * Tl * V * R * blk, generator=g).to(dev).to(dtype)
    # frames_to_rows: pack, unpack; rows_to_frames: pack, unpack
    send = _permute_blocks(h, (R, B, Tl, V), (1, Tl * V * R, V * R, R), blk)
    assert torch.equal(send.view(R, B, Tl, V, hl, width, D), h.view(B, Tl, V, R, hl, width, D).permute(3, 0, 1, 2, 4, 5, 6))
    hx = _permute_blocks(h, (B, R, Tl, V), (Tl * V, B * Tl * V, V, 1), blk)
    assert torch.equal(hx.view(B, R, Tl, V, hl, width, D), h.view(R, B, Tl, V, hl, width, D).permute(1, 0, 2, 3, 4, 5, 6))
    send2 = _permute_blocks(h, (R, B, Tl, V), (Tl * V, R * Tl * V, V, 1), blk)
    assert torch.equal(send2.view(R, B, Tl, V, hl, width, D), h.view(B, R, Tl, V, hl, width, D).permute(1, 0, 2, 3, 4, 5, 6))
    out = torch.empty_like(h)
    h2 = _permute_blocks(h, (B, Tl, V, R), (Tl * V, V, 1, B * Tl * V), blk, out=out)
    assert h2.data_ptr() == out.data_ptr()
    assert torch.equal(h2.view(B, Tl, V, R, hl, width, D), h.view(R, B, Tl, V, hl, width, D).permute(1, 2, 3, 0, 4, 5, 6))


def test_block_permute_rejects_bad_arguments(dev):
    from opendwm_amd import ops
    a = torch.zeros(64, device=dev, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError):
        ops.block_permute(a, torch.zeros(64, device=dev, dtype=torch.bfloat16), (2, 2, 2, 2), (8, 4, 2, 1), 3)          # 6-byte blocks
    with pytest.raises(RuntimeError):
        ops.block_permute(a, torch.zeros(32, device=dev, dtype=torch.bfloat16), (2, 2, 2, 1), (4, 2, 1, 1), 8)          # sizes differ
    with pytest.raises(RuntimeError):
        ops.block_permute(a.cpu(), a.cpu(), (2, 2, 2, 1), (4, 2, 1, 1), 8)


@pytest.mark.parametrize("I,N,Lc,heads,hs", [(6, 448, 154, 4, 2), (150, 256, 40, 4, 2), (3, 300, 33, 2, 1)])
def test_attention_stream_segments_far_apart_bit_identical(dev, I, N, Lc, heads, hs):
    """attn_stream_kernel folds the distance between the two segments of a joint launch into 32-bit row offsets while it fits (+-16 GiB)
    and otherwise runs its FAR instantiation (entries relative to each segment, the displacement added per row).  Inputs AND outputs
    with the segments 20 GiB apart - what separately allocated tensors on a 288-GB device can be - against the same launch with the
    segments adjacent: the streaming kernel serves both (dwm_attn_stream_launches), the results are bit-identical, and both are
    right against the reference."""
    from opendwm_amd import _lib, ops
    from tests.test_hip_gpu import TOL_KERNEL, _attn_ref, _rand
    from tests.common import rel_err
    bf16 = torch.bfloat16
    D = heads * 64
    n0, m0, n1, m1 = I * N * 3 * D, I * N * D, I * Lc * 3 * D, I * Lc * D
    gap = 10 * 2 ** 30                                                          # elements: 20 GiB
    torch.cuda.empty_cache()
    if torch.cuda.mem_get_info()[0] < 2 * (n0 + m0 + n1 + m1 + gap) + (4 << 30):
        pytest.skip("less than 24 GiB of free device memory")
    served = lambda: int(_lib.load().dwm_attn_stream_launches())
    src = _rand((I * (N + Lc), 3 * D), dev, 41)
    rm = ops.rowmap_identity(I, N)

    def place(buf, far):
        off = n0 + m0 + (gap if far else 0)
        qkv, out = buf[:n0].view(I * N, 3 * D), buf[n0:n0 + m0].view(I * N, D)
        cqkv, cout = buf[off:off + n1].view(I * Lc, 3 * D), buf[off + n1:off + n1 + m1].view(I * Lc, D)
        qkv.copy_(src[:I * N]); cqkv.copy_(src[I * N:])
        out.fill_(float("nan")); cout.fill_(float("nan"))
        return qkv, cqkv, out, cout

    res = {}
    for far in (False, True):
        buf = torch.empty(n0 + m0 + n1 + m1 + (gap if far else 0), dtype=bf16, device=dev)
        qkv, cqkv, out, cout = place(buf, far)
        n_before = served()
        ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out, rm, heads, variant=hs << 8,
                      q1=cqkv[:, :D], k1=cqkv[:, D:2 * D], v1=cqkv[:, 2 * D:], out1=cout)
        torch.cuda.synchronize()
        assert served() - n_before == 1, far                                    # the streaming kernel, near and far
        res[far] = (out.clone(), cout.clone())
        del buf, qkv, cqkv, out, cout
        torch.cuda.empty_cache()
    assert torch.equal(res[False][0], res[True][0]) and torch.equal(res[False][1], res[True][1])
    p0 = I - 2                                                                  # the last two problems against the reference
    f, cf = src[p0 * N:I * N].float(), src[I * N + p0 * Lc:].float()
    r0, r1 = _attn_ref(f[:, :D], f[:, D:2 * D], f[:, 2 * D:], ops.rowmap_identity(2, N).rows().to(dev), heads,
                       q1=cf[:, :D], k1=cf[:, D:2 * D], v1=cf[:, 2 * D:])
    e = max(rel_err(res[True][0][p0 * N:], r0), rel_err(res[True][1][p0 * Lc:], r1))
    assert e < TOL_KERNEL, e
