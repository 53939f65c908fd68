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
