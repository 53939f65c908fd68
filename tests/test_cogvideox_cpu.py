"""CPU leg of the CogVideoX temporal VAE: the oracle's own invariants, and that the product module tree / frame chunking /
resize index tables agree with the oracle (no compute through the HIP path here)."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cogvideox_vae_oracle as CV      # noqa: E402


def test_state_dict_keys_equal_published_tree():
    """215.6 M parameters with the diffusers 0.31.0 key names (THUDM/CogVideoX-2b vae)"""
    from opendwm_amd.vae_cogvideox import AutoencoderKLCogVideoX
    cfg = CV.make_cogvideox_config()
    with torch.device("meta"):
        m = AutoencoderKLCogVideoX()
    want = CV.param_shapes(cfg)
    got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert got == want
    assert sum(torch.Size(s).numel() for s in want.values()) == 215_583_907
    assert m.config.scaling_factor == cfg["scaling_factor"] and m.config.shift_factor is None


def test_frame_chunks_match_oracle():
    from opendwm_amd.vae_cogvideox import _chunks
    for n in (1, 2, 3, 5, 8, 9, 13, 16, 17, 33):
        for size in (2, 8):
            assert _chunks(n, size) == CV._chunks(n, size)
    assert CV._chunks(17, 8) == [(0, 9), (9, 17)] and CV._chunks(5, 2) == [(0, 3), (3, 5)]


@pytest.mark.parametrize("Tz,T", [(3, 3), (3, 5), (3, 9), (2, 2), (2, 4), (2, 8), (1, 1)])
def test_resize_frame_table_is_nearest_interpolate(Tz, T):
    from opendwm_amd.vae_cogvideox import _Ctx
    ctx = _Ctx(None, 1, "cpu")
    ctx.Tz = Tz
    zq = torch.arange(Tz, dtype=torch.float32).view(1, 1, Tz, 1, 1)
    f = torch.zeros(1, 1, T, 1, 1)
    assert ctx.zt(T) == CV._resize_like(zq, f).flatten().long().tolist()


def test_oracle_causal_conv_is_chunk_invariant():
    """conv_cache makes the chunked causal convolution equal to the convolution over the whole clip"""
    g = torch.Generator().manual_seed(0)
    sd = {"c.conv.weight": torch.randn(5, 4, 3, 3, 3, generator=g), "c.conv.bias": torch.randn(5, generator=g)}
    x = torch.randn(2, 4, 7, 6, 5, generator=g)
    whole = CV.causal_conv3d(sd, "c", x, CV.ConvCache())
    cache = CV.ConvCache()
    parts = torch.cat([CV.causal_conv3d(sd, "c", x[:, :, a:b], cache) for a, b in ((0, 3), (3, 5), (5, 7))], 2)
    assert torch.allclose(whole, parts, atol=1e-5)
    # frame t depends on frames <= t only
    x2 = x.clone()
    x2[:, :, 4:] += 1.0
    assert torch.equal(CV.causal_conv3d(sd, "c", x2, CV.ConvCache())[:, :, :4], whole[:, :, :4])


def test_oracle_shapes_and_temporal_compression():
    cfg = CV.make_cogvideox_config(block_out_channels=(32, 32, 64, 64), layers_per_block=1, norm_num_groups=8)
    sd = CV.make_state_dict(cfg, 0)
    x = torch.randn(1, 3, 17, 16, 24, generator=torch.Generator().manual_seed(1))
    m = CV.encode_moments(sd, cfg, x)
    assert m.shape == (1, 32, 5, 2, 3)
    y = CV.decode(sd, cfg, m[:, :16])
    assert y.shape == x.shape and torch.isfinite(y).all()
    # diffusion-forcing decode of ctsd.py:1611-1620: [frame, zeros] -> 8 frames, first half kept
    z2 = torch.cat([m[:, :16, :1], torch.zeros_like(m[:, :16, :1])], 2)
    assert CV.decode(sd, cfg, z2).shape[2] == 8


def test_down_up_sampling_match_torch_reference_ops():
    g = torch.Generator().manual_seed(2)
    sd = {"d.conv.weight": torch.randn(4, 4, 3, 3, generator=g), "d.conv.bias": torch.randn(4, generator=g)}
    x = torch.randn(1, 4, 5, 6, 8, generator=g)
    y = CV.downsample3d(sd, "d", x, True)
    assert y.shape == (1, 4, 3, 3, 4)
    pooled = torch.stack([x[:, :, 0], 0.5 * (x[:, :, 1] + x[:, :, 2]), 0.5 * (x[:, :, 3] + x[:, :, 4])], 2)
    ref = torch.stack([F.conv2d(F.pad(pooled[:, :, t], (0, 1, 0, 1)), sd["d.conv.weight"], sd["d.conv.bias"], stride=2) for t in range(3)], 2)
    assert torch.allclose(y, ref, atol=1e-5)
    u = CV.upsample3d(sd, "d", x, True)
    assert u.shape == (1, 4, 9, 12, 16)
