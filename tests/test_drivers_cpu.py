"""Window drivers (autoregressive / streaming FIFO generation) against the restated reference control flow
(oracle/drivers_oracle.py, ctsd.py:1656-1833, 2009-2275).  CPU only: the per-window loop is the fp32 oracle
loop (ctsd_oracle.denoise) over a cheap stand-in model, so what is compared is the window plan, the latent
carry-over, the emitted frames and the consumption order of the host random stream - bit exact."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ctsd_oracle as O          # noqa: E402
from oracle import drivers_oracle as DO      # noqa: E402
from opendwm_amd import drivers as D         # noqa: E402

G = 3.0


@pytest.fixture()
def fake_model(monkeypatch):
    def fwd(sd, cfg, sample, timestep, c=None, scale=None, **kw):
        # depends on the latent, the per-frame timestep and the per-frame condition
        y = 0.1 * sample + 1e-4 * timestep[..., None, None, None] + 0.01 * c[..., None, None, None]
        return y * (1.0 if scale is None else scale)
    monkeypatch.setattr(O, "dit_forward", fwd)


class LoopDenoiser:
    """CTSDDenoiser interface over the oracle loop"""

    def __init__(self, steps):
        self.steps, self.calls = steps, []

    def run(self, latents, conditions, stop=None, start=0, **kw):
        self.calls.append((start, stop, kw.get("take_time", 0), kw.get("reference_frame_count", 0)))
        return O.denoise(None, None, latents, conditions, self.steps, G, stop=stop, start=start, **kw)


def make_conditions(B, frames, V, seed=0):
    g = torch.Generator().manual_seed(seed)
    return {"c": torch.randn(2 * B, frames, V, generator=g), "scale": 1.25, "flag": torch.ones(2 * B)}


def oracle_window(steps, df, clear, calls):
    def window(latent_shape, cond, il, ref, start, stop, take_time, noise):
        calls.append((start, stop if stop is not None else None, take_time, ref))
        lat0 = noise if noise is not None else torch.zeros(tuple(latent_shape))
        lat = O.denoise(None, None, lat0, cond, steps, G, stop=stop, start=start, image_latents=il,
                        reference_frame_count=ref, diffusion_forcing=df, take_time=take_time,
                        clear_reference_frame_count=clear)
        img = lat[:, take_time].flatten(0, 1) if df else lat.flatten(0, 2)
        return {"latents": lat, "images": img}
    return window


@pytest.mark.parametrize("with_ref", [False, True])
@pytest.mark.parametrize("total,seq,ref", [(10, 4, 1), (9, 3, 1), (12, 4, 2), (4, 4, 1)])
def test_autoregressive_full_sequence(fake_model, with_ref, total, seq, ref):
    B, V, steps = 1, 2, 3
    shape = (B, seq, V, 2, 3, 4)
    cond = make_conditions(B, total, V)
    cfg = dict(inference_steps=steps, sequence_length_per_iteration=seq, reference_frame_count=ref,
               autoregression_data_exception_for_take_sequence=["scale"])
    il = torch.randn(B, ref, V, 2, 3, 4, generator=torch.Generator().manual_seed(5)) if with_ref else None
    calls = []
    want = DO.autoregressive(oracle_window(steps, False, 0, calls), shape, cond, total, cfg, False,
                             torch.Generator().manual_seed(7), image_latents=il)
    den = LoopDenoiser(steps)
    got = D.AutoregressiveDriver(den, cfg, generator=torch.Generator().manual_seed(7)).run(shape, cond, total, "cpu", image_latents=il)
    assert got["images"].shape == want["images"].shape
    assert torch.equal(got["images"], want["images"])
    assert len(den.calls) == len(calls)
    n_windows = len(range(0, total - seq + 1, seq - ref))
    assert len(calls) == n_windows
    emitted = (seq if not with_ref else seq - ref) + (n_windows - 1) * (seq - ref)
    assert got["images"].shape[0] == emitted * B * V


@pytest.mark.parametrize("clear,steps", [(0, 8), (1, 6), (2, 4)])
@pytest.mark.parametrize("total", [6, 9])
def test_autoregressive_diffusion_forcing(fake_model, clear, steps, total):
    B, V, T = 1, 2, 4
    shape = (B, T, V, 2, 3, 4)
    cond = make_conditions(B, total, V, seed=1)
    cfg = dict(inference_steps=steps, sequence_length_per_iteration=T, reference_frame_count=3,
               clear_reference_frame_count=clear, autoregression_data_exception_for_take_sequence=["scale"])
    calls = []
    want = DO.autoregressive(oracle_window(steps, True, clear, calls), shape, cond, total, cfg, True,
                             torch.Generator().manual_seed(11))
    den = LoopDenoiser(steps)
    drv = D.AutoregressiveDriver(den, cfg, diffusion_forcing=True, generator=torch.Generator().manual_seed(11))
    got = drv.run(shape, cond, total, "cpu")
    assert [c[:3] for c in den.calls] == [c[:3] for c in calls]          # (start, stop, take_time) of every window
    assert torch.equal(got["images"], want["images"])
    assert torch.equal(got["latents"], want["latents"])
    # every queue slot is emitted exactly once per window after the warm-up, and flushed at the end
    plan = drv.plan(T, total, False)
    assert plan[0].carry == "all" and plan[0].stop == steps - steps // (T - clear)
    assert got["images"].shape[0] == (len(plan) - 1) * B * V


def test_plan_rejects_bad_configs():
    cfg = dict(inference_steps=7, sequence_length_per_iteration=4, reference_frame_count=3)
    with pytest.raises(ValueError):
        D.AutoregressiveDriver(None, cfg, diffusion_forcing=True).plan(4, 9, False)       # 7 % 4 != 0
    with pytest.raises(ValueError):
        D.AutoregressiveDriver(None, dict(cfg, inference_steps=8), diffusion_forcing=True).plan(4, 4, False)
    with pytest.raises(ValueError):
        D.AutoregressiveDriver(None, dict(cfg, reference_frame_count=4)).plan(4, 9, False)


@pytest.mark.parametrize("total", [5, 8])
def test_streaming_fifo(fake_model, total):
    B, V, T, steps = 1, 2, 4, 8
    shape = (B, T, V, 2, 3, 4)
    cond = make_conditions(B, total, V, seed=2)
    cfg = dict(inference_steps=steps, sequence_length_per_iteration=T,
               autoregression_data_exception_for_take_sequence=["scale", "flag"],
               autoregression_condition_exception_for_take_sequence=["scale", "flag"])
    calls = []

    def window(latent_shape, conditions, latents, start, stop, take_time):
        calls.append((start, stop, take_time))
        lat = O.denoise(None, None, latents, conditions, steps, G, stop=stop, start=start, image_latents=latents,
                        diffusion_forcing=True, take_time=take_time)
        return lat, (lat[:, take_time].flatten(0, 1) if stop >= steps else None)
    want = DO.Streaming(window, cfg, torch.Generator().manual_seed(3)).fifo(shape, cond, total)
    den = LoopDenoiser(steps)
    got = D.StreamingDriver(den, cfg, generator=torch.Generator().manual_seed(3)).fifo(shape, cond, total, "cpu")
    assert [c[:3] for c in den.calls] == calls
    assert torch.equal(got, want)
    assert got.shape[0] == total * B * V                                  # one frame out per frame in


def test_streaming_protocol_errors():
    drv = D.StreamingDriver(LoopDenoiser(8), dict(inference_steps=8, sequence_length_per_iteration=4))
    with pytest.raises(RuntimeError):
        drv.send_frame_condition({})
    drv.reset_streaming((1, 4, 2, 2, 3, 4), "cpu")
    with pytest.raises(RuntimeError):
        drv.send_frame_condition(None)                                    # flush before the queue is full
    with pytest.raises(ValueError):
        D.StreamingDriver(None, dict(inference_steps=7, sequence_length_per_iteration=4)).reset_streaming((1, 4, 2, 2, 3, 4), "cpu")


def test_take_sequence_clip_and_latent_length():
    t = torch.arange(24).view(2, 6, 2)
    assert torch.equal(D.take_sequence_clip(t, 1, 3), t[:, 1:3])
    assert D.take_sequence_clip(3.5, 1, 3) == 3.5
    assert torch.equal(D.take_sequence_clip(torch.ones(4), 1, 3), torch.ones(4))
    assert D.take_sequence_clip([[1, 2, 3], [4, 5, 6]], 1, 3) == [[2, 3], [5, 6]]
    with pytest.raises(TypeError):
        D.take_sequence_clip({"a": 1}, 0, 1)
    assert D.latent_sequence_length(17, 1, 4) == 5 and D.latent_sequence_length(0, 1, 4) == 0
    assert D.latent_sequence_length(8) == 8
    with pytest.raises(ValueError):
        D.latent_sequence_length(16, 1, 4)


def test_bench_kernel_timer_summary_splits_gemm_launches_by_kernel():
    """bench.KernelTimer.summary(): totals per kind and the 4-wave / 8-wave split of the GEMM records (stand-in events)"""
    import bench

    class Ev:
        def __init__(self, t): self.t = t
        def elapsed_time(self, other): return other.t - self.t

    t = bench.KernelTimer()
    t.records = [("gemm", 2e12, Ev(0.0), Ev(2.0)), ("attn", 1e12, Ev(2.0), Ev(3.0)), ("gemm", 1e12, Ev(3.0), Ev(5.0)),
                 ("gemm", 3e12, Ev(5.0), Ev(6.0))]
    t.four_wave = [True, False, True]
    s = t.summary()
    assert s["gemm"]["launches"] == 3 and abs(s["gemm"]["ms"] - 5.0) < 1e-9 and abs(s["gemm"]["tflops"] - 6e12 / 5.0 / 1e9) < 1e-6
    by = s["gemm"]["by_kernel"]
    assert by["gemm4w_kernel"]["launches"] == 2 and abs(by["gemm4w_kernel"]["ms"] - 3.0) < 1e-9
    assert by["gemm_bf16_kernel"]["launches"] == 1 and abs(by["gemm_bf16_kernel"]["tflops"] - 1e12 / 2.0 / 1e9) < 1e-6
    assert s["attn"]["launches"] == 1
    t.four_wave = [True]                      # lengths out of step (a leg that recorded without the counter): no split, no error
    assert "by_kernel" not in t.summary()["gemm"]
