"""Golden vectors from EXECUTING the reference's generation control flow (src/dwm/pipelines/ctsd.py) - runs only where
/root/reference exists; the fixture it writes is committed.

`dwm.pipelines.ctsd` is imported behind import-only stubs of its third-party dependencies (diffusers, transformers,
torchvision, ...; see make_reference_fixtures.py) and the REAL methods are called on a hand-built instance:

  * CrossviewTemporalSD.inference_pipeline                      ctsd.py:1439-1654  (full sequence, reference frames, diffusion forcing;
                                                                its decode tail :1604-1647 also with non-trivial stand-in VAEs, 2-D and temporal)
  * CrossviewTemporalSD.autoregressive_inference_pipeline       ctsd.py:1656-1833
  * StreamingCrossviewTemporalSD.reset_streaming / inference_pipeline / send_frame_condition / receive_frame /
    fifo_inference_pipeline                                     ctsd.py:2009-2275
  * dwm.schedulers.temporal_independent.FlowMatchEulerDiscreteScheduler.step_by_indices   temporal_independent.py:176-197

What is faked (and therefore NOT pinned by these vectors): the denoiser (a cheap deterministic function of latents,
per-frame timesteps and a per-frame condition - the same one tests/test_drivers_cpu.py uses), get_conditions (the batch
already holds embedded per-frame conditions), the diffusers half of the scheduler (sigma table + Euler `step`, restated
in oracle.ctsd_oracle.flow_match_sigmas), the VAE (identity decode) and the image processor (identity).

usage: python tests/golden/make_reference_driver_fixtures.py  ->  tests/golden/reference_drivers.pt
"""
import contextlib
import importlib.abc
import importlib.machinery
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests.golden.make_reference_fixtures import _Stub    # noqa: E402

G = 3.0


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    PFX = ("diffusers", "transformers", "torchvision", "av", "tensorboard", "safetensors", "accelerate", "easydict", "PIL",
           "cv2", "timm", "lpips", "torchmetrics", "transforms3d", "nuscenes", "pyquaternion", "open3d", "bitsandbytes",
           "waymo_open_dataset", "torch_fidelity")

    def find_spec(self, name, path, target=None):
        if name.split(".")[0] in self.PFX or name == "torch.utils.tensorboard":
            return importlib.machinery.ModuleSpec(name, self)

    def create_module(self, spec):
        m = _Stub(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, m):
        pass


def fake_pred(x, ts, c, scale):
    return (0.1 * x + 1e-4 * ts[..., None, None, None] + 0.01 * c[..., None, None, None]) * scale


class FakeModel(torch.nn.Module):
    depth_net = None

    def forward(self, x, ts, c=None, scale=None, **kw):
        return [fake_pred(x.float(), ts.float(), c.float(), scale)], None, None


class FakeVae:
    dtype = torch.float32
    config = types.SimpleNamespace(scaling_factor=1.0, shift_factor=None)

    def decode(self, x, return_dict=False):
        return (x,)


class FakeImageVae:
    """2-D stand-in with scaling / shift factors and a decode that is not the identity (per-channel affine + 2x nearest)"""
    dtype = torch.float32
    config = types.SimpleNamespace(scaling_factor=0.7, shift_factor=0.1)

    def decode(self, x, return_dict=False):
        w = torch.linspace(0.5, 1.5, x.shape[1]).view(1, -1, 1, 1)
        return ((x * w).repeat_interleave(2, -1).repeat_interleave(2, -2),)


class FakeClipVae:
    """temporal stand-in: [N, C, t, h, w] -> [N, C, 2 t, h, w]; every output frame carries its own offset, so a wrong
    "(b v) c t h w" <-> "(b t v) c h w" rearrange or a wrong half of the [frame, zeros] decode changes the result"""
    dtype = torch.float32
    config = types.SimpleNamespace(scaling_factor=0.7, shift_factor=None)

    def decode(self, x, return_dict=False):
        y = x.repeat_interleave(2, 2) * 1.25
        ramp = torch.arange(y.shape[2], dtype=y.dtype).view(1, 1, -1, 1, 1) * 0.01
        return (y + ramp,)


class FakeRefVae(FakeVae):
    """identity decode (like FakeVae) + an encoder for the reference frames: [N, 3, 6, 8] -> [N, 2, 3, 4]"""
    config = types.SimpleNamespace(scaling_factor=0.7, shift_factor=0.1)

    def encode(self, x):
        p = torch.nn.functional.avg_pool2d(x, 2)
        y = p[:, :2] + 0.3 * p[:, 2:3]
        return types.SimpleNamespace(latent_dist=types.SimpleNamespace(mode=lambda: y, sample=lambda: y + 0.01))

    def decode(self, x, return_dict=False):
        return (x,)


def make_scheduler(Sched, steps):
    """the reference scheduler class with the diffusers half (sigma table, Euler step) filled in by hand"""
    from oracle import ctsd_oracle as O
    s = object.__new__(Sched)
    s.sigmas = O.flow_match_sigmas(steps)
    s.timesteps = s.sigmas[:-1] * 1000

    def set_timesteps(n, device=None):
        assert n == steps

    def step(model_output, t, sample):
        i = int((s.timesteps == t).nonzero()[0])
        return types.SimpleNamespace(prev_sample=sample.float() + (s.sigmas[i + 1] - s.sigmas[i]) * model_output.float())
    s.set_timesteps, s.step = set_timesteps, step
    return s


def make_pipeline(C, Sched, cls, steps, df, inference_config, conditions_of):
    p = object.__new__(cls)
    p.model = FakeModel()
    p.model_wrapper = p.model
    p.model_dtype = torch.float32
    p.common_config = {"frame_prediction_style": "diffusion_forcing"} if df else {}
    p.inference_config = dict(inference_config, inference_steps=steps, guidance_scale=G)
    p.device = torch.device("cpu")
    p.generator = torch.Generator().manual_seed(inference_config.get("_seed", 7))
    p.vae = FakeVae()
    p.is_temporal_vae = False
    p.image_processor = types.SimpleNamespace(postprocess=lambda x, output_type=None: x)
    p.test_scheduler = make_scheduler(Sched, steps)
    p.text_encoder = p.tokenizer = p.text_encoders = p.tokenizers = None
    return p


def main():
    sys.meta_path.insert(0, _Finder())
    sys.path.insert(0, "/root/reference/src")
    import dwm.pipelines.ctsd as C
    from dwm.schedulers.temporal_independent import FlowMatchEulerDiscreteScheduler as Sched
    import diffusers
    # the one diffusers name the reference code CALLS on these paths: a plain output record
    diffusers.schedulers.scheduling_flow_match_euler_discrete.FlowMatchEulerDiscreteSchedulerOutput = \
        lambda prev_sample: types.SimpleNamespace(prev_sample=prev_sample)

    # get_conditions: the batch already holds the CFG-doubled, embedded per-frame conditions
    def get_conditions(model, te, tok, common_config, latent_shape, batch, device, dtype, **kw):
        return {k: v for k, v in batch.items() if k != "pts"}
    C.CrossviewTemporalSD.get_conditions = staticmethod(get_conditions)

    def batch_of(B, frames, V, seed):
        g = torch.Generator().manual_seed(seed)
        return {"c": torch.randn(2 * B, frames, V, generator=g), "scale": 1.25, "pts": torch.zeros(B, frames, V)}

    out = {"guidance": G}
    B, V = 1, 2
    # ---- inference_pipeline, one window, its three modes
    T, steps = 4, 8
    shape = (B, T, V, 2, 3, 4)
    gi = torch.Generator().manual_seed(3)
    img = torch.randn(shape, generator=gi)
    single = {}
    for name, df, kw in (("full", False, {}), ("reference_frames", False, dict(image_latents=img, reference_frame_count=2)),
                         ("diffusion_forcing", True, dict(image_latents=img, start_timestep=6, stop_timestep=8, take_time=0)),
                         ("diffusion_forcing_warmup", True, dict(start_timestep=0, stop_timestep=6))):
        p = make_pipeline(C, Sched, C.CrossviewTemporalSD, steps, df, {"_seed": 11}, None)
        batch = batch_of(B, T, V, 5)
        r = C.CrossviewTemporalSD.inference_pipeline(p, shape, batch, "pt", **kw)
        single[name] = dict(kwargs={k: v for k, v in kw.items()}, batch=batch, latents=r["latents"], images=r["images"], seed=11,
                            shape=shape, steps=steps)
    out["inference_pipeline"] = single

    # ---- the decode tail of inference_pipeline (ctsd.py:1604-1647) with non-trivial stand-in VAEs: 2-D and temporal
    # ("(b v) c t h w" clips, the [frame, zeros] decode of the diffusion-forcing mode), memory_efficient_batch on / off
    tail = {}
    for name, vae, temporal, df, meb, kw in (
            ("image_full", FakeImageVae(), False, False, -1, {}),
            ("image_split2", FakeImageVae(), False, False, 2, dict(image_latents=img, reference_frame_count=1)),
            ("image_df", FakeImageVae(), False, True, -1, dict(image_latents=img, start_timestep=6, stop_timestep=8, take_time=1)),
            ("clip_full", FakeClipVae(), True, False, -1, {}),
            ("clip_split1", FakeClipVae(), True, False, 1, dict(image_latents=img, reference_frame_count=2)),
            ("clip_df", FakeClipVae(), True, True, -1, dict(image_latents=img, start_timestep=6, stop_timestep=8, take_time=2))):
        p = make_pipeline(C, Sched, C.CrossviewTemporalSD, steps, df, {"_seed": 13}, None)
        p.vae, p.is_temporal_vae = vae, temporal
        p.common_config = dict(p.common_config, memory_efficient_batch=meb)
        batch = batch_of(B, T, V, 6)
        r = C.CrossviewTemporalSD.inference_pipeline(p, shape, batch, "pt", **kw)
        tail[name] = dict(temporal=temporal, df=df, memory_efficient_batch=meb, take_time=kw.get("take_time", 0),
                          latents=r["latents"], images=r["images"])
    out["decode_tail"] = tail

    # ---- autoregressive_inference_pipeline over the REAL inference_pipeline
    ar = {}
    for name, df, cfg, total, steps in (
            # reference frames GIVEN (generate_frames_for_reference = False): encoded with vae.encode(...).latent_dist.mode()
            ("full_ref1_given", False, dict(sequence_length_per_iteration=4, reference_frame_count=1, generate_frames_for_reference=False), 10, 3),
            ("full_ref2_given_split", False, dict(sequence_length_per_iteration=4, reference_frame_count=2, generate_frames_for_reference=False,
                                                  _meb=3), 8, 3),
            ("full_ref1", False, dict(sequence_length_per_iteration=4, reference_frame_count=1), 10, 3),
            ("full_ref2", False, dict(sequence_length_per_iteration=4, reference_frame_count=2), 12, 3),
            ("df_clear0", True, dict(sequence_length_per_iteration=4, reference_frame_count=3, clear_reference_frame_count=0), 7, 8),
            ("df_clear1", True, dict(sequence_length_per_iteration=4, reference_frame_count=3, clear_reference_frame_count=1), 9, 6)):
        cfg = dict(cfg, autoregression_data_exception_for_take_sequence=["scale"], _seed=7)
        p = make_pipeline(C, Sched, C.CrossviewTemporalSD, steps, df, cfg, None)
        given = not cfg.get("generate_frames_for_reference", True)
        if given:
            p.vae = FakeRefVae()
            p.image_processor = types.SimpleNamespace(postprocess=lambda x, output_type=None: x, preprocess=lambda x: x * 2 - 1)
            p.common_config = dict(p.common_config, memory_efficient_batch=cfg.get("_meb", -1))
        calls = []
        first_ref = []
        real = C.CrossviewTemporalSD.inference_pipeline

        def spy(latent_shape, batch, output_type, image_latents=None, reference_frame_count=0, start_timestep=0,
                stop_timestep=None, take_time=0, _p=p, _calls=calls):
            _calls.append((start_timestep, stop_timestep, take_time, reference_frame_count))
            if not first_ref:
                first_ref.append(None if image_latents is None else image_latents.clone())
            return real(_p, latent_shape, batch, output_type, image_latents, reference_frame_count, start_timestep, stop_timestep, take_time)
        p.inference_pipeline = spy
        p.get_latent_sequence_length = lambda n, _p=p: C.CrossviewTemporalSD.get_latent_sequence_length(_p, n)
        batch = batch_of(B, total, V, 1)
        if given:
            batch["vae_images"] = torch.rand(B, total, V, 3, 6, 8, generator=torch.Generator().manual_seed(17))
        shape = (B, 4, V, 2, 3, 4)
        r = C.CrossviewTemporalSD.autoregressive_inference_pipeline(p, shape, batch, "pt")
        ar[name] = dict(config={k: v for k, v in cfg.items() if not k.startswith("_")}, df=df, total=total, steps=steps, batch=batch,
                        shape=shape, images=r["images"], calls=calls, seed=7, memory_efficient_batch=cfg.get("_meb", -1),
                        reference_latents=first_ref[0])
    out["autoregressive"] = ar

    # ---- streaming FIFO (every method real; only the fakes listed in the header)
    st = {}
    for name, total in (("fifo5", 5), ("fifo8", 8)):
        steps, T = 8, 4
        cfg = dict(sequence_length_per_iteration=T, autoregression_data_exception_for_take_sequence=["scale", "pts"],
                   autoregression_condition_exception_for_take_sequence=["scale"], _seed=3)
        p = make_pipeline(C, Sched, C.StreamingCrossviewTemporalSD, steps, True, cfg, None)
        p.get_autocast_context = lambda: contextlib.nullcontext()
        batch = batch_of(B, total, V, 2)
        shape = (B, T, V, 2, 3, 4)
        r = C.StreamingCrossviewTemporalSD.fifo_inference_pipeline(p, shape, batch, "pt")
        st[name] = dict(config={k: v for k, v in cfg.items() if not k.startswith("_")}, total=total, steps=steps, batch=batch, shape=shape,
                        images=r["images"], final_latents=p.latents, seed=3)
    out["streaming"] = st
    # ---- get_latent_sequence_length (ctsd.py:1113-1118) for 2-D and temporal VAEs (vae_pre / vae_stride)
    table = []
    for n, pre, stride in ((0, 0, 1), (3, 0, 1), (4, 0, 2), (0, 1, 4), (1, 1, 4), (5, 1, 4), (9, 1, 4), (17, 1, 4), (33, 1, 8)):
        q = object.__new__(C.CrossviewTemporalSD)
        q.inference_config = {"vae_pre": pre, "vae_stride": stride} if (pre, stride) != (0, 1) else {}
        table.append((n, pre, stride, C.CrossviewTemporalSD.get_latent_sequence_length(q, n)))
    out["latent_sequence_length"] = table
    torch.save(out, os.path.join(HERE, "reference_drivers.pt"))
    print("wrote reference_drivers.pt", {k: list(v["images"].shape) for k, v in ar.items()}, {k: list(v["images"].shape) for k, v in st.items()},
          {k: list(v["latents"].shape) for k, v in single.items()})


if __name__ == "__main__":
    main()
