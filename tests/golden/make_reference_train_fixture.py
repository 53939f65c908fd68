"""Golden vector from EXECUTING the reference training step - runs only where /root/reference exists.

The REAL CrossviewTemporalSD.train_step (src/dwm/pipelines/ctsd.py:1195-1437, SD 3 branch: logit-normal timestep draw
:1255-1262 via sd3_compute_density_for_timestep_sampling :807-831, sigma lookup sd3_get_sigmas :833-843, flow-matching
pair :1267-1272, try_make_input_for_prediction :619-741, x0 prediction + MSE :1355-1370, backward, optimizer step) is
called on a hand-built instance (ctsd.py imported behind import-only stubs, see make_reference_driver_fixtures.py).
Faked: the VAE (a fixed average pooling as `latent_dist.sample()`), the image processor (x -> 2x - 1), get_conditions,
the denoiser (one learnable scale on a cheap function of its inputs; records what it is called with), the training
scheduler's tables (diffusers FlowMatchEulerDiscreteScheduler constructor, restated in oracle.flow_match_train_sigmas).

usage: python tests/golden/make_reference_train_fixture.py  ->  tests/golden/reference_train_step.pt
"""
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import ctsd_oracle as O                                      # noqa: E402
from tests.golden.make_reference_driver_fixtures import _Finder          # noqa: E402


def main():
    sys.meta_path.insert(0, _Finder())
    sys.path.insert(0, "/root/reference/src")
    import diffusers
    import dwm.pipelines.ctsd as C

    class FakeSD3(diffusers.SD3Transformer2DModel):
        depth_net = None

        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.tensor(0.3))
            self.seen, self.seen_kw = [], []

        def forward(self, x, ts, c=None, **kw):
            self.seen.append((x.detach().clone(), ts.detach().clone()))
            self.seen_kw.append({k: v.detach().clone() for k, v in kw.items() if torch.is_tensor(v)})
            return [self.w * (x + 1e-3 * ts[..., None, None, None] + 0.05 * c[..., None, None, None])], None, None

    def get_conditions(model, te, tok, common_config, latent_shape, batch, device, dtype, *a, **kw):
        return {"c": batch["c"]}
    C.CrossviewTemporalSD.get_conditions = staticmethod(get_conditions)

    class FakeVae:
        config = types.SimpleNamespace(shift_factor=0.1, scaling_factor=1.5)

        def encode(self, x):
            return types.SimpleNamespace(latent_dist=types.SimpleNamespace(sample=lambda: torch.nn.functional.avg_pool2d(x, 8)))

    class FakeClipVae:
        """temporal stand-in: "(b v) c t h w" clips -> [N, 3, t, h/8, w/8] with a per-frame offset (frame order matters)"""
        config = types.SimpleNamespace(shift_factor=None, scaling_factor=0.8)

        def encode(self, x):
            y = torch.nn.functional.avg_pool3d(x, (1, 8, 8)) + torch.arange(x.shape[2], dtype=x.dtype).view(1, 1, -1, 1, 1) * 0.05
            return types.SimpleNamespace(latent_dist=types.SimpleNamespace(sample=lambda: y))

    out = {}
    for name, tcfg in (("plain", {}), ("loss_coef", {"loss_coef_dict": {"sd": 0.5}, "max_norm_for_grad_clip": 0.01}),
                       ("temporal_vae", {"_temporal": True, "_meb": 1}),
                       # per-frame timesteps + the task mixer of the diffusion-forcing checkpoints (ctsd.py:1232-1237, 643-664)
                       ("df_style", {"_common": {"frame_prediction_style": "diffusion_forcing"}, "image_generation_ratio": 0.5,
                                     "reference_frame_scale_std": 0.02, "reference_frame_offset_std": 0.02}),
                       # prediction task with clean reference frames excluded from the loss (:666-737, 1363-1367)
                       ("ctsd_style", {"_common": {"frame_prediction_style": "ctsd"}, "reference_frame_count": 2,
                                       "all_reference_visible_ratio": 0.5, "reference_visible_rate": 0.7, "generation_task_ratio": 0.3,
                                       "image_generation_ratio": 0.5, "disable_reference_frame_loss": True})):
        p = object.__new__(C.CrossviewTemporalSD)
        p.model = FakeSD3()
        p.model_wrapper = p.model
        tcfg = dict(tcfg)
        common = tcfg.pop("_common", {})
        temporal = tcfg.pop("_temporal", False)
        meb = tcfg.pop("_meb", -1)
        p.vae = FakeClipVae() if temporal else FakeVae()
        p.is_temporal_vae = temporal
        p.image_processor = types.SimpleNamespace(preprocess=lambda x: x * 2 - 1)
        p.common_config, p.training_config, p.inference_config = dict(common, memory_efficient_batch=meb), dict(tcfg), {}
        p.get_reference_latent_count = lambda _p=p: C.CrossviewTemporalSD.get_reference_latent_count(_p)
        p.get_latent_sequence_length = lambda n, _p=p: C.CrossviewTemporalSD.get_latent_sequence_length(_p, n)
        p.device, p.model_dtype = torch.device("cpu"), torch.float32
        p.generator = torch.Generator().manual_seed(5)
        sig = O.flow_match_train_sigmas()
        p.train_scheduler = types.SimpleNamespace(config=types.SimpleNamespace(num_train_timesteps=1000), timesteps=sig * 1000, sigmas=sig)
        p.text_encoders = p.tokenizers = p.text_encoder = p.tokenizer = None
        p.loss_report_list = []
        p.optimizer = torch.optim.SGD(p.model.parameters(), lr=0.1)
        p.lr_scheduler = None
        p.step_duration = 0.0
        p.distribution_framework = "ddp"
        B, T, V = 2, 3, 2
        g = torch.Generator().manual_seed(9)
        batch = {"vae_images": torch.rand(B, T, V, 3, 32, 48, generator=g), "c": torch.randn(B, T, V, generator=g)}
        torch.manual_seed(1234)                     # sd3_compute_density_for_timestep_sampling draws from the global generator
        C.CrossviewTemporalSD.train_step(p, batch, 0)
        x_t, ts = p.model.seen[0]
        out[name] = dict(batch=batch, training_config=dict(tcfg), common_config=dict(common), seen_kwargs=p.model.seen_kw[0], temporal_vae=temporal, memory_efficient_batch=meb, generator_seed=5, global_seed=1234, noisy_latents=x_t, timesteps=ts,
                         loss=torch.tensor(p.loss_report_list[0]["loss"]), w_before=torch.tensor(0.3), w_after=p.model.w.detach().clone(), lr=0.1)
        print(name, "loss", p.loss_report_list[0]["loss"], "w", float(p.model.w), "timesteps", ts[:, 0, 0].tolist())
    mixer = {}
    gl = torch.Generator().manual_seed(21)
    Bm, Tm, Vm = 3, 4, 2
    lat_m, noisy_m = torch.randn(Bm, Tm, Vm, 2, 3, 3, generator=gl), torch.randn(Bm, Tm, Vm, 2, 3, 3, generator=gl)
    ts_m = torch.rand(Bm, Tm, Vm, generator=gl) * 1000
    for name, common, tcfg, rlc in (
            ("none_with_augment_draws", {}, {"reference_frame_scale_std": 0.1, "reference_frame_offset_std": 0.1}, 0),
            ("diffusion_forcing", {"frame_prediction_style": "diffusion_forcing"},
             {"image_generation_ratio": 0.5, "reference_frame_scale_std": 0.1, "reference_frame_offset_std": 0.05}, 0),
            ("ctsd_int", {"frame_prediction_style": "ctsd"},
             {"generation_task_ratio": 0.3, "image_generation_ratio": 0.5, "all_reference_visible_ratio": 0.4, "reference_visible_rate": 0.6,
              "reference_frame_scale_std": 0.1}, 2),
            ("ctsd_dict", {"frame_prediction_style": "ctsd"},
             {"generation_task_ratio": 0.2, "all_reference_visible_ratio": 0.5, "reference_visible_rate": 0.5}, {"1": 0.3, "2": 0.3, "3": 0.4})):
        r = C.CrossviewTemporalSD.try_make_input_for_prediction(noisy_m.clone(), lat_m.clone(), ts_m.clone(), tcfg, common,
                                                                generator=torch.Generator().manual_seed(33), reference_latent_count=rlc)
        mixer[name] = dict(common_config=common, training_config=tcfg, reference_latent_count=rlc, seed=33, noisy=noisy_m, latents=lat_m,
                           timesteps=ts_m, made_noisy=r[0], made_timesteps=r[1], additional=r[2], indicator=r[3])
    out["task_mixer"] = mixer
    torch.save(out, os.path.join(HERE, "reference_train_step.pt"))
    print("wrote reference_train_step.pt")


if __name__ == "__main__":
    main()
