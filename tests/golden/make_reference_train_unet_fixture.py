"""Golden vector from EXECUTING the reference training step in its SD 2.1 (UNet) branch - runs only where /root/reference
exists.

The REAL CrossviewTemporalSD.train_step (src/dwm/pipelines/ctsd.py:1195-1437) is called with a model that IS a
`diffusers.UNetSpatioTemporalConditionModel` (stub base class), so it takes the branch :1240-1253: integer timesteps from
`torch.randint(..., generator=self.generator)` right after the noise draw, `train_scheduler.add_noise`, target = noise
("epsilon") or `train_scheduler.get_velocity` ("v_prediction") - both the REAL tensor-timestep methods of
dwm.schedulers.temporal_independent.DDPMScheduler (:8-45) - the (b, t, v) timestep expansion :1273-1276,
try_make_input_for_prediction, `sd_pred[0]` taken as it is :1358-1360, MSE, backward, optimizer step.
Faked as in make_reference_train_fixture.py: the VAE, the image processor, get_conditions, the denoiser (one learnable scale).

usage: python tests/golden/make_reference_train_unet_fixture.py  ->  tests/golden/reference_train_step_unet.pt
"""
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests.golden.make_reference_driver_fixtures import _Finder          # noqa: E402
from tests.golden.make_reference_scheduler_fixture import table          # noqa: E402


def main():
    sys.meta_path.insert(0, _Finder())
    sys.path.insert(0, "/root/reference/src")
    import diffusers
    import dwm.pipelines.ctsd as C
    import dwm.schedulers.temporal_independent as S

    class FakeUNet(diffusers.UNetSpatioTemporalConditionModel):
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
        config = types.SimpleNamespace(shift_factor=None, scaling_factor=0.18215)

        def encode(self, x):
            return types.SimpleNamespace(latent_dist=types.SimpleNamespace(sample=lambda: torch.nn.functional.avg_pool2d(x, 8)))

    out = {"alphas_cumprod": table()}
    for name, pt, tcfg in (("v_prediction", "v_prediction", {}),
                           ("epsilon", "epsilon", {"loss_coef_dict": {"sd": 0.5}}),
                           ("df_style", "v_prediction", {"_common": {"frame_prediction_style": "diffusion_forcing"},
                                                          "image_generation_ratio": 0.5, "reference_frame_scale_std": 0.02}),
                           ("ctsd_style", "epsilon", {"_common": {"frame_prediction_style": "ctsd"}, "reference_frame_count": 2,
                                                      "all_reference_visible_ratio": 0.5, "reference_visible_rate": 0.7,
                                                      "generation_task_ratio": 0.3, "disable_reference_frame_loss": True})):
        p = object.__new__(C.CrossviewTemporalSD)
        p.model = FakeUNet()
        p.model_wrapper = p.model
        tcfg = dict(tcfg)
        common = tcfg.pop("_common", {})
        p.vae = FakeVae()
        p.is_temporal_vae = False
        p.image_processor = types.SimpleNamespace(preprocess=lambda x: x * 2 - 1)
        p.common_config, p.training_config, p.inference_config = dict(common, memory_efficient_batch=-1), dict(tcfg), {}
        p.get_reference_latent_count = lambda _p=p: C.CrossviewTemporalSD.get_reference_latent_count(_p)
        p.get_latent_sequence_length = lambda n, _p=p: C.CrossviewTemporalSD.get_latent_sequence_length(_p, n)
        p.device, p.model_dtype = torch.device("cpu"), torch.float32
        p.generator = torch.Generator().manual_seed(5)
        sch = object.__new__(S.DDPMScheduler)
        sch.alphas_cumprod = table()
        sch.config = types.SimpleNamespace(num_train_timesteps=1000, prediction_type=pt)
        p.train_scheduler = sch
        p.text_encoders = p.tokenizers = p.text_encoder = p.tokenizer = None
        p.loss_report_list = []
        p.optimizer = torch.optim.SGD(p.model.parameters(), lr=0.1)
        p.lr_scheduler = None
        p.step_duration = 0.0
        p.distribution_framework = "ddp"
        B, T, V = 2, 3, 2
        g = torch.Generator().manual_seed(9)
        batch = {"vae_images": torch.rand(B, T, V, 3, 32, 48, generator=g), "c": torch.randn(B, T, V, generator=g)}
        C.CrossviewTemporalSD.train_step(p, batch, 0)
        x_t, ts = p.model.seen[0]
        out[name] = dict(batch=batch, training_config=dict(tcfg), common_config=dict(common), prediction_type=pt,
                         seen_kwargs=p.model.seen_kw[0], generator_seed=5, noisy_latents=x_t, timesteps=ts,
                         loss=torch.tensor(p.loss_report_list[0]["loss"]), w_before=torch.tensor(0.3),
                         w_after=p.model.w.detach().clone(), lr=0.1)
        print(name, "loss", p.loss_report_list[0]["loss"], "w", float(p.model.w), "timesteps", ts[:, :, 0].tolist())
    torch.save(out, os.path.join(HERE, "reference_train_step_unet.pt"))
    print("wrote reference_train_step_unet.pt")


if __name__ == "__main__":
    main()
