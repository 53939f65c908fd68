"""Scheduler vectors produced by EXECUTING the reference's tensor-timestep scheduler methods - runs only where
/root/reference exists.  `diffusers` is absent, so the classes are imported behind the import-only stub (their diffusers
base classes are empty) and the REAL methods are called on hand-built instances that carry exactly the attributes the
methods read (alphas_cumprod, final_alpha_cumprod, config, num_inference_steps):

  * dwm.schedulers.temporal_independent.DDPMScheduler.add_noise / get_velocity    (temporal_independent.py:8-45)
  * dwm.schedulers.temporal_independent.DDIMScheduler.step / _get_variance        (:47-170)

The alphas_cumprod table is the `scaled_linear` one of the SD 2.1 scheduler config (restated; diffusers arithmetic - unpinned).

usage: python tests/golden/make_reference_scheduler_fixture.py  ->  tests/golden/reference_schedulers.pt
"""
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests.golden.make_reference_driver_fixtures import _Finder          # noqa: E402


def table():
    betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, 0)


def main():
    sys.meta_path.insert(0, _Finder())
    sys.path.insert(0, "/root/reference/src")
    import dwm.schedulers.temporal_independent as S
    g = torch.Generator().manual_seed(0)
    acp = table()
    out = {"alphas_cumprod": acp}
    B, T, V, C, H, W = 2, 3, 2, 4, 4, 8
    x0 = torch.randn(B, T, V, C, H, W, generator=g)
    noise = torch.randn(B, T, V, C, H, W, generator=g)
    ddpm = object.__new__(S.DDPMScheduler)
    ddpm.alphas_cumprod = acp.clone()
    cases = {}
    for name, shape in (("per_sample", (B,)), ("per_frame", (B, T)), ("per_view", (B, T, V))):
        ts = torch.randint(0, 1000, shape, generator=g)
        cases[name] = dict(timesteps=ts, noisy=S.DDPMScheduler.add_noise(ddpm, x0, noise, ts),
                           velocity=S.DDPMScheduler.get_velocity(ddpm, x0, noise, ts))
    out["ddpm"] = dict(x0=x0, noise=noise, cases=cases)

    ddim_cases = {}
    sample = torch.randn(B, T, V, C, H, W, generator=g)
    mo = torch.randn(B, T, V, C, H, W, generator=g)
    vn = torch.randn(B, T, V, C, H, W, generator=g)
    for name, kw in {
        "v_eta0": dict(prediction_type="v_prediction"),
        "eps_eta0": dict(prediction_type="epsilon"),
        "sample_eta0": dict(prediction_type="sample"),
        "v_eta05": dict(prediction_type="v_prediction", eta=0.5),
        "eps_clip": dict(prediction_type="epsilon", clip_sample=True, clip_sample_range=0.8),
        "eps_clip_reuse": dict(prediction_type="epsilon", clip_sample=True, clip_sample_range=0.8, use_clipped=True, eta=0.3),
        "v_alpha_one": dict(prediction_type="v_prediction", set_alpha_to_one=True),
    }.items():
        n_inf = 50
        sch = object.__new__(S.DDIMScheduler)
        sch.alphas_cumprod = acp.clone()
        sch.final_alpha_cumprod = torch.tensor(1.0) if kw.get("set_alpha_to_one") else acp[0]
        sch.num_inference_steps = n_inf
        sch.config = types.SimpleNamespace(num_train_timesteps=1000, prediction_type=kw["prediction_type"], thresholding=False,
                                           clip_sample=kw.get("clip_sample", False), clip_sample_range=kw.get("clip_sample_range", 1.0))
        # per-(sample, frame, view) timesteps from the 'leading' grid (1, 21, ..., 981); the smallest ones step below zero
        grid = torch.arange(0, n_inf) * (1000 // n_inf) + 1
        ts = grid[torch.randint(0, n_inf, (B, T, V), generator=g)]
        ts[0, 0, 0], ts[1, 2, 1] = 1, 981
        eta = kw.get("eta", 0.0)
        prev, x0p = S.DDIMScheduler.step(sch, mo, ts, sample, eta=eta, use_clipped_model_output=kw.get("use_clipped", False),
                                         variance_noise=vn if eta > 0 else None, return_dict=False)
        tsb = ts.view(B, T, V, 1, 1, 1)
        var = S.DDIMScheduler._get_variance(sch, tsb, tsb - 1000 // n_inf)
        ddim_cases[name] = dict(kw=kw, timesteps=ts, prev_sample=prev, pred_original_sample=x0p, variance=var.view(B, T, V),
                                final_alpha_cumprod=sch.final_alpha_cumprod, num_inference_steps=n_inf)
    out["ddim"] = dict(sample=sample, model_output=mo, variance_noise=vn, cases=ddim_cases)
    torch.save(out, os.path.join(HERE, "reference_schedulers.pt"))
    print("wrote reference_schedulers.pt:", list(cases), list(ddim_cases))


if __name__ == "__main__":
    main()
