"""Why does the text+layout model drift 2.5x more than the text-only model over the 40 guided Euler steps, although both have
the same single-forward error?  

Runs the 8-layer full-width proxy of tests/test_fulldepth_gpu.py (6 views x 4 frames x 32x56 latents, 40 FlowMatch-Euler steps,
guidance 4) for a set of model / adapter variants and, per variant, records against the fp32 oracle loop evaluated on the device:

  * free-running error of the latents after step 1, 5, 10, 20, 30, 40;
  * TEACHER-FORCED per-step error of the guided prediction: the product is fed the ORACLE's latents of step i and
    delta_i = v_product(x_i) - v_oracle(x_i) is kept.  cos(delta_i, delta_{i-1}) and cos(delta_i, delta_0) say whether the
    per-step errors are independent (they then add in quadrature over the steps) or one fixed bias (they add linearly);
  * what the two accumulation laws predict for step 40 from the same deltas: |sum_i dsigma_i delta_i| (what a shared bias
    gives) against sqrt(sum_i (dsigma_i |delta_i|)^2) (independent errors).

Variants: text-only / text+layout, each with row-wise and point-wise temporal attention; text+layout with the adapter
residuals taken from the fp32 oracle (rounded to bf16, or kept fp32 and added in fp32): isolates the ImageAdapter's own
arithmetic from the bf16 residual add.

usage (GPU box): python scripts/drift_bisect.py [out.json]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench                                   # noqa: E402
from opendwm_amd import ops                    # noqa: E402
from opendwm_amd.pipeline import CTSDDenoiser  # noqa: E402
from oracle import ctsd_oracle as O            # noqa: E402  (checker only)

bf16 = torch.bfloat16
dev = torch.device("cuda:0")
STEPS, G = 40, 4.0
MARKS = (1, 5, 10, 20, 30, 40)


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return (a @ b / (a.norm() * b.norm()).clamp_min(1e-30)).item()


def oracle_trajectory(sd, cfg, lat, condf):
    """latents after every step of the fp32 oracle loop (list of STEPS + 1 tensors)"""
    fwd0 = O.dit_forward
    O.dit_forward = lambda sd_, cfg_, sample, timestep, **kw: fwd0(sd_, cfg_, sample, timestep.to(sample.device), **kw)
    traj = [lat.clone()]
    try:
        with torch.no_grad():
            for i in range(STEPS):
                traj.append(O.denoise(sd, cfg, traj[-1], condf, steps=STEPS, guidance_scale=G, start=i, stop=i + 1))
    finally:
        O.dit_forward = fwd0
    return traj


def run_product(model, lat, cond, traj):
    """free-running latents at MARKS and the teacher-forced guided predictions of every step"""
    den = CTSDDenoiser(model, guidance_scale=G, inference_steps=STEPS)
    sig = den.schedule.sigmas
    with torch.no_grad():
        den.prepare(lat, cond)
        free = {}
        for i in range(STEPS):
            den.step(i)
            if i + 1 in MARKS:
                free[i + 1] = rel(den.result(), traj[i + 1])
        deltas, vnorm = [], []
        den.prepare(lat, cond)
        for i in range(STEPS):
            den.latents.copy_(traj[i])
            den._refresh_model_in()
            den.step(i)
            ds = float(sig[i + 1] - sig[i])
            v_p = (den.latents - traj[i]) / ds
            v_o = (traj[i + 1] - traj[i]) / ds
            deltas.append((v_p - v_o).float())
            vnorm.append(v_o.double().norm().item())
    return free, deltas, vnorm, sig


def analyse(name, free, deltas, vnorm, sig, traj):
    ds = [float(sig[i + 1] - sig[i]) for i in range(STEPS)]
    per_step = [d.double().norm().item() / n for d, n in zip(deltas, vnorm)]
    c_prev = [cos(deltas[i], deltas[i - 1]) for i in range(1, STEPS)]
    c_first = [cos(deltas[i], deltas[0]) for i in range(1, STEPS)]
    acc = torch.zeros_like(deltas[0], dtype=torch.float64)
    quad = 0.0
    for d, s in zip(deltas, ds):
        acc += s * d.double()
        quad += (s * d.double().norm().item()) ** 2
    xn = traj[-1].double().norm().item()
    out = dict(variant=name, free_running_rel={str(k): v for k, v in free.items()},
               teacher_forced_pred_rel=dict(first=per_step[0], mean=sum(per_step) / STEPS, last=per_step[-1]),
               cos_with_previous_step=dict(mean=sum(c_prev) / len(c_prev), min=min(c_prev), max=max(c_prev)),
               cos_with_first_step=dict(mean=sum(c_first) / len(c_first), last=c_first[-1]),
               predicted_step40_if_shared_bias=acc.norm().item() / xn, predicted_step40_if_independent=quad ** 0.5 / xn)
    print(json.dumps(out), flush=True)
    return out


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "drift_bisect.json")
    n = 8
    wl = dict(bench.WORKLOAD, T=4)
    g = torch.Generator(device="cuda").manual_seed(9)
    lat = torch.randn(1, wl["T"], wl["V"], wl["C"], wl["H"], wl["W"], device=dev, generator=g)
    results = []
    only = os.environ.get("DRIFT_ONLY")                 # e.g. "text+layout/pointwise": just that model
    for layout in (False, True):
        for tt in ("rowwise", "pointwise"):
            if only and only != ("text+layout" if layout else "text_only") + "/" + tt:
                continue
            kwargs = bench.variant_kwargs(layout)
            kwargs.update(num_layers=n, dual_attention_layers=list(range(n)), crossview_block_layers=[1, 5],
                          temporal_block_layers=[2, 3, 6, 7], temporal_attention_type=tt)
            model = bench.build_model(kwargs, dev, seed=0)
            cond = bench.make_conditions(dev, seed=3, w=wl, layout=layout)
            sd = {k: v.detach().float() for k, v in model.state_dict().items()}
            cfg = O.make_config(**kwargs)
            condf = {k: (v.float() if v.is_floating_point() else v) for k, v in cond.items()}
            traj = oracle_trajectory(sd, cfg, lat, condf)
            name = ("text+layout" if layout else "text_only") + "/" + tt
            results.append(analyse(name, *run_product(model, lat, cond, traj), traj))
            if layout:
                # the two adapter modes of the product (cached fp32 residuals above; recomputed + fused here) and the
                # round-2 adapter (bf16 skip path, bf16 residuals, bf16 add)
                model.cache_adapter_residuals = False
                results.append(analyse(name + "/recompute_fused", *run_product(model, lat, cond, traj), traj))
                model.cache_adapter_residuals = True
                run00 = model.condition_image_adapter.run
                model.condition_image_adapter.run = lambda x, precise=False: run00(x, precise=False)
                model._adapter_cache = (None, None)
                results.append(analyse(name + "/bf16_adapter_round2", *run_product(model, lat, cond, traj), traj))
                model.condition_image_adapter.run = run00
                model._adapter_cache = (None, None)
                # the adapter's residuals from the fp32 oracle instead of the bf16 HIP adapter
                with torch.no_grad():
                    feats = O.image_adapter(sd, cfg, condf["condition_image_tensor"])
                tok = [f.flatten(0, -4).permute(0, 2, 3, 1).reshape(-1, f.shape[-3]).contiguous() for f in feats]
                run0 = model.condition_image_adapter.run
                # (a) oracle residuals rounded to bf16, added by the bf16 add kernel
                model.condition_image_adapter.run = lambda x, precise=False: [t.to(bf16) for t in tok]
                model._adapter_cache = (None, None)
                results.append(analyse(name + "/oracle_residuals_bf16", *run_product(model, lat, cond, traj), traj))
                # (b) oracle residuals kept in fp32 and added in fp32 (one rounding of the sum)
                add0 = ops.add_

                def add_fp32(h, r):
                    if r.dtype == torch.float32:
                        h.copy_((h.float() + r).to(h.dtype))
                        return h
                    return add0(h, r)
                import opendwm_amd.dit as dit
                model.condition_image_adapter.run = lambda x, precise=False: list(tok)
                model._adapter_cache = (None, None)
                dit.ops.add_ = add_fp32
                try:
                    results.append(analyse(name + "/oracle_residuals_fp32_add", *run_product(model, lat, cond, traj), traj))
                    # (c) the HIP adapter's own bf16 residuals, but added in fp32 (isolates the bf16 add)
                    model.condition_image_adapter.run = lambda x, precise=False: [t.float() for t in run0(x, precise=False)]
                    model._adapter_cache = (None, None)
                    results.append(analyse(name + "/hip_residuals_fp32_add", *run_product(model, lat, cond, traj), traj))
                finally:
                    dit.ops.add_ = add0
                    model.condition_image_adapter.run = run0
                # the adapter's own error, per level
                with torch.no_grad():
                    mine = run0(cond["condition_image_tensor"], precise=True)
                    mine16 = run0(cond["condition_image_tensor"], precise=False)
                print(json.dumps({"variant": name, "adapter_residual_rel_per_level": [rel(a.float(), b) for a, b in zip(mine, tok)],
                                  "adapter_residual_rel_per_level_bf16_round2": [rel(a.float(), b) for a, b in zip(mine16, tok)],
                                  "residual_to_hidden_note": "relative to the residual itself"}), flush=True)
            del model, sd, traj
            torch.cuda.empty_cache()
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    json.dump(results, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
