"""Pin the oracle's restatement of the `diffusers==0.31.0` arithmetic against diffusers itself - ONE command on any box that has
diffusers 0.31.0 (and, for the model-level checks, the OpenDWM checkout on PYTHONPATH):

    PYTHONPATH=/path/to/OpenDWM/src python scripts/pin_with_diffusers.py [--cuda] [--out profiles/pin_with_diffusers.json]

The build container of this repository has no diffusers (no wheel, no network), so SURVEY.md §8c's row "oracle" stays "parity
unpinned for the diffusers leaves" until somebody runs this: every check below instantiates the REAL class with small
dimensions, loads the oracle's seeded state dict into it (the keys are the reference's, that is the point), feeds both the same
inputs and compares.  The script is CPU-only by default and takes well under a minute.  What it covers, and the oracle function
each check pins:

  leaf modules of the MMDiT (diffusers.models)    JointTransformerBlock (plain / dual / context_pre_only), PatchEmbed,
                                                  CombinedTimestepTextProjEmbeddings, AdaLayerNormContinuous, FeedForward (GEGLU)
                                                  -> oracle.ctsd_oracle: joint_transformer_block, patch_embed, time_text_embed, ...
  the reference models (needs dwm on the path)    dwm.models.crossview_temporal_dit.DiTCrossviewTemporalConditionModel.forward
                                                  (src/dwm/models/crossview_temporal_dit.py:372-630) -> oracle.dit_forward;
                                                  dwm.models.crossview_temporal_unet.UNetCrossviewTemporalConditionModel.forward
                                                  (src/dwm/models/crossview_temporal_unet.py:648-835) -> oracle.unet_forward
  VAEs                                            diffusers.AutoencoderKL (SD 3.5 / SD 2.1 shapes) encode / decode
                                                  -> oracle.vae_encode_moments / vae_decode;
                                                  diffusers.AutoencoderKLCogVideoX encode / decode -> oracle.cogvideox_vae_oracle
  schedulers                                      FlowMatchEulerDiscreteScheduler (sigmas + step), DPMSolverMultistepScheduler
                                                  (SD 2.1 config: tables + 2M updates), DDIMScheduler / DDPMScheduler
                                                  (step, add_noise, get_velocity)
                                                  -> oracle.flow_match_sigmas / denoise, unet_oracle.dpm_solver_*, scheduler_oracle

Exit code 0 = every executed check within tolerance; checks whose imports are missing are reported as "skipped" (and make the
exit code 2 unless --allow-skips).  The JSON written with --out is what to commit under profiles/."""
from __future__ import annotations

import argparse
import json
import os
import sys
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import cogvideox_vae_oracle as CV      # noqa: E402
from oracle import ctsd_oracle as O                # noqa: E402
from oracle import scheduler_oracle as S           # noqa: E402
from oracle import unet_oracle as U                # noqa: E402

RTOL, ATOL = 1e-4, 1e-5
RESULTS = []


def rel(a: torch.Tensor, b: torch.Tensor) -> float:
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def check(name):
    def deco(fn):
        def run(dev):
            try:
                errs = fn(dev)
                ok = all(e < 1e-4 for e in errs.values())
                RESULTS.append(dict(check=name, status="ok" if ok else "MISMATCH", rel=errs))
            except ImportError as e:
                RESULTS.append(dict(check=name, status="skipped", reason=f"{type(e).__name__}: {e}"))
            except Exception as e:                                   # a constructor / key mismatch is a finding, not a crash
                RESULTS.append(dict(check=name, status="ERROR", reason=f"{type(e).__name__}: {e}", trace=traceback.format_exc()[-1500:]))
            print(json.dumps(RESULTS[-1])[:600], flush=True)
        run.__name__ = fn.__name__
        CHECKS.append(run)
        return run
    return deco


CHECKS = []


def _sub(sd, prefix):
    return {k[len(prefix) + 1:]: v for k, v in sd.items() if k.startswith(prefix + ".")}


def _small_cfg(**over):
    from tests.common import small_config
    return small_config(**over)


# ------------------------------------------------------------------------------------------ MMDiT leaves
@check("diffusers.JointTransformerBlock (plain, dual, context_pre_only) vs oracle.joint_transformer_block")
def leaf_joint_block(dev):
    from diffusers.models.attention import JointTransformerBlock
    cfg = _small_cfg()
    sd = O.make_state_dict(cfg, 0)
    D, H = cfg["num_attention_heads"] * cfg["attention_head_dim"], cfg["num_attention_heads"]
    g = torch.Generator().manual_seed(1)
    h, c, temb = torch.randn(5, 24, D, generator=g), torch.randn(5, 10, D, generator=g), torch.randn(5, D, generator=g) * 0.5
    errs = {}
    last = cfg["num_layers"] - 1
    for i in sorted({0, last, next((j for j in range(cfg["num_layers"]) if j not in cfg["dual_attention_layers"] and j != last), 0)}):
        blk = JointTransformerBlock(dim=D, num_attention_heads=H, attention_head_dim=cfg["attention_head_dim"],
                                    context_pre_only=i == last, qk_norm=cfg.get("qk_norm"),
                                    use_dual_attention=i in cfg["dual_attention_layers"]).to(dev)
        blk.load_state_dict(_sub(sd, f"transformer_blocks.{i}"), strict=True)
        with torch.no_grad():
            out = blk(hidden_states=h.to(dev), encoder_hidden_states=c.to(dev), temb=temb.to(dev))
        rc, rh = O.joint_transformer_block(sd, f"transformer_blocks.{i}", cfg, i, h, c, temb)
        gc, gh = (None, out) if not isinstance(out, tuple) else out         # context_pre_only returns the hidden states only
        errs[f"layer{i}.hidden"] = rel(gh.cpu(), rh)
        if rc is not None:
            errs[f"layer{i}.context"] = rel(gc.cpu(), rc)
    return errs


@check("diffusers PatchEmbed / CombinedTimestepTextProjEmbeddings / AdaLayerNormContinuous vs the oracle's embeddings")
def leaf_embeddings(dev):
    from diffusers.models.embeddings import CombinedTimestepTextProjEmbeddings, PatchEmbed
    from diffusers.models.normalization import AdaLayerNormContinuous
    cfg = _small_cfg()
    sd = O.make_state_dict(cfg, 0)
    D = cfg["num_attention_heads"] * cfg["attention_head_dim"]
    g = torch.Generator().manual_seed(2)
    x = torch.randn(4, cfg["in_channels"], 8, 12, generator=g)
    pe = PatchEmbed(height=cfg["sample_size"], width=cfg["sample_size"], patch_size=cfg["patch_size"], in_channels=cfg["in_channels"],
                    embed_dim=D, pos_embed_max_size=cfg["pos_embed_max_size"]).to(dev)
    pe.load_state_dict(_sub(sd, "pos_embed"), strict=False)              # (the sin-cos table is a buffer the module builds itself)
    errs = {}
    with torch.no_grad():
        errs["patch_embed"] = rel(pe(x.to(dev)).cpu(), O.patch_embed(sd, cfg, x))
        tt = CombinedTimestepTextProjEmbeddings(embedding_dim=D, pooled_projection_dim=cfg["pooled_projection_dim"]).to(dev)
        tt.load_state_dict(_sub(sd, "time_text_embed"), strict=True)
        t, pooled = torch.tensor([3.0, 500.0, 999.0, 41.5]), torch.randn(4, cfg["pooled_projection_dim"], generator=g)
        want = O.timestep_embedding_mlp(sd, "time_text_embed.timestep_embedder", O.timesteps_sinusoid(t, 256)) \
            + O.timestep_embedding_mlp(sd, "time_text_embed.text_embedder", pooled)          # (oracle.dit_forward, "CombinedTimestepTextProjEmbeddings")
        errs["time_text_embed"] = rel(tt(t.to(dev), pooled.to(dev)).cpu(), want)
        no = AdaLayerNormContinuous(D, D, elementwise_affine=False, eps=1e-6).to(dev)
        no.load_state_dict(_sub(sd, "norm_out"), strict=True)
        hh, temb = torch.randn(4, 24, D, generator=g), torch.randn(4, D, generator=g)
        emb = O.linear(sd, "norm_out.linear", torch.nn.functional.silu(temb))
        scale, shift = emb.chunk(2, dim=1)
        errs["norm_out"] = rel(no(hh.to(dev), temb.to(dev)).cpu(), O.layer_norm_noaffine(hh) * (1 + scale)[:, None] + shift[:, None])
    return errs


# ------------------------------------------------------------------------------------------ the reference's models
def _tensors(d, dev):
    return {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in d.items()}


@check("dwm DiTCrossviewTemporalConditionModel.forward (rowwise / pointwise / full temporal, layout adapter) vs oracle.dit_forward")
def model_dit(dev):
    import dwm.models.crossview_temporal_dit as ref
    errs = {}
    adapter = dict(in_channels=6, channels=[128, 128, 128], is_downblocks=[True, False, False], num_res_blocks=2, downscale_factor=8,
                   use_zero_convs=True)
    for tag, over in (("rowwise", {}), ("pointwise", dict(temporal_attention_type="pointwise")), ("full", dict(temporal_attention_type="full")),
                      ("layout", dict(temporal_attention_type="pointwise", condition_image_adapter_config=adapter))):
        cfg = _small_cfg(**over)
        sd = O.make_state_dict(cfg, 0)
        m = ref.DiTCrossviewTemporalConditionModel(**cfg).to(dev).eval()
        missing, unexpected = m.load_state_dict(sd, strict=False)
        bad = [k for k in list(missing) + list(unexpected) if "pos_embed.pos_embed" not in k]
        if bad:
            raise KeyError(f"state-dict keys differ ({tag}): {bad[:8]}")
        from tests.common import small_inputs
        inp = small_inputs(cfg, 0)
        if tag == "layout":
            inp["condition_image_tensor"] = torch.rand(2, 3, 3, 6, 64, 96, generator=torch.Generator().manual_seed(5))
        with torch.no_grad():
            di = _tensors(inp, dev)
            y = m(di.pop("sample"), di.pop("timestep"), **di)[0][0]
            errs[tag] = rel(y.cpu(), O.dit_forward(sd, cfg, **inp))
    return errs


@check("dwm UNetCrossviewTemporalConditionModel.forward vs oracle.unet_forward")
def model_unet(dev):
    import dwm.models.crossview_temporal_unet as ref
    cfg = U.make_unet_config(block_out_channels=(128, 256, 512, 512), num_attention_heads=(2, 4, 8, 8), cross_attention_dim=128,
                             projection_class_embeddings_input_dim=11 * 256)           # (the small configuration of tests/test_unet_gpu.py)
    sd = U.make_unet_state_dict(cfg, 0)
    m = ref.UNetCrossviewTemporalConditionModel(**cfg).to(dev).eval()
    m.load_state_dict(sd, strict=True)
    inp = U.make_unet_inputs(cfg, 2, 3, 3, 16, 24, text_len=11)
    with torch.no_grad():
        di = _tensors(inp, dev)
        y = m(di.pop("sample"), di.pop("timesteps"), **di)[0][0]
    return {"unet": rel(y.cpu(), U.unet_forward(sd, cfg, **inp))}


# ------------------------------------------------------------------------------------------ VAEs
@check("diffusers.AutoencoderKL encode / decode (SD 3.5 and SD 2.1 shapes) vs oracle.vae_encode_moments / vae_decode")
def vae_2d(dev):
    from diffusers import AutoencoderKL
    errs = {}
    for tag, vcfg in (("sd35", dict(block_out_channels=(32, 64, 64, 64), latent_channels=16, layers_per_block=2, norm_num_groups=16,
                                    use_quant_conv=False, use_post_quant_conv=False)),
                      ("sd21", dict(block_out_channels=(32, 64, 64, 64), latent_channels=4, layers_per_block=2, norm_num_groups=16,
                                    use_quant_conv=True, use_post_quant_conv=True))):
        sd = O.make_vae_state_dict(vcfg, 0)
        n = len(vcfg["block_out_channels"])
        m = AutoencoderKL(in_channels=3, out_channels=3, down_block_types=("DownEncoderBlock2D",) * n, up_block_types=("UpDecoderBlock2D",) * n,
                          **vcfg).to(dev).eval()
        m.load_state_dict(sd, strict=True)
        g = torch.Generator().manual_seed(3)
        x, z = torch.randn(2, 3, 64, 96, generator=g), torch.randn(2, vcfg["latent_channels"], 8, 12, generator=g)
        with torch.no_grad():
            dist = m.encode(x.to(dev)).latent_dist
            moments = torch.cat([dist.mean, dist.logvar], 1).cpu()
            errs[tag + ".encode"] = rel(moments, _moments(O.vae_encode_moments(sd, vcfg, x)))
            errs[tag + ".decode"] = rel(m.decode(z.to(dev)).sample.cpu(), O.vae_decode(sd, vcfg, z))
    return errs


def _moments(m: torch.Tensor) -> torch.Tensor:
    """DiagonalGaussianDistribution clamps logvar to [-30, 20]"""
    mean, logvar = m.chunk(2, dim=1)
    return torch.cat([mean, logvar.clamp(-30.0, 20.0)], 1)


@check("diffusers.AutoencoderKLCogVideoX encode / decode (frame chunks, conv caches) vs oracle.cogvideox_vae_oracle")
def vae_cogvideox(dev):
    from diffusers import AutoencoderKLCogVideoX
    cfg = CV.make_cogvideox_config(block_out_channels=(32, 64, 64, 64), layers_per_block=1, norm_num_groups=16)
    sd = CV.make_state_dict(cfg, 0)
    n = len(cfg["block_out_channels"])
    m = AutoencoderKLCogVideoX(in_channels=3, out_channels=3, down_block_types=("CogVideoXDownBlock3D",) * n,
                               up_block_types=("CogVideoXUpBlock3D",) * n, block_out_channels=cfg["block_out_channels"],
                               latent_channels=cfg["latent_channels"], layers_per_block=cfg["layers_per_block"],
                               norm_eps=cfg["norm_eps"], norm_num_groups=cfg["norm_num_groups"],
                               temporal_compression_ratio=cfg["temporal_compression_ratio"], use_quant_conv=False,
                               use_post_quant_conv=False).to(dev).eval()
    m.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(4)
    x, z = torch.randn(1, 3, 17, 32, 48, generator=g), torch.randn(1, cfg["latent_channels"], 5, 4, 6, generator=g)
    with torch.no_grad():
        dist = m.encode(x.to(dev)).latent_dist
        errs = {"encode": rel(torch.cat([dist.mean, dist.logvar], 1).cpu(), _moments(CV.encode_moments(sd, cfg, x))),
                "decode": rel(m.decode(z.to(dev)).sample.cpu(), CV.decode(sd, cfg, z))}
    return errs


# ------------------------------------------------------------------------------------------ schedulers
@check("FlowMatchEulerDiscreteScheduler (shift 3: sigmas, timesteps, step) vs oracle.flow_match_sigmas + the Euler update")
def sched_flow_match(dev):
    from diffusers import FlowMatchEulerDiscreteScheduler
    s = FlowMatchEulerDiscreteScheduler(num_train_timesteps=1000, shift=3.0)
    s.set_timesteps(40)
    sig = O.flow_match_sigmas(40, 3.0)
    errs = {"sigmas": rel(s.sigmas.float().cpu(), sig.float()), "timesteps": rel(s.timesteps.float().cpu(), sig[:-1].float() * 1000.0)}
    g = torch.Generator().manual_seed(6)
    x, v = torch.randn(2, 16, 8, 12, generator=g), torch.randn(2, 16, 8, 12, generator=g)
    ours = x.clone()
    for i, t in enumerate(s.timesteps[:5]):
        x = s.step(v, t, x).prev_sample
        ours = ours + (sig[i + 1] - sig[i]) * v
    errs["five_steps"] = rel(x, ours)
    return errs


@check("DPMSolverMultistepScheduler (SD 2.1 config) vs unet_oracle.dpm_solver_tables / dpm_solver_coefficients")
def sched_dpm(dev):
    from diffusers import DPMSolverMultistepScheduler
    errs = {}
    for pt in ("epsilon", "v_prediction"):
        s = DPMSolverMultistepScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                                        prediction_type=pt, solver_order=2, algorithm_type="dpmsolver++", timestep_spacing="linspace",
                                        final_sigmas_type="zero")
        n = 12
        s.set_timesteps(n)
        ts, sig = U.dpm_solver_tables(n)
        errs[pt + ".timesteps"] = rel(s.timesteps.float(), ts.float())
        errs[pt + ".sigmas"] = rel(s.sigmas.float(), sig.float())
        g = torch.Generator().manual_seed(7)
        x = torch.randn(2, 4, 8, 12, generator=g).double()
        ours, x0_prev = x.clone(), None
        for i, t in enumerate(s.timesteps):
            out = torch.randn(2, 4, 8, 12, generator=g).double()
            x = s.step(out, t, x).prev_sample
            kx, ko, A, B, C = U.dpm_solver_coefficients(sig, i, pt)
            x0 = kx * ours + ko * out
            ours = A * ours + B * x0 + (C * x0_prev if x0_prev is not None else 0.0)
            x0_prev = x0
        errs[pt + ".loop"] = rel(x, ours)
    return errs


@check("DDIMScheduler.step / DDPMScheduler.add_noise, get_velocity vs oracle.scheduler_oracle")
def sched_ddim(dev):
    from diffusers import DDIMScheduler, DDPMScheduler
    errs = {}
    g = torch.Generator().manual_seed(8)
    x0, noise = torch.randn(3, 4, 8, 12, generator=g), torch.randn(3, 4, 8, 12, generator=g)
    t = torch.tensor([10, 500, 990])
    d = DDPMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear")
    errs["add_noise"] = rel(d.add_noise(x0, noise, t), S.add_noise(d.alphas_cumprod, x0, noise, t))
    errs["get_velocity"] = rel(d.get_velocity(x0, noise, t), S.get_velocity(d.alphas_cumprod, x0, noise, t))
    for pt in ("epsilon", "v_prediction", "sample"):
        s = DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", prediction_type=pt,
                          clip_sample=False, set_alpha_to_one=False)
        s.set_timesteps(20)
        out = torch.randn(3, 4, 8, 12, generator=g)
        ts = int(s.timesteps[4])
        got = s.step(out, ts, x0, eta=0.0).prev_sample
        want = S.ddim_step(s.alphas_cumprod, s.final_alpha_cumprod, 1000, 20, pt, out, torch.full((3,), ts), x0)
        errs["ddim." + pt] = rel(got, want[0] if isinstance(want, (tuple, list)) else want)
    return errs


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--cuda", action="store_true", help="run the real modules on cuda:0 (default: CPU, fp32)")
    ap.add_argument("--out", default=None, help="write the results as JSON (commit it under profiles/)")
    ap.add_argument("--allow-skips", action="store_true")
    args = ap.parse_args()
    try:
        import diffusers
        ver = diffusers.__version__
    except ImportError:
        print("diffusers is not importable here: nothing can be pinned (this is the state of the build container).", file=sys.stderr)
        return 3
    if ver != "0.31.0":
        print(f"warning: diffusers {ver} (the reference pins 0.31.0, SURVEY.md Appendix A)", file=sys.stderr)
    dev = torch.device("cuda:0" if args.cuda else "cpu")
    torch.manual_seed(0)
    for run in CHECKS:
        run(dev)
    summary = dict(diffusers=ver, torch=torch.__version__, device=str(dev), results=RESULTS,
                   ok=sum(r["status"] == "ok" for r in RESULTS), skipped=sum(r["status"] == "skipped" for r in RESULTS),
                   failed=sum(r["status"] in ("MISMATCH", "ERROR") for r in RESULTS))
    if args.out:
        json.dump(summary, open(args.out, "w"), indent=1)
    print(json.dumps({k: v for k, v in summary.items() if k != "results"}))
    if summary["failed"]:
        return 1
    return 2 if summary["skipped"] and not args.allow_skips else 0


if __name__ == "__main__":
    sys.exit(main())
