"""End-to-end generation on synthetic weights (reduced-width models, seconds on one MI355X): the pieces a CTSD user
touches, wired together the way src/dwm/pipelines/ctsd.py wires them -

    VAE.encode(reference frames).latent_dist.mode()        ctsd.py:1677-1703  (opendwm_amd.drivers.LatentEncoder)
    autoregressive windows over the denoise loop           ctsd.py:1656-1833  (opendwm_amd.drivers.AutoregressiveDriver)
      model forward at the CFG batch + guidance + scheduler ctsd.py:1496-1575  (opendwm_amd.pipeline.CTSDDenoiser, HIP graph)
    VAE.decode(latents / scaling + shift), postprocess     ctsd.py:1606-1647  (opendwm_amd.drivers.LatentDecoder)

usage: python scripts/e2e_demo.py [--temporal-vae] [--frames 7] [--window 3] [--out /tmp/frames.pt]
       python -m torch.distributed.run --nproc-per-node R --master-addr 127.0.0.1 scripts/e2e_demo.py --window 4
           -> ONE sample over R ranks: frame-sharded denoise windows (all-to-all around the temporal blocks) and a VAE
              decode split over the ranks; RCCL when every rank has its own GPU, gloo when the ranks share one
              (the window length and the token-row count must be multiples of R)"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                     # synthetic weights / conditions (seeded, generated on the device)
from opendwm_amd.drivers import AutoregressiveDriver, LatentDecoder, LatentEncoder
from opendwm_amd.pipeline import CTSDDenoiser
from opendwm_amd.vae import AutoencoderKL
from opendwm_amd.vae_cogvideox import AutoencoderKLCogVideoX

bf16 = torch.bfloat16


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--temporal-vae", action="store_true")
    ap.add_argument("--frames", type=int, default=7)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--window", type=int, default=3, help="latent frames per window")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    group = None
    own_gpu = torch.cuda.device_count() >= world
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) if own_gpu else 0)
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl" if own_gpu else "gloo", **({"device_id": dev} if own_gpu else {}))
        group = dist.group.WORLD
    # the full module graph of the SD 3.5 CTSD model (dual blocks, cross-view + temporal VT blocks, implicit camera
    # embedding) at 2 heads x 64
    cfg = dict(bench.MODEL_KWARGS, num_layers=4, dual_attention_layers=[0, 1], num_attention_heads=2, caption_projection_dim=128,
               joint_attention_dim=128, pooled_projection_dim=64, pos_embed_max_size=32, sample_size=32,
               crossview_block_layers=[1], temporal_block_layers=[2, 3])
    model = bench.build_model(cfg, dev, seed=0)
    if a.temporal_vae:
        vae = AutoencoderKLCogVideoX(block_out_channels=(64, 64, 128, 128), layers_per_block=1, norm_num_groups=8, latent_channels=16)
    else:
        vae = AutoencoderKL(block_out_channels=(64, 64, 128, 128), layers_per_block=2, norm_num_groups=16, latent_channels=16)
    vae = vae.to(dev).to(bf16).eval()
    bench.synth_init_(vae, 1)

    B, T, V, H, W = 1, a.window, 3, 8, 16                         # latent window [B, T, V, 16, H, W]; pixels 8x
    # (the drivers slice per-frame conditions by latent frame; with the temporal VAE one window = 3 latent frames
    #  = 9 pixel frames is generated - the reference maps pixel-frame clips to latent windows in its dataset glue)
    total = T if a.temporal_vae else a.frames
    cond = bench._make_conditions(dev, 0, dict(B=B, T=total, V=V, text_len=10), 11, text_dim=128, pooled_dim=64)
    gen = torch.Generator().manual_seed(0)
    # reference frame -> latents
    ref_px = torch.rand(B, 1, V, 3, 8 * H, 8 * W, generator=gen) * 2 - 1
    image_latents = LatentEncoder(vae)(ref_px.to(dev), sample=False).contiguous()          # b t v c h w
    den = CTSDDenoiser(model, guidance_scale=4.0, inference_steps=a.steps, frame_group=group)
    if group is None:
        den.enable_graph()
    drv = AutoregressiveDriver(den, dict(inference_steps=a.steps, sequence_length_per_iteration=T, reference_frame_count=1,
                                         autoregression_data_exception_for_take_sequence=["disable_crossview", "disable_temporal",
                                                                                          "crossview_attention_mask"]),
                               decode=LatentDecoder(vae, group=group), generator=gen)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = drv.run((B, T, V, 16, H, W), cond, total, dev, image_latents=image_latents)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    img = out["images"]
    if rank == 0:
        print(json.dumps({"ranks": world, "checksum": float(img.double().sum()), "frames_x_views": int(img.shape[0]), "image_shape": list(img.shape[1:]), "seconds": round(dt, 3),
                          "min": float(img.min()), "max": float(img.max()), "finite": bool(torch.isfinite(img).all()),
                          "windows": len(drv.plan(T, total, True)), "vae": type(vae).__name__}))
    if a.out and rank == 0:
        torch.save(img.cpu(), a.out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
