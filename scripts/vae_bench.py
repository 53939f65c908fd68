"""Full-size SD 3.5 VAE decode: parity vs the oracle on-device (2 images) + timing (96 images)."""
import os, sys, time, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from opendwm_amd.vae import AutoencoderKL
from oracle import ctsd_oracle as O
dev = torch.device("cuda:0"); bf16 = torch.bfloat16
vcfg = dict(block_out_channels=(128, 256, 512, 512), layers_per_block=2, norm_num_groups=32, latent_channels=16)
sd = {k: v.to(bf16).float() for k, v in O.make_vae_state_dict(vcfg, 0).items()}
vae = AutoencoderKL(**vcfg); vae.load_state_dict(sd); vae = vae.to(dev).to(bf16).eval()
g = torch.Generator().manual_seed(0)
z = torch.randn(2, 16, 32, 56, generator=g).to(bf16).float().to(dev)
ref = O.vae_decode({k: v.to(dev) for k, v in sd.items()}, vcfg, z)
out = vae.decode(z)[0]
rel = ((out.double() - ref.double()).norm() / ref.double().norm()).item()
z96 = torch.randn(96, 16, 32, 56, device=dev)
for chunk in (8, 12, 24):
    vae.decode(z96[:chunk], chunk=chunk); torch.cuda.synchronize()
    t0 = time.perf_counter(); y = vae.decode(z96, chunk=chunk)[0]; torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(json.dumps({"vae_decode_96img_256x448": round(dt, 4), "chunk": chunk, "img_per_s": round(96 / dt, 1), "rel_vs_oracle_fullsize": rel, "finite": bool(torch.isfinite(y.float()).all())}))
