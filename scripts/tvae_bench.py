"""Full-size CogVideoX temporal VAE (THUDM/CogVideoX-2b widths, synthetic weights) on BASELINE configs[4] shapes:
6 views x 17 frames x 256x448 px <-> latents [6, 16, 5, 32, 56].  Timing of decode and encode + on-device fp32 oracle
parity on a reduced clip (the fp32 torch oracle at full size needs > 100 GB)."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import cogvideox_vae_oracle as CV
from opendwm_amd.vae_cogvideox import AutoencoderKLCogVideoX

dev, bf16 = torch.device("cuda:0"), torch.bfloat16
cfg = CV.make_cogvideox_config()
sd = {k: v.to(bf16).float() for k, v in CV.make_state_dict(cfg, 0).items()}
vae = AutoencoderKLCogVideoX()
vae.load_state_dict(sd)
vae = vae.to(dev).to(bf16).eval()
g = torch.Generator().manual_seed(0)
# parity at full width, 1 view x 5 latent frames x 8x14 latent (64x112 px)
z = torch.randn(1, 16, 5, 8, 14, generator=g).to(bf16).float().to(dev)
sdd = {k: v.to(dev) for k, v in sd.items()}
ref = CV.decode(sdd, cfg, z)
out = vae.decode(z)[0]
rel = ((out.float() - ref).norm() / ref.norm()).item()
x = ref.clamp(-1, 1).to(bf16).float()
mref = CV.encode_moments(sdd, cfg, x)
mout = vae.encode(x).latent_dist.parameters
rel_e = ((mout - mref).norm() / mref.norm()).item()
del ref, sdd, mref
torch.cuda.empty_cache()
V, T, H, W = 6, 5, 32, 56
zf = torch.randn(V, 16, T, H, W, generator=g).to(dev)
vae.decode(zf); torch.cuda.synchronize()
t0 = time.perf_counter(); y = vae.decode(zf)[0]; torch.cuda.synchronize(); td = time.perf_counter() - t0
xf = y.float().clamp(-1, 1)
vae.encode(xf); torch.cuda.synchronize()
t0 = time.perf_counter(); m = vae.encode(xf).latent_dist.parameters; torch.cuda.synchronize(); te = time.perf_counter() - t0
print(json.dumps({"tvae_decode_6x17f_256x448_s": round(td, 4), "frames_per_s_decode": round(V * y.shape[2] / td, 1),
                  "tvae_encode_6x17f_256x448_s": round(te, 4), "frames_per_s_encode": round(V * y.shape[2] / te, 1),
                  "out_shape": list(y.shape), "moments_shape": list(m.shape), "rel_decode_vs_oracle": rel,
                  "rel_encode_vs_oracle": rel_e, "finite": bool(torch.isfinite(y.float()).all() and torch.isfinite(m).all()),
                  "peak_GiB": round(torch.cuda.max_memory_allocated() / 2**30, 1)}))
