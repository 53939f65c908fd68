"""Full-size CogVideoX temporal VAE (THUDM/CogVideoX-2b widths, synthetic weights) on BASELINE configs[4] shapes:
6 views x 17 frames x 256x448 px <-> latents [6, 16, 5, 32, 56]: timing of decode and encode.  Parity at this width is
a test: tests/test_cogvideox_gpu.py::test_full_width_clip_vs_oracle_on_device."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                    # seeded synthetic weights
from opendwm_amd.vae_cogvideox import AutoencoderKLCogVideoX

dev, bf16 = torch.device("cuda:0"), torch.bfloat16
vae = AutoencoderKLCogVideoX().to(dev).to(bf16).eval()
bench.synth_init_(vae, 0)
g = torch.Generator().manual_seed(0)
V, T, H, W = 6, 5, 32, 56
zf = torch.randn(V, 16, T, H, W, generator=g).to(dev)
vae.decode(zf)
torch.cuda.synchronize()
t0 = time.perf_counter()
y = vae.decode(zf)[0]
torch.cuda.synchronize()
td = time.perf_counter() - t0
xf = y.float().clamp(-1, 1)
vae.encode(xf)
torch.cuda.synchronize()
t0 = time.perf_counter()
m = vae.encode(xf).latent_dist.parameters
torch.cuda.synchronize()
te = time.perf_counter() - t0
print(json.dumps({"tvae_decode_6x17f_256x448_s": round(td, 4), "frames_per_s_decode": round(V * y.shape[2] / td, 1),
                  "tvae_encode_6x17f_256x448_s": round(te, 4), "frames_per_s_encode": round(V * y.shape[2] / te, 1),
                  "out_shape": list(y.shape), "moments_shape": list(m.shape),
                  "finite": bool(torch.isfinite(y.float()).all() and torch.isfinite(m).all()),
                  "peak_GiB": round(torch.cuda.max_memory_allocated() / 2**30, 1)}))
