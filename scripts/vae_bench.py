"""Full-size SD 3.5 VAE decode timing (96 images of 256x448 px = one 6-view x 16-frame sample), synthetic weights.
Parity at this width is a test: tests/test_hip_gpu.py::test_vae_full_width_decode_vs_oracle_on_device."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                    # seeded synthetic weights
from opendwm_amd.vae import AutoencoderKL

dev, bf16 = torch.device("cuda:0"), torch.bfloat16
vae = AutoencoderKL(block_out_channels=(128, 256, 512, 512), layers_per_block=2, norm_num_groups=32, latent_channels=16).to(dev).to(bf16).eval()
bench.synth_init_(vae, 0)
z96 = torch.randn(96, 16, 32, 56, device=dev)
for chunk in (8, 12, 24):
    vae.decode(z96[:chunk], chunk=chunk)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    y = vae.decode(z96, chunk=chunk)[0]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"vae_decode_96img_256x448": round(dt, 4), "chunk": chunk, "img_per_s": round(96 / dt, 1),
                      "finite": bool(torch.isfinite(y.float()).all())}))
# roofline line: the convolutions (implicit GEMMs of dwm_gemm_bf16) timed with HIP events inside one decode, the rest of the
# decode (GroupNorm / SiLU / upsampling: HBM-bound passes over the activations) as the remainder
timer = bench.KernelTimer().install()
timer.enabled = True
torch.cuda.synchronize()
t0 = time.perf_counter()
vae.decode(z96, chunk=12)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
timer.enabled = False
g = timer.summary().get("gemm", {})
timer.uninstall()
print(json.dumps({"roofline": {"workload": "SD 3.5 VAE decode, 96 images of 256x448 (one 6-view x 16-frame sample), chunk 12",
                               "bound": "mfma", "kernel": "gemm_bf16_kernel (3x3 convolutions as implicit GEMMs, 1x1 convolutions, attention projections)",
                               "achieved": g.get("tflops"), "peak": bench.PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                               "frac": (g.get("tflops") or 0.0) / bench.PEAK_BF16_TFLOPS, "launches": g.get("launches"),
                               "gemm_ms": g.get("ms"), "decode_ms_with_event_overhead": 1e3 * dt,
                               "gemm_share_of_decode": (g.get("ms") or 0.0) / (1e3 * dt),
                               "whole_decode_mfma_frac": (g.get("flops") or 0.0) / dt / (bench.PEAK_BF16_TFLOPS * 1e12)}}))
