"""ONE full-size denoise step of BASELINE.json configs[2] on the HOST cores: the fp32 CPU oracle (oracle/ctsd_oracle.py, the
restated reference path) at 24 layers, d = 1536, latents [1,16,6,16,32,56] -> CFG batch 2, 154 text tokens, text+layout
model (ImageAdapter + point-wise temporal attention) or --text-only (row-wise temporal), CFG combine + Euler update
included.  Takes ~10-20 minutes and ~60 GB of RAM; run once per round, outside bench.py (whose `cpu_baseline` times a
bounded sample of the same workload), and commit the JSON it prints under profiles/.

    python scripts/cpu_full_step.py [--threads N] [--text-only] > profiles/rN_cpu_full_step.json
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--text-only", action="store_true")
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    import bench
    from oracle import ctsd_oracle as O
    from opendwm_amd.dit import model_flops
    layout = not args.text_only
    kwargs = bench.variant_kwargs(layout)
    cfg = O.make_config(**kwargs)
    gen = torch.Generator().manual_seed(0)
    t0 = time.perf_counter()
    sd = {n: O.synth_param(n, s, cfg, gen) for n, s in O.param_shapes(cfg).items()}
    w = bench.WORKLOAD
    inp = O.make_inputs(cfg, 2 * w["B"], w["T"], w["V"], w["H"], w["W"], seed=0, text_len=w["text_len"], n_time_ids=13 if layout else 11)
    lat = inp.pop("sample")[: w["B"]]
    inp.pop("timestep")
    if layout:
        inp["condition_image_tensor"] = torch.rand(2 * w["B"], w["T"], w["V"], 6, 8 * w["H"], 8 * w["W"], generator=gen)
    t_setup = time.perf_counter() - t0
    with torch.no_grad():
        t0 = time.perf_counter()
        out = O.denoise(sd, cfg, lat, inp, steps=w["inference_steps"], guidance_scale=w["guidance_scale"], stop=1)
        dt = time.perf_counter() - t0
    fl = model_flops(kwargs, 2 * w["B"], w["T"], w["V"], w["H"], w["W"], w["text_len"])
    flop = fl["total"] + (fl["adapter"] if layout else 0)
    print(json.dumps({
        "what": "one FULL-SIZE denoise step of BASELINE configs[2] on the host CPU: fp32 PyTorch oracle (restated reference path), "
                "model forward at the CFG batch + guidance + FlowMatch-Euler update",
        "variant": "text+layout (ImageAdapter + point-wise temporal)" if layout else "text only (row-wise temporal)",
        "seconds_per_step": dt, "denoise_steps_per_s": 1.0 / dt, "threads": args.threads, "host_cores": os.cpu_count(),
        "flop_per_step": flop, "tflops": flop / dt / 1e12, "setup_seconds": t_setup, "finite": bool(torch.isfinite(out).all()),
        "latents": list(lat.shape)}))


if __name__ == "__main__":
    main()
