"""Attention throughput vs sequence length at a fixed token count (fixed-cost-per-workgroup probe)."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from opendwm_amd import ops
from scripts.microbench import timeit, rnd, dev, bf16
H, D = 24, 1536
variants = [int(v) for v in sys.argv[1:]] or [1, 3]
for L in (128, 256, 448, 896, 1792, 3584):
    I = 86016 // L
    qkv = rnd(I * L, 3 * D)
    out = torch.empty(I * L, D, device=dev, dtype=bf16)
    rm = ops.rowmap_identity(I, L)
    fl = 4.0 * I * H * L * L * 64
    for var in variants:
        ms = timeit(lambda: ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out, rm, H, variant=var))
        print(json.dumps({"L": L, "I": I, "variant": var, "ms": round(ms, 4), "tflops": round(fl / ms / 1e9, 1)}), flush=True)
