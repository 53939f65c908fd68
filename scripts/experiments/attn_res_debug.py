"""per query tile / per head error of the resident attention kernel against the fp32 reference (debug aid)"""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from opendwm_amd import ops
from oracle import ctsd_oracle as O
dev = torch.device("cuda:0"); bf16 = torch.bfloat16
def run(I, N, heads, variant, seed=11):
    D = heads * 64
    g = torch.Generator().manual_seed(seed)
    qkv = torch.randn(I * N, 3 * D, generator=g).to(dev).to(bf16)
    out = torch.full((I * N, D), float("nan"), dtype=bf16, device=dev)
    ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out, ops.rowmap_identity(I, N), heads, variant=variant)
    f = qkv.float().view(I, N, 3, heads, 64)
    ref = O.sdpa(f[:, :, 0].transpose(1, 2), f[:, :, 1].transpose(1, 2), f[:, :, 2].transpose(1, 2)).transpose(1, 2)   # [I,N,heads,64]
    o = out.float().view(I, N, heads, 64)
    nt = (N + 31) // 32
    errs = []
    for t in range(nt):
        a, b = o[:, t * 32:(t + 1) * 32], ref[:, t * 32:(t + 1) * 32]
        errs.append(round(((a - b).norm() / b.norm()).item(), 4))
    perhead = [round(((o[:, :, h] - ref[:, :, h]).norm() / ref[:, :, h].norm()).item(), 4) for h in range(heads)]
    nan = int(torch.isnan(out.float()).sum())
    print(json.dumps(dict(I=I, N=N, heads=heads, variant=variant, nan=nan, per_tile=errs, per_head=perhead)), flush=True)
for N in (64, 96, 128, 160, 192, 256, 448, 575):
    for v in (1, 2):
        run(1, N, 2, v)
run(2, 448, 6, 1)
