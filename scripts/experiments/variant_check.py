"""GPU box: the experimental kernel variants of this round against the default kernels on the same inputs -
attention tile body (variant bits 6 / 7) and GEMM main-loop schedules (reserved bits 9-10) - max abs / rel difference."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from opendwm_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
bf16 = torch.bfloat16
torch.manual_seed(0)


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, device=dev) * scale).to(bf16)


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def attn_cases():
    H, D = 4, 256
    for I, N, Lc in ((3, 448, 154), (2, 100, 0), (2, 64, 10), (1, 300, 3), (5, 602, 0), (2, 33, 0), (2, 129, 0), (3, 192, 0), (2, 1792, 0)):
        qkv = rnd(I * N, 3 * D)
        cqkv = rnd(max(I * Lc, 1), 3 * D)
        kw = dict(q1=cqkv[:, :D], k1=cqkv[:, D:2 * D], v1=cqkv[:, 2 * D:]) if Lc else {}
        outs = {}
        for var in (0, 64):
            out = torch.zeros(I * N, D, device=dev, dtype=bf16)
            cout = torch.zeros(max(I * Lc, 1), D, device=dev, dtype=bf16)
            if Lc:
                kw["out1"] = cout
            ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out, ops.rowmap_identity(I, N), H, variant=var, **kw)
            torch.cuda.synchronize()
            outs[var] = (out.clone(), cout.clone())
        for var in (64,):
            print(json.dumps({"check": "attention", "I": I, "N": N, "Lc": Lc, "variant": var,
                              "rel_sample": rel(outs[var][0], outs[0][0]), "rel_ctx": rel(outs[var][1], outs[0][1]) if Lc else 0.0,
                              "finite": bool(torch.isfinite(outs[var][0].float()).all())}), flush=True)


def gemm_cases():
    for M, N, K, kind in ((896, 1536, 1536, "resid"), (462, 4608, 1536, "plain"), (1000, 6144, 1536, "gelu"), (700, 1536, 6144, "resid")):
        a, w, b = rnd(M, K), rnd(N, K, scale=K ** -0.5), rnd(N)
        gate, res_t = rnd(M // 448 + 1, N), rnd(M, N)
        outs = {}
        for dbg in (0, 1 << 9, 2 << 9, 3 << 9):
            out = torch.zeros(M, N, device=dev, dtype=bf16)
            if kind == "resid":
                ops.gemm(a, w, b, epilogue=ops.EPI_RESID, gate=gate, rows_per_gate=448, res=res_t, out=out, _debug=dbg)
            elif kind == "gelu":
                ops.gemm(a, w, b, act=ops.ACT_GELU_TANH, out=out, _debug=dbg)
            else:
                ops.gemm(a, w, b, out=out, _debug=dbg)
            torch.cuda.synchronize()
            outs[dbg] = out.clone()
        for dbg in (1 << 9, 2 << 9, 3 << 9):
            print(json.dumps({"check": "gemm", "M": M, "N": N, "K": K, "kind": kind, "sched": dbg >> 9,
                              "max_abs_diff": (outs[dbg].float() - outs[0].float()).abs().max().item()}), flush=True)


if __name__ == "__main__":
    attn_cases()
