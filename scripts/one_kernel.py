"""Run one kernel configuration a few times (for rocprofv3 --pmc passes).
usage: python scripts/one_kernel.py attn_dual|attn_joint|gemm_ff2|gemm_out|gemm_geglu [variant]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from opendwm_amd import ops
dev = torch.device("cuda:0"); bf16 = torch.bfloat16
what = sys.argv[1]; var = int(sys.argv[2]) if len(sys.argv) > 2 else 0
def rnd(*s, scale=1.0): return (torch.randn(*s, device=dev) * scale).to(bf16)
H, D = 24, 1536
if what.startswith("attn"):
    I, N, Lc = 192, 448, 154
    qkv, cqkv = rnd(I * N, 3 * D), rnd(I * Lc, 3 * D)
    out, cout = torch.empty(I * N, D, device=dev, dtype=bf16), torch.empty(I * Lc, D, device=dev, dtype=bf16)
    kw = dict(q1=cqkv[:, :D], k1=cqkv[:, D:2 * D], v1=cqkv[:, 2 * D:], out1=cout) if what == "attn_joint" else {}
    f = lambda: ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out, ops.rowmap_identity(I, N), H, variant=var, **kw)
else:
    M, N, K = dict(gemm_ff2=(86016, 1536, 6144), gemm_out=(86016, 1536, 1536), gemm_geglu=(86016, 12288, 1536), gemm_ff1=(86016, 6144, 1536))[what]
    a, w, b = rnd(M, K), rnd(N, K, scale=K ** -0.5), rnd(N)
    f = lambda: ops.gemm(a, w, b)
for _ in range(3): f()
torch.cuda.synchronize()
