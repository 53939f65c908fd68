"""debug aid: are first launches bit-identical to later ones - for the streaming kernel AND for the 12-wave resident kernel?"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from opendwm_amd import ops
from tests.test_hip_gpu import _rand
dev = torch.device("cuda:0")
bf16 = torch.bfloat16
for rep in range(3):
  for (I, N, Lc, heads, hs) in [(150, 256, 40, 4, 2), (3, 448, 154, 24, 6)]:
    D = heads * 64
    qkv = _rand((I * N, 3 * D), dev, 21 + rep)
    cqkv = _rand((I * Lc, 3 * D), dev, 22 + rep) if Lc else None
    rm = ops.rowmap_identity(I, N)
    def run(variant):
        out = torch.full((I * N, D), float("nan"), dtype=bf16, device=dev)
        cout = torch.full((I * Lc, D), float("nan"), dtype=bf16, device=dev) if Lc else None
        kw = dict(q1=cqkv[:, :D], k1=cqkv[:, D:2 * D], v1=cqkv[:, 2 * D:], out1=cout) if Lc else {}
        ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out, rm, heads, variant=variant, **kw)
        torch.cuda.synchronize()
        return torch.cat([out.view(I, N, heads, 64), cout.view(I, Lc, heads, 64)], 1) if Lc else out.view(I, N, heads, 64)
    for name, var in (("12-wave", hs << 8), ("stream", (1 << 12) | (hs << 8)), ("12-wave", hs << 8), ("stream", (1 << 12) | (hs << 8))):
        rs = [run(var) for _ in range(4)]
        print(rep, (I, N, Lc, heads, hs), name, "launches 1..3 bit-equal to launch 0:", [bool(torch.equal(r, rs[0])) for r in rs[1:]],
              "launch 0 vs 1: differing elements", int((rs[0] != rs[1]).sum()), "max abs", float((rs[0].float() - rs[1].float()).abs().max()), flush=True)
