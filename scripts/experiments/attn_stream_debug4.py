"""debug aid: first launch of attn_stream_kernel with the fallback disabled (-DST_X_NO_FALLBACK) after another kernel used the LDS: garbage tells what it reads before writing"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from opendwm_amd import ops
from tests.test_hip_gpu import _rand
dev = torch.device("cuda:0")
bf16 = torch.bfloat16
for (I, N, Lc, heads, hs) in [(3, 448, 154, 24, 6), (150, 256, 40, 4, 2), (3, 448, 154, 24, 6)]:
    D = heads * 64
    qkv = _rand((I * N, 3 * D), dev, 21)
    cqkv = _rand((I * Lc, 3 * D), dev, 22) if Lc else None
    rm = ops.rowmap_identity(I, N)
    trace = torch.zeros(max(I * heads * (N + Lc), 8 * 4 * 64 * 8 * 2), dtype=torch.float32, device=dev)
    def run(variant):
        out = torch.full((I * N, D), float("nan"), dtype=bf16, device=dev)
        cout = torch.full((I * Lc, D), float("nan"), dtype=bf16, device=dev) if Lc else None
        kw = dict(q1=cqkv[:, :D], k1=cqkv[:, D:2 * D], v1=cqkv[:, 2 * D:], out1=cout) if Lc else {}
        if os.environ.get("ST_TRACE") and (variant >> 12) & 1:
            trace.zero_()
            kw["lse"] = trace
        ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out, rm, heads, variant=variant, **kw)
        torch.cuda.synchronize()
        if "lse" in kw:
            t = trace.view(torch.int64)[:8 * 4 * 64 * 8].view(8, 4, 64, 8)
            fb = (t[:, :, :, 6] == 0x7fffffff).nonzero()
            print("    units that took the fallback (workgroup < 8, wave, head index):", len(fb), fb[:12].tolist())
        return torch.cat([out.view(I, N, heads, 64), cout.view(I, Lc, heads, 64)], 1) if Lc else out.view(I, N, heads, 64)
    ref = run(hs << 8).float()                       # the 12-wave kernel (also leaves ITS images in the LDS)
    L = N + Lc
    for i in range(3):
        o = run((1 << 12) | (hs << 8)).float()
        d = (o - ref).abs().amax(3)                  # [I, L, heads]
        bad = ((d > 0.02) | torch.isnan(d)).nonzero()
        if i == 0:
            first = o
        print("    bit-equal to launch 0:", bool(torch.equal(o.nan_to_num(), first.nan_to_num())))
        print((I, N, Lc, heads, hs), "launch", i, "bad rows", len(bad), "nan", int(torch.isnan(o).sum()),
              "first (problem, row, head)", bad[:6].tolist(), "rows", sorted(set(bad[:, 1].tolist()))[:24], "heads", sorted(set(bad[:, 2].tolist()))[:24], flush=True)
