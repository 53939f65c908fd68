"""debug aid: fast path of attn_stream_kernel with the fallback disabled (library built with -DST_X_NO_FALLBACK), error pattern per (problem, tile, head)"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from opendwm_amd import ops
from tests.test_hip_gpu import _rand
dev = torch.device("cuda:0")
for (I, N, heads, hs) in [(2, 448, 6, 1), (8, 448, 6, 6), (1, 288, 2, 1)]:
    D = heads * 64
    qkv = _rand((I * N, 3 * D), dev, 11, 1.0)
    rm = ops.rowmap_identity(I, N)
    ref = torch.full((I * N, D), float("nan"), dtype=torch.bfloat16, device=dev)
    ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], ref, rm, heads, variant=0)
    out = torch.full((I * N, D), float("nan"), dtype=torch.bfloat16, device=dev)
    ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out, rm, heads, variant=(1 << 12) | (hs << 8))
    torch.cuda.synchronize()
    o, r = out.float().view(I, N // 32, 32, heads, 64), ref.float().view(I, N // 32, 32, heads, 64)
    err = ((o - r).pow(2).sum((2, 4)) / r.pow(2).sum((2, 4)).clamp_min(1e-30)).sqrt()      # [I, tiles, heads]
    print((I, N, heads, hs), "nan", int(torch.isnan(o).sum()), "inf", int(torch.isinf(o).sum()), "absmax", float(o.nan_to_num(0, 0, 0).abs().max()))
    for p in range(min(I, 2)):
        for h in range(heads):
            print("  problem", p, "head", h, "tile errs", [round(float(e), 3) if e == e else "nan" for e in err[p, :, h]])
    # ratio pattern of the first bad tile
    bad = (err > 0.02).nonzero()
    if len(bad):
        p, t, h = bad[0].tolist()
        ro, rr = o[p, t, :, h], r[p, t, :, h]
        print("  first bad (p,t,h)", (p, t, h), "out row0[:6]", ro[0, :6].tolist(), "ref", rr[0, :6].tolist(), "ratio per query (median)", (ro / rr).median(1).values[:8].tolist())
