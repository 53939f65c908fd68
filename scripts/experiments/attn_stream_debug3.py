"""debug aid: repeatability of attn_stream_kernel on two-segment, several-heads-per-item problems; where two runs differ"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from opendwm_amd import ops
from tests.test_hip_gpu import _rand
dev = torch.device("cuda:0")
bf16 = torch.bfloat16
for (I, N, Lc, heads, hs) in [(150, 256, 40, 4, 2), (3, 448, 154, 24, 6), (150, 256, 40, 4, 2), (3, 448, 154, 24, 6)]:
    D = heads * 64
    qkv = _rand((I * N, 3 * D), dev, 21)
    cqkv = _rand((I * Lc, 3 * D), dev, 22) if Lc else None
    rm = ops.rowmap_identity(I, N)
    def run(variant):
        out = torch.full((I * N, D), float("nan"), dtype=bf16, device=dev)
        cout = torch.full((I * Lc, D), float("nan"), dtype=bf16, device=dev) if Lc else None
        kw = dict(q1=cqkv[:, :D], k1=cqkv[:, D:2 * D], v1=cqkv[:, 2 * D:], out1=cout) if Lc else {}
        ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out, rm, heads, variant=variant, **kw)
        torch.cuda.synchronize()
        return out, cout
    ref = run(hs << 8)
    runs = [run((1 << 12) | (hs << 8)) for _ in range(8)]
    print((I, N, Lc, heads, hs), "seg1 delta (elements)", (cqkv.data_ptr() - qkv.data_ptr()) // 2 if Lc else None)
    for i, (o, co) in enumerate(runs):
        d0 = (o.float() - ref[0].float()).abs().view(I, N, heads, 64).amax(3)           # [I, N, heads]
        bad0 = (d0 > 0.05).nonzero()
        msg = f"  run {i}: seg0 bad (problem, row, head) n={len(bad0)} first {bad0[:4].tolist()}"
        if bad0.numel():
            msg += f" heads {sorted(set(bad0[:, 2].tolist()))[:12]} rows/32 {sorted(set((bad0[:, 1] // 32).tolist()))[:20]}"
        if Lc:
            d1 = (co.float() - ref[1].float()).abs().view(I, Lc, heads, 64).amax(3)
            bad1 = (d1 > 0.05).nonzero()
            msg += f" | seg1 bad n={len(bad1)} first {bad1[:4].tolist()}"
            if bad1.numel():
                msg += f" heads {sorted(set(bad1[:, 2].tolist()))[:12]}"
        print(msg, "nan", int(torch.isnan(o.float()).sum()), flush=True)
        if i > 0:
            ne0 = (o != runs[0][0]).view(I, N // 32, 32, heads, 64).any(4).any(2).nonzero()          # (problem, tile, head)
            print(f"      vs run 0: seg0 differing (problem, tile, head) n={len(ne0)} {ne0[:10].tolist()} max abs {float((o.float() - runs[0][0].float()).abs().max()):.4g}", end="")
            if Lc:
                ne1 = (co != runs[0][1]).view(I, Lc, heads, 64).any(3).nonzero()
                print(f" | seg1 differing (problem, row, head) n={len(ne1)} {ne1[:10].tolist()} max abs {float((co.float() - runs[0][1].float()).abs().max()):.4g}", end="")
                if len(ne1):
                    pb, rw, hd = ne1[0].tolist()
                    print(f" e.g. run0 {runs[0][1].view(I, Lc, heads, 64)[pb, rw, hd, :4].tolist()} this {co.view(I, Lc, heads, 64)[pb, rw, hd, :4].tolist()} ref {ref[1].view(I, Lc, heads, 64)[pb, rw, hd, :4].tolist()}", end="")
            print(flush=True)
