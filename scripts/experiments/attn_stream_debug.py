"""debug aid: attn_stream_kernel (variant bit 12) against the fp32 reference and the 12-wave kernel, per (head, query tile), repeated"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from opendwm_amd import ops
from tests.test_hip_gpu import _attn_ref, _rand
from tests.common import rel_err
dev = torch.device("cuda:0")
lib = os.environ.get("DWM_HIP_LIB", "product")[-22:]
for (I, N, Lc, heads) in [(1, 288, 0, 2), (2, 448, 0, 6), (40, 448, 0, 12), (192, 448, 0, 24)]:
    D = heads * 64
    qkv = _rand((I * N, 3 * D), dev, 11, 1.0)
    rm = ops.rowmap_identity(I, N)
    out0 = torch.full((I * N, D), float("nan"), dtype=torch.bfloat16, device=dev)
    ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out0, rm, heads, variant=0)
    for rep in range(3):
        for var in (1 << 12, (1 << 12) | (1 << 15)):
            ref = out0
            if var >> 15:
                q2 = qkv.clone(); q2[:, :D] = (qkv[:, :D].float() * (0.125 * 1.4426950408889634)).to(torch.bfloat16)
            else:
                q2 = qkv
            out = torch.full((I * N, D), float("nan"), dtype=torch.bfloat16, device=dev)
            ops.attention(q2[:, :D], q2[:, D:2 * D], q2[:, 2 * D:], out, rm, heads, variant=var)
            torch.cuda.synchronize()
            o, r = out.float().view(I, N // 32, 32, heads, 64), ref.float().view(I, N // 32, 32, heads, 64)
            err = ((o - r).pow(2).sum((2, 4)) / r.pow(2).sum((2, 4)).clamp_min(1e-30)).sqrt()      # [I, tiles, heads]
            bad = (err > 0.02).nonzero()
            print(lib, (I, N, Lc, heads), hex(var), "rep", rep, "equal", bool(torch.equal(out, ref)), "max tile err", round(float(err.max()), 4),
                  "bad (problem, tile, head):", bad[:12].tolist(), "n_bad", len(bad), flush=True)
