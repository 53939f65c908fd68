"""timeline of attn_res4_kernel's heads (library built with DWM_EXTRA_FLAGS=-DDWM_ATTN_TRACE; run with DWM_ATTN_RES4=1 or 2): per head, for the 4 waves of workgroups
0-7, shader-clock stamps at: 0 head top, 1 own copy landed, 2 after barrier A, 4 tile loop starts, 5 tile loop ends, 3 unit done (stores
issued), 6 after the closing barrier, 7 next head's copy issued"""
import os, sys, statistics as st, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from opendwm_amd import ops
dev = torch.device("cuda:0"); bf16 = torch.bfloat16
H, D = 24, 1536
I, N, Lc = 192, 448, int(sys.argv[1]) if len(sys.argv) > 1 else 154
qkv = (torch.randn(I * N, 3 * D, device=dev)).to(bf16); cqkv = (torch.randn(max(I * Lc, 1), 3 * D, device=dev)).to(bf16)
out = torch.empty(I * N, D, device=dev, dtype=bf16); cout = torch.empty(max(I * Lc, 1), D, device=dev, dtype=bf16)
kw = dict(q1=cqkv[:, :D], k1=cqkv[:, D:2 * D], v1=cqkv[:, 2 * D:], out1=cout) if Lc else {}
L = N + Lc
trace = torch.zeros(I * H * L, dtype=torch.float32, device=dev)
rm = ops.rowmap_identity(I, N)
for _ in range(3):
    ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out, rm, H, **kw)
trace.zero_()
ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out, rm, H, lse=trace, **kw)
torch.cuda.synchronize()
t = trace.view(torch.int64)[:8 * 4 * 64 * 8].view(8, 4, 64, 8).cpu()
G = 18
order = (0, 1, 2, 4, 5, 3, 6, 7)
for b in (0, 3):
    t0 = t[b, :, 0, 0].min().item()
    print(f"--- workgroup {b}: cycles relative to its first stamp, stamps in the order {order}; per head: waves 0 .. 3")
    for g in range(4):
        print(f"head {g:2d} | " + " | ".join(" ".join(f"{(t[b, w, g, s].item() - t0) if t[b, w, g, s].item() else -1:>7d}" for s in order) for w in range(4)))
    for w in range(4):
        dw = lambda a, bb: round(st.mean([(t[b, w, g, bb] - t[b, w, g, a]).item() for g in range(2, G - 1)]))
        print(f"wave {w} mean cycles: head period", round(st.mean([(t[b, w, g + 1, 0] - t[b, w, g, 0]).item() for g in range(2, G - 1)])),
              "| own copy lands", dw(0, 1), "| barrier A", dw(1, 2), "| to loop", dw(2, 4), "| loop", dw(4, 5), "| stores", dw(5, 3),
              "| barrier C", dw(3, 6), "| copy issue", dw(6, 7))
