"""timeline of attn_res_kernel's heads (library built with DWM_EXTRA_FLAGS=-DDWM_ATTN_TRACE): per head, for the waves of
workgroups 0-7, shader-clock stamps at: 0 head top, 1 own copy landed, 2 after barrier A, 3 first unit done, 4 refill point
reached, 5 after its barrier, 6 all units done, 7 after the closing barrier"""
import os, sys, json, torch
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
t = trace.view(torch.int64)[:8 * 12 * 64 * 8].view(8, 12, 64, 8).cpu()
G = 18
for b in (0, 3):
    t0 = t[b, :, 0, 0].min().item()
    print(f"--- workgroup {b}: cycles relative to its first stamp; per head: wave 0 / wave 7 / wave 11")
    for g in range(G):
        rows = []
        for w in (0, 7, 11):
            rows.append(" ".join(f"{(t[b, w, g, s].item() - t0) if t[b, w, g, s].item() else -1:>8d}" for s in range(8)))
        if g < 4: print(f"head {g:2d} | " + " | ".join(rows))
    # summary over heads 2..G-2: mean durations
    d = lambda a, bb: [(t[b, 0, g, bb] - t[b, 0, g, a]).item() for g in range(2, G - 1)]
    import statistics as st
    for w in (0, 4, 8, 7, 11):
        dw = lambda a, bb: round(st.mean([(t[b, w, g, bb] - t[b, w, g, a]).item() for g in range(2, G - 1)]))
        print(f"wave {w:2d} mean cycles: head period", round(st.mean([(t[b, w, g + 1, 0] - t[b, w, g, 0]).item() for g in range(2, G - 1)])),
              "| own copy", dw(0, 1), "| barrier A", dw(1, 2), "| unit 0: to loop", dw(2, 4), "loop", dw(4, 5), "after loop", dw(5, 3),
              "| later rounds / idle", dw(3, 6), "| barrier C", dw(6, 7))
