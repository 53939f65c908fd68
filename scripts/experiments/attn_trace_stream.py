"""timeline of attn_stream_kernel's heads (library built with -DDWM_ATTN_TRACE for attention.hip and attention_stream.hip:
scripts/dev/build_variant.sh trace attention.hip,attention_stream.hip -DDWM_ATTN_TRACE; run with DWM_HIP_LIB=<that library>): per head, for the
4 waves of workgroups 0-7, shader-clock stamps (low 32 bits) at: 0 head top, 1 first key step done, 7 main loop done (the two or three last
steps follow), 2 tile loop over, 3 outputs normalised + stores issued, 4 vmcnt(0) passed, 5 behind the barrier(s)"""
import os, sys, statistics as st, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from opendwm_amd import ops
dev = torch.device("cuda:0"); bf16 = torch.bfloat16
H, D = 24, 1536
I, N, Lc = 192, 448, int(sys.argv[1]) if len(sys.argv) > 1 else 154
variant = int(sys.argv[2], 0) if len(sys.argv) > 2 else (1 << 12)
qkv = (torch.randn(I * N, 3 * D, device=dev)).to(bf16); cqkv = (torch.randn(max(I * Lc, 1), 3 * D, device=dev)).to(bf16)
out = torch.empty(I * N, D, device=dev, dtype=bf16); cout = torch.empty(max(I * Lc, 1), D, device=dev, dtype=bf16)
kw = dict(q1=cqkv[:, :D], k1=cqkv[:, D:2 * D], v1=cqkv[:, 2 * D:], out1=cout) if Lc else {}
L = N + Lc
trace = torch.zeros(I * H * L, dtype=torch.float32, device=dev)
rm = ops.rowmap_identity(I, N)
for _ in range(3):
    ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out, rm, H, variant=variant, **kw)
trace.zero_()
ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out, rm, H, lse=trace, variant=variant, **kw)
torch.cuda.synchronize()
t = trace.view(torch.int64)[:8 * 4 * 64 * 8].view(8, 4, 64, 8).cpu()
G = 18
order = (0, 1, 7, 2, 3, 4, 5)
for b in (0, 3):
    t0 = t[b, :, 0, 0].min().item()
    print(f"--- workgroup {b}: cycles relative to its first stamp, stamps in the order {order}; per head: waves 0 .. 3")
    for g in range(3):
        print(f"head {g:2d} | " + " | ".join(" ".join(f"{(t[b, w, g, s].item() - t0) if t[b, w, g, s].item() else -1:>7d}" for s in order) for w in range(4)))
    for w in range(4):
        dw = lambda a, bb: round(st.mean([((t[b, w, g, bb] - t[b, w, g, a]).item()) % (1 << 32) for g in range(2, G - 1)]))
        print(f"wave {w} mean cycles: head period", round(st.mean([((t[b, w, g + 1, 0] - t[b, w, g, 0]).item()) % (1 << 32) for g in range(2, G - 1)])),
              "| step 0", dw(0, 1), "| main loop", dw(1, 7), "| last steps", dw(7, 2), "| Q + normalise + stores", dw(2, 3),
              "| vmcnt(0)", dw(3, 4), "| barrier(s)", dw(4, 5), "| next head top", round(st.mean([((t[b, w, g + 1, 0] - t[b, w, g, 5]).item()) % (1 << 32) for g in range(2, G - 1)])))
