"""Timeline of gemm4w_kernel's epilogue on the N = K = 1536 RESID launch of the fp32 stream (timing build with -DDWM_G4_TRACE:
scripts/experiments/gemm4w_resid_prefetch/trace.patch; thread 0 of every workgroup stamps s_memtime into the GEMM workspace).
usage: DWM_HIP_LIB=.../libdwm_hip_trc.so python scripts/experiments/gemm4w_trace.py [K]"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from opendwm_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
M, D = 86016, 1536
K = int(sys.argv[1]) if len(sys.argv) > 1 else 1536
bf = torch.bfloat16
a = (torch.randn(M, K, device=dev)).to(bf)
w = (torch.randn(D, K, device=dev) * K ** -0.5).to(bf)
b = torch.randn(D, device=dev).to(bf)
gate = torch.randn(192, D, device=dev).to(bf)
h32 = torch.randn(M, D, device=dev)
ws = ops._gemm_workspace(dev)
os.environ["DWM_G4_TRACE_PTR"] = str(ws.data_ptr())
for rep in range(3):
    ws.zero_()
    with ops.gemm_4wave_scope(True):
        ops.gemm(a, w, b, epilogue=ops.EPI_RESID, gate=gate, rows_per_gate=448, res=h32, out32=h32, mirror=False)
    torch.cuda.synchronize()
ntiles = (M // 256) * (D // 256)
t = ws.view(torch.int64)[: ntiles * 16].view(ntiles, 16).cpu().double()
t0 = t[:, 0:1]
names = ["start", "loop_end", "epi_ready"] + [f"pass{i}" for i in range(8)] + ["drained"]
d = t[:, :12] - t0
print("tiles", ntiles, "100 MHz ticks = 10 ns; per-workgroup means (us) since its start, and the step from the previous stamp")
prev = 0.0
for k, nm in enumerate(names):
    m = d[:, k].mean().item() / 100.0
    print(f"  {nm:10s} {m:8.2f} us   (+{m - prev:6.2f})   p10 {d[:, k].quantile(0.1).item() / 100:.2f} p90 {d[:, k].quantile(0.9).item() / 100:.2f}")
    prev = m
first = t[:, 0].min()
# how many workgroups are inside their epilogue at a time (1 us bins): bunched rounds show as peaks near the CU count
span = int((t[:, 11].max() - first).item() // 100) + 1
conc = torch.zeros(span + 1)
for a0, b0 in zip(((t[:, 2] - first) / 100).long().tolist(), ((t[:, 11] - first) / 100).long().tolist()):
    conc[a0:b0 + 1] += 1
print("workgroups in their epilogue per 1-us bin: max", int(conc.max()), "mean", round(conc.mean().item(), 1),
      "share of time with > 160:", round((conc > 160).float().mean().item(), 3), "with < 40:", round((conc < 40).float().mean().item(), 3))
print("  every 10 us:", [int(x) for x in conc[::10].tolist()][:64])
wait = (t[:, 2] - t[:, 1]).mean().item() / 100
print("  mean wait at the admission point (loop_end -> epi_ready)", round(wait, 2), "us")
print("kernel span (first start -> last drained)", (t[:, 11].max() - first).item() / 100.0, "us; starts of the first 256 / all tiles: p50",
      ((t[:256, 0] - first).median().item()) / 100.0, ((t[:, 0] - first).median().item()) / 100.0)
