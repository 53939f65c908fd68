import os, sys, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from opendwm_amd import ops
from scripts.microbench import timeit, rnd
for name, M, N, K in [("out-proj", 86016, 1536, 1536), ("qkv", 86016, 4608, 1536), ("ff1", 86016, 6144, 1536), ("ff2", 86016, 1536, 6144), ("geglu", 86016, 12288, 1536), ("ctx", 29568, 4608, 1536), ("sq8k", 8192, 8192, 8192)]:
    a, w, b = rnd(M, K), rnd(N, K, scale=K ** -0.5), rnd(N)
    res, gate = rnd(M, N), rnd(M // 448 + 1, N)
    fl = 2.0 * M * N * K
    r = {"case": name}
    r["noepi"] = round(fl / timeit(lambda: ops.gemm(a, w, b, _debug=1)) / 1e9)
    for gm in (1, 2, 4, 16, 31):
        r[f"gm{gm}"] = round(fl / timeit(lambda: ops.gemm(a, w, b, _debug=gm << 4)) / 1e9)
    r["nostore"] = round(fl / timeit(lambda: ops.gemm(a, w, b, _debug=2)) / 1e9)
    r["storeonly"] = round(fl / timeit(lambda: ops.gemm(a, w, b, _debug=4)) / 1e9)
    r["plain"] = round(fl / timeit(lambda: ops.gemm(a, w, b)) / 1e9)
    r["gelu"] = round(fl / timeit(lambda: ops.gemm(a, w, b, act=ops.ACT_GELU_TANH)) / 1e9)
    r["resid"] = round(fl / timeit(lambda: ops.gemm(a, w, b, epilogue=ops.EPI_RESID, gate=gate, rows_per_gate=448, res=res, out=res)) / 1e9)
    if N % 64 == 0:
        r["geglu"] = round(fl / timeit(lambda: ops.gemm(a, w, b, epilogue=ops.EPI_GEGLU)) / 1e9)
    r["hipblaslt"] = round(fl / timeit(lambda: torch.matmul(a, w.t())) / 1e9)
    print(json.dumps(r), flush=True)
