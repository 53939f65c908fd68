import os, sys, json, torch
sys.path.insert(0, "/root/repo")
from opendwm_amd import ops
from scripts.microbench import timeit, rnd
for name, M, N, K in [("out-proj", 86016, 1536, 1536), ("ff1", 86016, 6144, 1536), ("ff2", 86016, 1536, 6144)]:
    a, w, b = rnd(M, K), rnd(N, K, scale=K ** -0.5), rnd(N)
    res, gate = rnd(M, N), rnd(M // 448 + 1, N)
    fl = 2.0 * M * N * K
    r = {"case": name}
    for mult in (0, 1, 2, 4, 8):
        dbg = 0 if mult == 0 else (2048 | (mult << 12))
        r[f"resid_s{mult}"] = round(fl / timeit(lambda: ops.gemm(a, w, b, epilogue=ops.EPI_RESID, gate=gate, rows_per_gate=448, res=res, out=res, _debug=dbg)) / 1e9)
        r[f"gelu_s{mult}"] = round(fl / timeit(lambda: ops.gemm(a, w, b, act=ops.ACT_GELU_TANH, _debug=dbg)) / 1e9)
    print(json.dumps(r), flush=True)
