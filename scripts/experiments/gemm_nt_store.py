import os, sys, json, torch
sys.path.insert(0, "/root/repo")
from opendwm_amd import ops
from opendwm_amd.blocks import geglu_pack
from scripts.microbench import timeit, rnd
for name, M, N, K in [("out-proj", 86016, 1536, 1536), ("ff1", 86016, 6144, 1536), ("geglu", 86016, 12288, 1536), ("ff2", 86016, 1536, 6144)]:
    a, w, b = rnd(M, K), rnd(N, K, scale=K ** -0.5), rnd(N)
    res, gate = rnd(M, N), rnd(M // 448 + 1, N)
    fl = 2.0 * M * N * K
    r = {"case": name}
    for rep in range(2):
        for nm, dbg in (("", 0), ("_nt", 4)):
            r[f"plain{nm}{rep}"] = round(fl / timeit(lambda: ops.gemm(a, w, b, _debug=dbg)) / 1e9)
            r[f"resid{nm}{rep}"] = round(fl / timeit(lambda: ops.gemm(a, w, b, epilogue=ops.EPI_RESID, gate=gate, rows_per_gate=448, res=res, out=res, _debug=dbg)) / 1e9)
            r[f"geglu{nm}{rep}"] = round(fl / timeit(lambda: ops.gemm(a, w, b, epilogue=ops.EPI_GEGLU, _debug=dbg)) / 1e9)
    print(json.dumps(r), flush=True)
