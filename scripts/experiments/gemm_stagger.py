"""Start stagger of the GEMM's first workgroup per CU (reserved bits 12..23 = spread in us): are the epilogues an HBM
burst because the CUs run in lockstep?  usage (GPU box): python scripts/experiments/gemm_stagger.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from opendwm_amd import ops
from scripts.microbench import timeit, rnd

timeit(lambda: ops.gemm(rnd(8192, 8192), rnd(8192, 8192)), iters=50)
for name, M, N, K in [("out-proj", 86016, 1536, 1536), ("ff1", 86016, 6144, 1536), ("ff2", 86016, 1536, 6144), ("geglu", 86016, 12288, 1536)]:
    a, w, b = rnd(M, K), rnd(N, K, scale=K ** -0.5), rnd(N)
    res, gate = rnd(M, N), rnd(M // 448 + 1, N)
    fl = 2.0 * M * N * K
    r = {"case": name}
    for spread in (0, 10, 25, 50, 100, 0):
        dbg = spread << 12
        if name == "geglu":
            r[f"geglu_s{spread}"] = round(fl / timeit(lambda: ops.gemm(a, w, b, epilogue=ops.EPI_GEGLU, _debug=dbg)) / 1e9)
        else:
            r[f"resid_s{spread}"] = round(fl / timeit(lambda: ops.gemm(a, w, b, epilogue=ops.EPI_RESID, gate=gate, rows_per_gate=448, res=res, out=res, _debug=dbg)) / 1e9)
            r[f"gelu_s{spread}"] = round(fl / timeit(lambda: ops.gemm(a, w, b, act=ops.ACT_GELU_TANH, _debug=dbg)) / 1e9)
    print(json.dumps(r), flush=True)
