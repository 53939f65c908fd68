import os, sys, json, torch
sys.path.insert(0, "/root/repo")
from opendwm_amd import ops
from scripts.microbench import timeit, rnd
for name, M, N, K in [("geglu", 86016, 12288, 1536), ("ff2", 86016, 1536, 6144), ("sq8k", 8192, 8192, 8192)]:
    a, w, b = rnd(M, K), rnd(N, K, scale=K ** -0.5), rnd(N)
    fl = 2.0 * M * N * K
    r = {"case": name}
    for nm, dbg in (("noepi", 1), ("dma4B", 1 | 4)):
        r[nm] = round(fl / timeit(lambda: ops.gemm(a, w, b, _debug=dbg)) / 1e9)
    print(json.dumps(r), flush=True)
