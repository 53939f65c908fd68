import os, sys, json, torch
sys.path.insert(0, "/root/repo")
from opendwm_amd import ops
from scripts.microbench import timeit, rnd
for name, M, N, K in [("ff2", 86016, 1536, 6144), ("geglu", 86016, 12288, 1536), ("sq8k", 8192, 8192, 8192)]:
    a, w, b = rnd(M, K), rnd(N, K, scale=K ** -0.5), rnd(N)
    fl = 2.0 * M * N * K
    r = {"case": name}
    for nm, dbg in (("noepi", 1), ("skipA", 1 | 4), ("skipW", 1 | 8), ("skipAW", 1 | 12), ("skipDMA", 1 | 512), ("hotDMA", 1 | 1024), ("skipAll", 1 | 12 | 512)):
        r[nm] = round(fl / timeit(lambda: ops.gemm(a, w, b, _debug=dbg)) / 1e9)
    print(json.dumps(r), flush=True)
