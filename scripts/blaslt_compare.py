"""Library GEMM (what torch.matmul / F.linear dispatch to on ROCm: hipBLASLt / rocBLAS) against dwm_gemm_bf16 on the bench's GEMM
shapes, plain and with a bias: how much room a hand-tuned library main loop still has over this repo's.  A comparison probe only -
nothing under opendwm_amd/ calls a library GEMM.   usage (GPU box): python scripts/blaslt_compare.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opendwm_amd import ops                                   # noqa: E402
from scripts.microbench import rnd, timeit                    # noqa: E402

SHAPES = [("vt geglu in", 86016, 12288, 1536), ("ff1", 86016, 6144, 1536), ("qkv", 86016, 4608, 1536), ("out-proj", 86016, 1536, 1536),
          ("ff2", 86016, 1536, 6144), ("ctx ff1", 29568, 6144, 1536), ("adapter conv as GEMM", 86016, 1536, 13824), ("8192^3", 8192, 8192, 8192)]

if __name__ == "__main__":
    print(torch.cuda.get_device_name(0), torch.__version__)
    timeit(lambda: ops.gemm(rnd(8192, 8192), rnd(8192, 8192)), iters=30)              # clocks up
    for name, M, N, K in SHAPES:
        a, w, b = rnd(M, K), rnd(N, K, scale=K ** -0.5), rnd(N)
        out = torch.empty(M, N, device=a.device, dtype=a.dtype)
        fl = 2.0 * M * N * K
        res = {"case": name, "M": M, "N": N, "K": K}
        for rep in range(2):
            res.setdefault("dwm_plain", []).append(round(fl / timeit(lambda: ops.gemm(a, w, None, out=out)) / 1e9, 1))
            res.setdefault("dwm_bias", []).append(round(fl / timeit(lambda: ops.gemm(a, w, b, out=out)) / 1e9, 1))
            res.setdefault("torch_matmul", []).append(round(fl / timeit(lambda: torch.matmul(a, w.t(), out=out)) / 1e9, 1))
            res.setdefault("torch_linear_bias", []).append(round(fl / timeit(lambda: torch.nn.functional.linear(a, w, b)) / 1e9, 1))
        print(json.dumps(res), flush=True)
