"""Library GEMM (what torch.matmul / F.linear dispatch to on ROCm: hipBLASLt / rocBLAS) against the kernels the bench SHIPS, shape by
shape of the denoise step (profiles/r5k_gemm_shapes.jsonl): the 4-wave kernels of gemm_bf16_4w.hip WITH the fused epilogue the step
runs on that shape (GEGLU, q / k RMSNorm, GELU, gate + residual on the fp32 stream), next to the same kernel with a plain epilogue
and next to the library's plain GEMM and GEMM + bias of the same M, N, K.  The library has no counterpart of the fused epilogues: its
number is the bare product, i.e. what the fused launch would have to add a second pass to.  A comparison probe only - nothing
under opendwm_amd/ calls a library GEMM.   usage (GPU box): python scripts/blaslt_compare.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opendwm_amd import ops                                   # noqa: E402
from scripts.microbench import dev, rnd, timeit               # noqa: E402

# (name, M, N, K, fused epilogue of the step on this shape)
SHAPES = [("vt geglu in", 86016, 12288, 1536, "geglu"), ("ff2 + gate + fp32 stream", 86016, 1536, 6144, "resid32"),
          ("qkv + rms", 86016, 4608, 1536, "rms"), ("out-proj + gate + fp32 stream", 86016, 1536, 1536, "resid32"),
          ("ff1 + gelu", 86016, 6144, 1536, "gelu"), ("ctx ff1 + gelu", 29568, 6144, 1536, "gelu"),
          ("ctx ff2 + gate + fp32 stream", 29568, 1536, 6144, "resid32"), ("ctx qkv + rms", 29568, 4608, 1536, "rms"),
          ("ctx out-proj + gate + fp32 stream", 29568, 1536, 1536, "resid32"), ("8192^3", 8192, 8192, 8192, "plain")]


def fused_call(kind, a, w, b, M, N, K):
    if kind == "geglu":
        out = torch.empty(M, N // 2, device=dev, dtype=a.dtype)
        return lambda: ops.gemm(a, w, b, epilogue=ops.EPI_GEGLU, out=out)
    if kind == "rms":
        out = torch.empty(M, N, device=dev, dtype=a.dtype)
        rw = rnd(N)
        return lambda: ops.gemm(a, w, b, epilogue=ops.EPI_RMSHEAD, rms_w=rw, rms_ncols=2 * N // 3, out=out)
    if kind == "gelu":
        out = torch.empty(M, N, device=dev, dtype=a.dtype)
        return lambda: ops.gemm(a, w, b, act=ops.ACT_GELU_TANH, out=out)
    if kind == "resid32":
        h32 = torch.randn(M, N, device=dev)
        rpg = 448 if M == 86016 else 154
        gate = rnd(M // rpg, N)
        return lambda: ops.gemm(a, w, b, epilogue=ops.EPI_RESID, gate=gate, rows_per_gate=rpg, res=h32, out32=h32, mirror=False)
    out = torch.empty(M, N, device=dev, dtype=a.dtype)
    return lambda: ops.gemm(a, w, b, out=out)


if __name__ == "__main__":
    print(torch.cuda.get_device_name(0), torch.__version__)
    with ops.gemm_4wave_scope(True):
        timeit(lambda: ops.gemm(rnd(8192, 8192), rnd(8192, 8192)), iters=30)              # clocks up
    for name, M, N, K, kind in SHAPES:
        a, w, b = rnd(M, K), rnd(N, K, scale=K ** -0.5), rnd(N)
        out = torch.empty(M, N, device=dev, dtype=a.dtype)
        fl = 2.0 * M * N * K
        res = {"case": name, "M": M, "N": N, "K": K, "fused": kind}
        fused = fused_call(kind, a, w, b, M, N, K)
        for rep in range(2):
            with ops.gemm_4wave_scope(True):
                n0 = ops._lib.load().dwm_gemm4w_launches()
                res.setdefault("dwm4w_fused", []).append(round(fl / timeit(fused) / 1e9, 1))
                res.setdefault("dwm4w_plain_bias", []).append(round(fl / timeit(lambda: ops.gemm(a, w, b, out=out)) / 1e9, 1))
                res["served_by_4wave"] = ops._lib.load().dwm_gemm4w_launches() > n0
            res.setdefault("lib_matmul", []).append(round(fl / timeit(lambda: torch.matmul(a, w.t(), out=out)) / 1e9, 1))
            res.setdefault("lib_linear_bias", []).append(round(fl / timeit(lambda: torch.nn.functional.linear(a, w, b)) / 1e9, 1))
        best = lambda k: max(res[k])                                                           # noqa: E731
        res["fused_vs_lib_bias"] = round(best("dwm4w_fused") / best("lib_linear_bias"), 3)
        res["plain_vs_lib_bias"] = round(best("dwm4w_plain_bias") / best("lib_linear_bias"), 3)
        print(json.dumps(res), flush=True)
        del a, w, b, out, fused
        torch.cuda.empty_cache()
