"""Build and run the 4-wave GEMM experiment (gemm4w.hip) against dwm_gemm_bf16 and the library GEMM behind torch.matmul.
usage (GPU box): python scripts/experiments/gemm4w/run.py        (the .so is built on the CPU box by `run.py --build` and travels)"""
import ctypes
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
SO = os.path.join(HERE, "libgemm4w.so")


def build():
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-gpu-rdc",
                           "-o", SO, os.path.join(HERE, "gemm4w.hip")])
    return SO


if __name__ == "__main__":
    if "--build" in sys.argv or not os.path.exists(SO):
        print(build())
        if "--build" in sys.argv:
            sys.exit(0)
    import torch
    sys.path.insert(0, ROOT)
    from opendwm_amd import ops
    from scripts.microbench import rnd, timeit
    lib = ctypes.CDLL(SO)
    lib.gemm4w_plain.restype = ctypes.c_int
    lib.gemm4w_plain.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                 ctypes.c_int, ctypes.c_void_p]

    def g4(a, w, out, buf):
        rc = lib.gemm4w_plain(a.data_ptr(), w.data_ptr(), out.data_ptr(), a.shape[0], w.shape[0], a.shape[1], buf,
                              torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
        return out

    print(torch.cuda.get_device_name(0))
    # correctness first (small shapes incl. K = 64 / 128 / 192: the three loop forms), then the bench shapes
    for M, N, K in [(256, 256, 64), (512, 256, 128), (256, 512, 192), (1024, 768, 1536), (2048, 512, 6144)]:
        a, w = rnd(M, K), rnd(N, K, scale=K ** -0.5)
        ref = a.float() @ w.float().T
        for buf in (0, 1, 2, 3):
            out = g4(a, w, torch.full((M, N), float("nan"), device=a.device, dtype=a.dtype), buf)
            torch.cuda.synchronize()
            err = ((out.float() - ref).norm() / ref.norm()).item()
            print(json.dumps({"check": [M, N, K], "buffer_form": buf, "rel_err": err, "finite": bool(torch.isfinite(out).all())}), flush=True)
    timeit(lambda: ops.gemm(rnd(8192, 8192), rnd(8192, 8192)), iters=30)
    for name, M, N, K in [("vt geglu in", 86016, 12288, 1536), ("ff1", 86016, 6144, 1536), ("out-proj", 86016, 1536, 1536),
                          ("ff2", 86016, 1536, 6144), ("8192^3", 8192, 8192, 8192)]:
        a, w = rnd(M, K), rnd(N, K, scale=K ** -0.5)
        out = torch.empty(M, N, device=a.device, dtype=a.dtype)
        fl = 2.0 * M * N * K
        res = {"case": name, "M": M, "N": N, "K": K}
        for rep in range(2):
            res.setdefault("dwm_8wave", []).append(round(fl / timeit(lambda: ops.gemm(a, w, None, out=out)) / 1e9, 1))
            res.setdefault("exp_4wave", []).append(round(fl / timeit(lambda: g4(a, w, out, 0)) / 1e9, 1))
            res.setdefault("exp_4wave_buffer_loads", []).append(round(fl / timeit(lambda: g4(a, w, out, 1)) / 1e9, 1))
            res.setdefault("exp_4wave_row_stores", []).append(round(fl / timeit(lambda: g4(a, w, out, 2)) / 1e9, 1))
            res.setdefault("exp_4wave_buffer_loads_row_stores", []).append(round(fl / timeit(lambda: g4(a, w, out, 3)) / 1e9, 1))
            res.setdefault("library", []).append(round(fl / timeit(lambda: torch.matmul(a, w.t(), out=out)) / 1e9, 1))
        print(json.dumps(res), flush=True)
