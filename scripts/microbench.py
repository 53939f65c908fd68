"""Per-kernel microbenchmarks on BASELINE config-3 shapes (runs on the GPU box).
usage: python scripts/microbench.py [attn] [gemm] [ln]"""
import json
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from opendwm_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
bf16 = torch.bfloat16


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, device=dev) * scale).to(bf16)


def bench_attn(variants, only=None):
    H, D = 24, 1536
    B, T, V, h, w = 2, 16, 6, 16, 28
    I, N, Lc = B * T * V, h * w, 154
    qkv = rnd(I * N, 3 * D)
    cqkv = rnd(I * Lc, 3 * D)
    out = torch.empty(I * N, D, device=dev, dtype=bf16)
    cout = torch.empty(I * Lc, D, device=dev, dtype=bf16)
    mask = torch.ones(B, V, V, dtype=torch.bool, device=dev)
    cases = {
        "joint L=602": (ops.rowmap_identity(I, N), dict(q1=cqkv[:, :D], k1=cqkv[:, D:2 * D], v1=cqkv[:, 2 * D:], out1=cout), 602),
        "dual L=448": (ops.rowmap_identity(I, N), {}, 448),
        "crossview L=168 mask": (ops.rowmap_crossview_rowwise(B, T, V, h, w), dict(group_mask=mask), 168),
        "temporal rowwise L=448": (ops.rowmap_temporal_rowwise(B, T, V, h, w), {}, 448),
        "temporal pointwise L=16": (ops.rowmap_temporal_pointwise(B, T, V, h, w), {}, 16),
    }
    rm0 = ops.rowmap_identity(I, N)
    timeit(lambda: ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out, rm0, H), iters=300)   # clocks up
    for name, (rm, kw, L) in cases.items():
        if only is not None and name not in only:
            continue
        fl = 4.0 * rm.n_problems * H * L * L * 64
        for var in list(variants) * 2:
            ms = timeit(lambda: ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out, rm, H, variant=var, **kw))
            print(json.dumps({"kernel": "attn", "case": name, "variant": var, "ms": round(ms, 4), "tflops": round(fl / ms / 1e9, 1)}), flush=True)


def bench_attn_full():
    """"full" temporal attention at the length the shipped UniMLVG example gives it (examples/ctsd_unimlvg_6views_video_generation.json:40,72:
    19 frames x 448 tokens = 8512 per (CFG half, view); crossview_temporal_dit.py:336-344): lands on the tiled kernel (128 / 256-query
    workgroups); never timed before round 5"""
    H, D = 24, 1536
    B, T, V, h, w = 2, 19, 6, 16, 28
    R = B * T * V * h * w
    qkv = rnd(R, 3 * D)
    out = torch.empty(R, D, device=dev, dtype=bf16)
    rm = ops.rowmap_temporal_full(B, T, V, h, w)
    L = T * h * w
    fl = 4.0 * rm.n_problems * H * L * L * 64
    for var in (1, 2, 0, 1, 2, 0):            # 1: 32 queries per wave, 2: 64, 0: automatic
        ms = timeit(lambda: ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out, rm, H, variant=var), iters=3, warm=1)
        print(json.dumps({"kernel": "attn", "case": f"temporal full L={L}", "variant": var, "ms": round(ms, 3), "tflops": round(fl / ms / 1e9, 1)}), flush=True)


def bench_attn_unet():
    """spatial self-attention of the SD 2.1 UNet's first level (configs[1]: 2 x 6 frames x 6 views images of 32 x 56 latent pixels, 5 heads
    of 64): L = 1792 on the tiled kernel, 32 (variant 1) against 64 (variant 2) queries per wave"""
    H, D = 5, 320
    I, L = 72, 1792
    qkv = rnd(I * L, 3 * D)
    out = torch.empty(I * L, D, device=dev, dtype=bf16)
    rm = ops.rowmap_identity(I, L)
    fl = 4.0 * I * H * L * L * 64
    for var in (1, 2, 1, 2):
        ms = timeit(lambda: ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out, rm, H, variant=var), iters=10, warm=2)
        print(json.dumps({"kernel": "attn", "case": f"unet spatial L={L}", "variant": var, "ms": round(ms, 3), "tflops": round(fl / ms / 1e9, 1)}), flush=True)


def bench_pointwise():
    """point-wise temporal attention (L = 16): packed small-L kernel (heads per wave 8 / 4 / 12 / 2) vs the tiled kernel"""
    H, D = 24, 1536
    B, T, V, h, w = 2, 16, 6, 16, 28
    R = B * T * V * h * w
    qkv = rnd(R, 3 * D)
    out = torch.empty(R, D, device=dev, dtype=bf16)
    rm = ops.rowmap_temporal_pointwise(B, T, V, h, w)
    gb = 4 * R * D * 2 / 1e9
    timeit(lambda: ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out, rm, H), iters=300)
    for name, var in (("packed hs=8", 0), ("packed hs=4", 4 << 8), ("packed hs=12", 12 << 8), ("packed hs=2", 2 << 8),
                      ("tiled", 32), ("packed hs=8", 0)):
        ms = timeit(lambda: ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out, rm, H, variant=var))
        print(json.dumps({"kernel": "attn-pointwise", "case": name, "ms": round(ms, 4), "GBps": round(gb / ms * 1e3, 1)}), flush=True)


def bench_crossview():
    """row-wise cross-view attention with the ring view mask (6 x 28 tokens per problem): group kernel (hs = heads per wave) vs
    the tiled kernel (variant bit 5); GB/s of algorithmic q / k / v / o bytes"""
    H, D = 24, 1536
    B, T, V, h, w = 2, 16, 6, 16, 28
    R = B * T * V * h * w
    qkv = rnd(R, 3 * D)
    out = torch.empty(R, D, device=dev, dtype=bf16)
    rm = ops.rowmap_crossview_rowwise(B, T, V, h, w)
    ring = torch.zeros(V, V, dtype=torch.bool)
    for i in range(V):
        for d in (-1, 0, 1):
            ring[i, (i + d) % V] = True
    mask = ring[None].repeat(B, 1, 1).to(dev)
    gb = 4 * R * D * 2 / 1e9
    timeit(lambda: ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out, rm, H, group_mask=mask), iters=300)
    for name, var in (("shared hs=8", 0), ("shared hs=8", 0), ("shared hs=4", 4 << 8), ("shared hs=12", 12 << 8), ("shared hs=6", 6 << 8),
                      ("shared hs=3", 3 << 8), ("per-wave hs=8", 128), ("tiled", 32), ("shared hs=8", 0)):
        ms = timeit(lambda: ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out, rm, H, group_mask=mask, variant=var))
        print(json.dumps({"kernel": "attn-crossview", "case": name, "ms": round(ms, 4), "GBps": round(gb / ms * 1e3, 1)}), flush=True)


def bench_transpose():
    from opendwm_amd import train_ops as T
    for rows, cols in ((86016, 1536), (86016, 6144), (29568, 1536)):
        x = rnd(rows, cols)
        ms = timeit(lambda: T.transpose(x))
        print(json.dumps({"kernel": "transpose", "rows": rows, "cols": cols, "ms": round(ms, 4),
                          "TB_per_s": round(4.0 * rows * cols / ms / 1e9, 2)}), flush=True)


def bench_gemm(dbg_list=(0, 1)):
    from opendwm_amd.blocks import geglu_pack
    shapes = [("qkv rmshead", 86016, 4608, 1536, "rms"), ("out-proj resid", 86016, 1536, 1536, "resid"),
              ("ff1 gelu", 86016, 6144, 1536, "gelu"), ("ff2 resid", 86016, 1536, 6144, "resid"),
              ("vt geglu", 86016, 12288, 1536, "geglu"), ("ctx qkv", 29568, 4608, 1536, "plain"),
              ("ctx embed", 29568, 1536, 4096, "plain"), ("adaln M=192", 192, 13824, 1536, "plain"),
              ("plain 8192^3", 8192, 8192, 8192, "plain")]
    for name, M, N, K, kind in shapes:
        a, w, b = rnd(M, K), rnd(N, K, scale=K ** -0.5), rnd(N)
        fl = 2.0 * M * N * K
        res = {}
        for di, dbg in enumerate([dbg_list[0]] + list(dbg_list)):     # the first variant is measured twice and its first pass dropped:
            # the first timed loop of a shape runs 10 % slower than any later one (same kernel: 958 vs 1082 TFLOP/s in r2g)
            if kind == "rms":
                rms = rnd(2 * 1536) * 0.1 + 1
                f = lambda: ops.gemm(a, w, b, epilogue=ops.EPI_RMSHEAD, rms_w=rms, rms_ncols=3072, rms_eps=1e-6, _debug=dbg)
            elif kind == "resid":
                gate, res_t = rnd(M // 448 + 1, N), rnd(M, N)
                f = lambda: ops.gemm(a, w, b, epilogue=ops.EPI_RESID, gate=gate, rows_per_gate=448, res=res_t, out=res_t, _debug=dbg)
            elif kind == "gelu":
                f = lambda: ops.gemm(a, w, b, act=ops.ACT_GELU_TANH, _debug=dbg)
            elif kind == "geglu":
                wp, bp = geglu_pack(w), geglu_pack(b)
                f = lambda: ops.gemm(a, wp, bp, epilogue=ops.EPI_GEGLU, _debug=dbg)
            else:
                f = lambda: ops.gemm(a, w, b, _debug=dbg)
            ms = timeit(f)
            if di == 0:
                continue
            res[{0: "w8", 1: "w8-noepi", 4: "w8-general-resid"}.get(dbg, str(dbg))] = round(fl / ms / 1e9, 1)
        ms_t = timeit(lambda: torch.matmul(a, w.t()))
        print(json.dumps({"kernel": "gemm", "case": name, "M": M, "N": N, "K": K, "tflops": res,
                          "hipblaslt_plain_tflops": round(fl / ms_t / 1e9, 1)}), flush=True)


def bench_gemm_tiles():
    """256 x 256 (one workgroup per CU) against 256 x 128 (two per CU) on the headline shapes and the UNet's N = 320 / 640"""
    from opendwm_amd.blocks import geglu_pack
    shapes = [("qkv rmshead", 86016, 4608, 1536, "rms"), ("out-proj resid", 86016, 1536, 1536, "resid"),
              ("ff1 gelu", 86016, 6144, 1536, "gelu"), ("ff2 resid", 86016, 1536, 6144, "resid"),
              ("vt geglu", 86016, 12288, 1536, "geglu"), ("ctx qkv", 29568, 4608, 1536, "rms"),
              ("adapter conv K=13824", 86016, 1536, 13824, "plain"),
              ("unet N=320 K=2880", 129024, 320, 2880, "plain"), ("unet N=320 K=320", 129024, 320, 320, "resid"),
              ("unet geglu N=2560 K=320", 129024, 2560, 320, "geglu"), ("unet N=640 K=5760", 32256, 640, 5760, "plain"),
              ("unet N=1280 K=1280", 8064, 1280, 1280, "plain")]
    for name, M, N, K, kind in shapes:
        a, w, b = rnd(M, K), rnd(N, K, scale=K ** -0.5), rnd(N)
        fl = 2.0 * M * N * K
        res = {}
        for di, tile in enumerate((1, 1, 2, 1, 2)):
            if kind == "rms":
                rms = rnd(2 * N // 3) * 0.1 + 1
                f = lambda: ops.gemm(a, w, b, epilogue=ops.EPI_RMSHEAD, rms_w=rms, rms_ncols=2 * N // 3, rms_eps=1e-6, tile=tile, split_k=1)
            elif kind == "resid":
                gate, res_t = rnd(M // 448 + 1, N), rnd(M, N)
                f = lambda: ops.gemm(a, w, b, epilogue=ops.EPI_RESID, gate=gate, rows_per_gate=448, res=res_t, out=res_t, tile=tile, split_k=1)
            elif kind == "gelu":
                f = lambda: ops.gemm(a, w, b, act=ops.ACT_GELU_TANH, tile=tile, split_k=1)
            elif kind == "geglu":
                wp, bp = geglu_pack(w), geglu_pack(b)
                f = lambda: ops.gemm(a, wp, bp, epilogue=ops.EPI_GEGLU, tile=tile, split_k=1)
            else:
                f = lambda: ops.gemm(a, w, b, tile=tile, split_k=1)
            ms = timeit(f)
            if di == 0:
                continue
            res.setdefault("256x256" if tile == 1 else "256x128", []).append(round(fl / ms / 1e9, 1))
        print(json.dumps({"kernel": "gemm_tiles", "case": name, "M": M, "N": N, "K": K, "tflops": res}), flush=True)


def bench_gemm_tiles_dev():
    """development library only (DWM_HIP_LIB = a -DDWM_DEV_HOOKS build): the 256 x 128 tile with one / two workgroups per CU
    and both tiles without their epilogue"""
    from opendwm_amd.blocks import geglu_pack
    names = {0: "256x256", 1: "256x256 no epilogue", 0x200: "256x128", 0x201: "256x128 no epilogue",
             0x600: "256x128 one workgroup per CU", 0x601: "256x128 one workgroup per CU, no epilogue"}
    for name, M, N, K, kind in [("vt geglu", 86016, 12288, 1536, "geglu"), ("ff2 resid", 86016, 1536, 6144, "resid"),
                                ("plain K=13824", 86016, 1536, 13824, "plain"), ("qkv plain", 86016, 4608, 1536, "plain")]:
        a, w, b = rnd(M, K), rnd(N, K, scale=K ** -0.5), rnd(N)
        fl = 2.0 * M * N * K
        res = {}
        for di, dbg in enumerate([0] + list(names)):
            if kind == "resid":
                gate, res_t = rnd(M // 448 + 1, N), rnd(M, N)
                f = lambda: ops.gemm(a, w, b, epilogue=ops.EPI_RESID, gate=gate, rows_per_gate=448, res=res_t, out=res_t, _debug=dbg, split_k=1)
            elif kind == "geglu":
                wp, bp = geglu_pack(w), geglu_pack(b)
                f = lambda: ops.gemm(a, wp, bp, epilogue=ops.EPI_GEGLU, _debug=dbg, split_k=1)
            else:
                f = lambda: ops.gemm(a, w, b, _debug=dbg, split_k=1)
            ms = timeit(f)
            if di:
                res[names[dbg]] = round(fl / ms / 1e9, 1)
        print(json.dumps({"kernel": "gemm_tiles_dev", "case": name, "M": M, "N": N, "K": K, "tflops": res}), flush=True)


def bench_gemm_epilogue_dev():
    """development library only: the 256 x 256 tile complete / without its output stores / without its epilogue"""
    from opendwm_amd.blocks import geglu_pack
    names = {0: "complete", 2: "no stores", 1: "no epilogue"}
    for name, M, N, K, kind in [("qkv rmshead", 86016, 4608, 1536, "rms"), ("ff1 gelu", 86016, 6144, 1536, "gelu"), ("vt geglu", 86016, 12288, 1536, "geglu"),
                                ("out-proj resid", 86016, 1536, 1536, "resid"), ("ff2 resid", 86016, 1536, 6144, "resid")]:
        a, w, b = rnd(M, K), rnd(N, K, scale=K ** -0.5), rnd(N)
        fl = 2.0 * M * N * K
        res = {}
        for di, dbg in enumerate([0] + list(names) + list(names)):
            if kind == "rms":
                rms = rnd(2 * N // 3) * 0.1 + 1
                f = lambda: ops.gemm(a, w, b, epilogue=ops.EPI_RMSHEAD, rms_w=rms, rms_ncols=2 * N // 3, rms_eps=1e-6, _debug=dbg, split_k=1)
            elif kind == "resid":
                gate, res_t = rnd(M // 448 + 1, N), rnd(M, N)
                f = lambda: ops.gemm(a, w, b, epilogue=ops.EPI_RESID, gate=gate, rows_per_gate=448, res=res_t, out=res_t, _debug=dbg, split_k=1)
            elif kind == "gelu":
                f = lambda: ops.gemm(a, w, b, act=ops.ACT_GELU_TANH, _debug=dbg, split_k=1)
            else:
                wp, bp = geglu_pack(w), geglu_pack(b)
                f = lambda: ops.gemm(a, wp, bp, epilogue=ops.EPI_GEGLU, _debug=dbg, split_k=1)
            ms = timeit(f)
            if di:
                res.setdefault(names[dbg], []).append(round(fl / ms / 1e9, 1))
        print(json.dumps({"kernel": "gemm_epilogue_dev", "case": name, "M": M, "N": N, "K": K, "tflops": res}), flush=True)


def bench_gemm_tn():
    """weight-gradient GEMM from the row-major operands (dwm_gemm_tn) against the path it replaces (two transposes + NT GEMM)"""
    from opendwm_amd import train_ops as T
    for name, M, N, Cc in [("dW out-proj", 86016, 1536, 1536), ("dW ff1", 86016, 6144, 1536), ("dW ff2", 86016, 1536, 6144),
                           ("dW geglu", 86016, 12288, 1536), ("dW ctx qkv", 29568, 4608, 1536)]:
        dy, x = rnd(M, N), rnd(M, Cc, scale=M ** -0.5)
        fl = 2.0 * M * N * Cc
        ms_tn = min(timeit(lambda: T.gemm_tn(dy, x)) for _ in range(2))
        ms_old = min(timeit(lambda: ops.gemm(T.transpose(dy), T.transpose(x), None)) for _ in range(2))
        dyt, xt = T.transpose(dy), T.transpose(x)
        ms_nt = min(timeit(lambda: ops.gemm(dyt, xt, None)) for _ in range(2))
        print(json.dumps({"kernel": "gemm_tn", "case": name, "M": M, "N": N, "C": Cc, "tn_tflops": round(fl / ms_tn / 1e9, 1),
                          "transposes_plus_nt_tflops": round(fl / ms_old / 1e9, 1), "nt_alone_tflops": round(fl / ms_nt / 1e9, 1)}), flush=True)


def bench_gemm_raster_dev():
    """development library only: group height of the tile rasterisation (reserved bits 4-8) per shape"""
    from opendwm_amd.blocks import geglu_pack
    for name, M, N, K, kind in [("qkv", 86016, 4608, 1536, "plain"), ("ff1", 86016, 6144, 1536, "plain"), ("vt geglu", 86016, 12288, 1536, "geglu"),
                                ("out-proj", 86016, 1536, 1536, "plain"), ("ff2", 86016, 1536, 6144, "plain")]:
        a, w, b = rnd(M, K), rnd(N, K, scale=K ** -0.5), rnd(N)
        fl = 2.0 * M * N * K
        res = {}
        for di, gm in enumerate([8, 2, 4, 8, 16, 24, 2, 4, 8, 16, 24]):
            dbg = gm << 4
            if kind == "geglu":
                wp, bp = geglu_pack(w), geglu_pack(b)
                f = lambda: ops.gemm(a, wp, bp, epilogue=ops.EPI_GEGLU, _debug=dbg, split_k=1)
            else:
                f = lambda: ops.gemm(a, w, b, _debug=dbg, split_k=1)
            ms = timeit(f)
            if di:
                res.setdefault(f"gm={gm}", []).append(round(fl / ms / 1e9, 1))
        print(json.dumps({"kernel": "gemm_raster_dev", "case": name, "M": M, "N": N, "K": K, "tflops": res}), flush=True)


def bench_stream32():
    """the residual-stream forms of round 4: RESID GEMM into a bf16 stream vs into the fp32 stream (fp32 residual in, fp32 out,
    no bf16 copy), LayerNorm from bf16 vs from fp32, at the hidden-state size of config 3"""
    M, D = 86016, 1536
    h16, h32 = rnd(M, D), torch.randn(M, D, device=dev)
    gate = rnd(192, D)
    for name, K in (("out-proj K=1536", 1536), ("ff2 K=6144", 6144)):
        a, w, b = rnd(M, K), rnd(D, K, scale=K ** -0.5), rnd(D)
        fl = 2.0 * M * D * K
        for rep in range(2):
            ms = timeit(lambda: ops.gemm(a, w, b, epilogue=ops.EPI_RESID, gate=gate, rows_per_gate=448, res=h16, out=h16))
            print(json.dumps({"kernel": "gemm-resid", "case": name, "stream": "bf16", "ms": round(ms, 4), "tflops": round(fl / ms / 1e9, 1)}), flush=True)
            ms = timeit(lambda: ops.gemm(a, w, b, epilogue=ops.EPI_RESID, gate=gate, rows_per_gate=448, res=h32, out32=h32, mirror=False))
            print(json.dumps({"kernel": "gemm-resid", "case": name, "stream": "fp32", "ms": round(ms, 4), "tflops": round(fl / ms / 1e9, 1)}), flush=True)
    mod = rnd(192, 9 * D)
    y, y2 = torch.empty_like(h16), torch.empty_like(h16)
    for x, nm in ((h16, "bf16"), (h32, "fp32")):
        ms = timeit(lambda: ops.layernorm(x, eps=1e-6, scale=mod[:, D:2 * D], shift=mod[:, :D], rows_per_mod=448, out=y, x32=x.dtype == torch.float32))
        print(json.dumps({"kernel": "ln-mod", "stream": nm, "ms": round(ms, 4), "GBps": round((x.numel() * x.element_size() + y.numel() * 2) / ms / 1e6, 1)}))
        ms = timeit(lambda: ops.layernorm(x, eps=1e-6, scale=mod[:, D:2 * D], shift=mod[:, :D], rows_per_mod=448, out=y,
                                          scale2=mod[:, 7 * D:8 * D], shift2=mod[:, 6 * D:7 * D], out2=y2, x32=x.dtype == torch.float32))
        print(json.dumps({"kernel": "ln-mod-dual", "stream": nm, "ms": round(ms, 4), "GBps": round((x.numel() * x.element_size() + 2 * y.numel() * 2) / ms / 1e6, 1)}))


def bench_ln32():
    """layernorm_kernel on the fp32 residual stream at the hidden-state size of the headline config (modulated, one and two outputs) and at
    the context stream's size; GB/s of the algorithmic bytes (fp32 row in, bf16 row(s) out)"""
    D = 1536
    mod = rnd(192, 9 * D)
    for M, rpm in ((86016, 448), (29568, 154)):
        x = torch.randn(M, D, device=dev)
        y, y2 = torch.empty(M, D, device=dev, dtype=torch.bfloat16), torch.empty(M, D, device=dev, dtype=torch.bfloat16)
        for rep in range(2):
            ms = timeit(lambda: ops.layernorm(x, eps=1e-6, scale=mod[:, D:2 * D], shift=mod[:, :D], rows_per_mod=rpm, out=y, x32=True), iters=20)
            print(json.dumps({"kernel": "ln-mod", "rows": M, "stream": "fp32", "ms": round(ms, 4), "GBps": round((x.numel() * 4 + y.numel() * 2) / ms / 1e6, 1)}))
            ms = timeit(lambda: ops.layernorm(x, eps=1e-6, scale=mod[:, D:2 * D], shift=mod[:, :D], rows_per_mod=rpm, out=y,
                                              scale2=mod[:, 7 * D:8 * D], shift2=mod[:, 6 * D:7 * D], out2=y2, x32=True), iters=20)
            print(json.dumps({"kernel": "ln-mod-dual", "rows": M, "stream": "fp32", "ms": round(ms, 4), "GBps": round((x.numel() * 4 + 2 * y.numel() * 2) / ms / 1e6, 1)}))


def bench_stream32_tiles():
    """the K = 1536 RESID launch on the fp32 stream (34 ms of the step at 852-877 TFLOP/s): 8-wave 256 x 256 (tile 1), two 4-wave
    workgroups per CU on 256 x 128 x 32 (tile 2), the 4-wave 256 x 256 kernel (tile 3) - does a second workgroup per CU overlap the
    epilogue's stream traffic with the other's main loop?"""
    M, D = 86016, 1536
    h32 = torch.randn(M, D, device=dev)
    gate = rnd(192, D)
    for name, K in (("out-proj K=1536", 1536), ("ff2 K=6144", 6144)):
        a, w, b = rnd(M, K), rnd(D, K, scale=K ** -0.5), rnd(D)
        fl = 2.0 * M * D * K
        for tile in [int(t) for t in os.environ.get("S32T_TILES", "3,1,2,3,1,2").split(",")]:
            def call():
                with ops.gemm_4wave_scope(tile == 3):
                    ops.gemm(a, w, b, epilogue=ops.EPI_RESID, gate=gate, rows_per_gate=448, res=h32, out32=h32, mirror=False, tile=0 if tile == 3 else tile)
            try:
                ms = timeit(call)
                print(json.dumps({"kernel": "gemm-resid-fp32", "case": name, "tile": tile, "ms": round(ms, 4), "tflops": round(fl / ms / 1e9, 1)}), flush=True)
            except RuntimeError as e:
                print(json.dumps({"kernel": "gemm-resid-fp32", "case": name, "tile": tile, "error": str(e)[:80]}), flush=True)


def bench_ln():
    x = rnd(86016, 1536)
    mod = rnd(192, 9 * 1536)
    y2 = torch.empty_like(x)
    D = 1536
    ms = timeit(lambda: ops.layernorm(x, eps=1e-6, scale=mod[:, D:2 * D], shift=mod[:, :D], rows_per_mod=448))
    print(json.dumps({"kernel": "ln-mod", "ms": round(ms, 4), "GBps": round(2 * x.numel() * 2 / ms / 1e6, 1)}))
    ms = timeit(lambda: ops.layernorm(x, eps=1e-6, scale=mod[:, D:2 * D], shift=mod[:, :D], rows_per_mod=448,
                                      scale2=mod[:, 7 * D:8 * D], shift2=mod[:, 6 * D:7 * D], out2=y2))
    print(json.dumps({"kernel": "ln-mod-dual", "ms": round(ms, 4), "GBps": round(3 * x.numel() * 2 / ms / 1e6, 1)}))


if __name__ == "__main__":
    what = sys.argv[1:] or ["cv", "attn", "gemm", "ln"]
    print(torch.cuda.get_device_name(0))
    if "tr" in what:
        bench_transpose()
    if "pw" in what:
        bench_pointwise()
    if "cv" in what:
        bench_crossview()
    if "attn" in what:
        bench_attn([0])
    if "attnx" in what:                  # resident kernel geometries (12 waves x 1 tile / 8 x 2; 6 / 3 / 2 heads per workgroup; online softmax) vs tiled
        bench_attn([0, 32 | 1])
    if "attnr" in what:                  # the resident kernel's launches only
        bench_attn([0], only=("joint L=602", "dual L=448", "temporal rowwise L=448"))
    if "attnr4" in what:                 # the one-wave-per-SIMD streaming form (default, 0) against the 12-wave kernel (bit 13); bit 15: Q arrives pre-scaled
        bench_attn([1 << 13, 0, 1 << 15, (1 << 13) | (1 << 15)], only=("joint L=602", "dual L=448", "temporal rowwise L=448"))
    if "attnfull" in what:
        bench_attn_full()
    if "attnunet" in what:
        bench_attn_unet()
    if "attnr4x" in what:                # diagnostics: forced online-softmax fallback (16) of both kernels
        bench_attn([0, 1 << 12, 16, 16 | (1 << 12)], only=("joint L=602", "dual L=448"))
    if "s32t" in what:
        bench_stream32_tiles()
    if "s32" in what:
        bench_stream32()
    if "ln32" in what:
        bench_ln32()
    if "gemm" in what:
        bench_gemm()
    if "gemmx" in what:                  # reserved bit 2: the general RESID epilogue instead of its FAST form; bit 0: no epilogue
        bench_gemm((0, 4, 1))
    if "gemmt" in what:
        bench_gemm_tiles()
    if "gemmtn" in what:
        bench_gemm_tn()
    if "gemmd" in what:
        bench_gemm_tiles_dev()
    if "gemme" in what:
        bench_gemm_epilogue_dev()
    if "gemmr" in what:
        bench_gemm_raster_dev()
    if "ln" in what:
        bench_ln()
