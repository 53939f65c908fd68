"""GPU leg: the 4-wave GEMM kernels (opendwm_amd/csrc/gemm_bf16_4w.hip; what the MMDiT inference forward asks for with
dwm_gemm_args.tile = 3, and what DWM_GEMM4W=1 turns on for every call).

The switch is read once per process by dwm_gemm_bf16, so the battery runs in a subprocess with the variable set: every epilogue
the 4-wave kernels cover (bias / activations, GEGLU, q-k RMSNorm heads, the gated / plain / blended residual on the bf16 and on the
fp32 stream, in place) on shapes they accept, against fp64 matrix products of the same bf16 inputs, plus the fallback for shapes
they do not accept; `dwm_gemm4w_launches` must count exactly the covered calls.  Direct ops.gemm calls elsewhere in tests/ (variable
unset, tile 0) run the 8-wave kernels; the full-size model tests run the mix the inference forward produces."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
TOL = 4e-3                  # one bf16 rounding of fp32-accumulated results
TOL32 = 2e-5                # fp32 stream outputs


def _battery():
    sys.path.insert(0, ROOT)
    from opendwm_amd import _lib, ops
    from opendwm_amd.blocks import geglu_pack
    bf16, f32 = torch.bfloat16, torch.float32
    dev = torch.device("cuda:0")
    lib = _lib.load()
    out = {}

    def rnd(shape, seed, scale=1.0, dtype=bf16):
        g = torch.Generator().manual_seed(seed)
        return (torch.randn(*shape, generator=g) * scale).to(dev).to(dtype)

    def rel(a, b):
        a, b = a.double(), b.double()
        return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()

    def launches():
        return int(lib.dwm_gemm4w_launches())

    n0 = launches()
    # PLAIN: bias / no bias, activations
    for M, N, K in [(256, 256, 128), (512, 768, 192), (1024, 1536, 1536), (2048, 512, 6144)]:
        a, w, b = rnd((M, K), 1), rnd((N, K), 2, K ** -0.5), rnd((N,), 3)
        y = a.double() @ w.double().T
        # (split_k=1: small tile grids with a long K would otherwise take dwm_gemm_bf16's split-K kernels, which stay 8-wave)
        errs = {"none": rel(ops.gemm(a, w, None, split_k=1), y), "bias": rel(ops.gemm(a, w, b, split_k=1), y + b.double()),
                "gelu_tanh": rel(ops.gemm(a, w, b, act=ops.ACT_GELU_TANH, split_k=1), torch.nn.functional.gelu(y + b.double(), approximate="tanh")),
                "silu": rel(ops.gemm(a, w, b, act=ops.ACT_SILU, split_k=1), torch.nn.functional.silu(y + b.double())),
                "relu": rel(ops.gemm(a, w, b, act=ops.ACT_RELU, split_k=1), torch.relu(y + b.double()))}
        out[f"plain_{M}x{N}x{K}"] = errs
    out["launches_plain"] = launches() - n0
    n0 = launches()
    # GEGLU (weights packed in 64-row groups [32 value | 32 gate])
    for M, N, K in [(256, 256, 128), (768, 12288, 1536)]:
        a, w, b = rnd((M, K), 4), rnd((N, K), 5, K ** -0.5), rnd((N,), 6)
        y = a.double() @ w.double().T + b.double()
        ref = y[:, :N // 2] * torch.nn.functional.gelu(y[:, N // 2:])
        got = ops.gemm(a, geglu_pack(w), geglu_pack(b), epilogue=ops.EPI_GEGLU)
        out[f"geglu_{M}x{N}x{K}"] = rel(got, ref)
    out["launches_geglu"] = launches() - n0
    n0 = launches()
    # q / k RMSNorm per 64-column head on the first 2/3 of the columns (fused qkv projection)
    for M, heads, K, biased in [(512, 24, 1536, True), (256, 4, 128, False)]:
        D = heads * 64
        a, w = rnd((M, K), 7), rnd((3 * D, K), 8, K ** -0.5)
        b = rnd((3 * D,), 9) if biased else None
        rms = (1.0 + 0.1 * torch.randn(2 * D, generator=torch.Generator().manual_seed(10))).to(dev).to(bf16)
        y = a.double() @ w.double().T + (b.double() if biased else 0.0)
        qk = y[:, :2 * D].view(M, 2 * heads, 64)
        qk = qk * torch.rsqrt(qk.pow(2).mean(-1, keepdim=True) + 1e-6) * rms.double().view(2 * heads, 64)
        ref = torch.cat([qk.reshape(M, 2 * D), y[:, 2 * D:]], 1)
        got = ops.gemm(a, w, b, epilogue=ops.EPI_RMSHEAD, rms_w=rms, rms_ncols=2 * D, rms_eps=1e-6)
        out[f"rmshead_{M}x{heads}x{K}"] = rel(got, ref)
    out["launches_rmshead"] = launches() - n0
    n0 = launches()
    # RESID on a bf16 stream and on the fp32 stream: gate + residual, residual + blend, residual; in place
    for M, N, K, rpg in [(1792, 1536, 1536, 448), (512, 256, 128, 128), (768, 1536, 6144, 256)]:
        a, w, b = rnd((M, K), 11), rnd((N, K), 12, K ** -0.5), rnd((N,), 13)
        groups = (M + rpg - 1) // rpg
        gate = rnd((groups, N), 14)
        alpha = torch.rand(groups, generator=torch.Generator().manual_seed(15)).to(dev)
        rows = torch.arange(M, device=dev) // rpg
        y = a.double() @ w.double().T + b.double()
        al = alpha.double()[rows][:, None]
        for name, dt, tol in (("bf16", bf16, TOL), ("fp32", f32, TOL32)):
            res, blend = rnd((M, N), 16, dtype=dt), rnd((M, N), 17, dtype=dt)
            kw = (lambda t: dict(out32=t, mirror=False, split_k=1)) if dt == f32 else (lambda t: dict(out=t, split_k=1))
            r1 = res.clone()
            ops.gemm(a, w, b, epilogue=ops.EPI_RESID, gate=gate, rows_per_gate=rpg, res=r1, **kw(r1))
            bl = blend.clone()
            ops.gemm(a, w, b, epilogue=ops.EPI_RESID, res=res, blend=bl, alpha=alpha, rows_per_alpha=rpg, **kw(bl))
            r3 = res.clone()
            ops.gemm(a, w, None, epilogue=ops.EPI_RESID, res=r3, **kw(r3))
            out[f"resid_{name}_{M}x{N}x{K}"] = {
                "gate": rel(r1, res.double() + gate.double()[rows] * y),
                "blend": rel(bl, al * blend.double() + (1 - al) * (res.double() + y)),
                "plain": rel(r3, res.double() + a.double() @ w.double().T), "tol": tol}
    out["launches_resid"] = launches() - n0
    n0 = launches()
    # ragged M and N % 256 != 0: the general form of the 4-wave kernels (round 5; its own battery: test_round5_kernels_gpu.py); one K
    # step: not covered, the 8-wave kernels answer and the counter stays
    for M, N, K in [(300, 256, 128), (256, 320, 128), (256, 256, 64)]:
        a, w = rnd((M, K), 18), rnd((N, K), 19, K ** -0.5)
        out[f"fallback_{M}x{N}x{K}"] = rel(ops.gemm(a, w, None, split_k=1), a.double() @ w.double().T)
    out["launches_fallback"] = launches() - n0
    torch.cuda.synchronize()
    print("GEMM4W " + json.dumps(out))


def test_four_wave_gemm_kernels_opt_in():
    if not torch.cuda.is_available():
        pytest.fail("the gpu-marked tests need a HIP device (torch.cuda.is_available() is False)")
    env = dict(os.environ, DWM_GEMM4W="1")
    r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("GEMM4W ")]
    assert len(line) == 1, r.stdout[-2000:]
    out = json.loads(line[0][7:])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "gpu_parity.log"), "a") as f:
        f.write(json.dumps({"test": "gemm4w_opt_in", **out}) + "\n")
    # every covered call went through the 4-wave kernels, no uncovered one did
    assert out["launches_plain"] == 4 * 5 and out["launches_geglu"] == 2 and out["launches_rmshead"] == 2
    assert out["launches_resid"] == 3 * 2 * 3 and out["launches_fallback"] == 2, out
    for k, v in out.items():
        if k.startswith("launches"):
            continue
        if isinstance(v, dict):
            tol = v.pop("tol", TOL)
            assert all(e < tol for e in v.values()), (k, v)
        else:
            assert v < TOL, (k, v)


if __name__ == "__main__":
    _battery()
