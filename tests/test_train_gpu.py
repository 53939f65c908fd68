"""GPU leg (`-m gpu`): the backward / optimizer kernels of the train step against fp32 torch
autograd of the same op on the same (bf16-rounded) inputs.  Gradients are bf16 outputs of fp32
arithmetic, so the per-kernel bound is the same one-rounding tolerance as the forward kernels;
fp32 reductions (bias / modulation gradients) are held to 1e-3."""
import json
import os

import pytest
import torch
import torch.nn.functional as F

from tests.common import rel_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
bf16 = torch.bfloat16
TOL_KERNEL = 6e-3
TOL_REDUCE = 2e-3


def _log(name, **kv):
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "gpu_parity.log"), "a") as f:
        f.write(json.dumps({"test": name, **kv}) + "\n")
    print(name, kv)


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("the gpu-marked tests need a HIP device (torch.cuda.is_available() is False)")
    from opendwm_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _rand(shape, dev, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dev).to(bf16)


def test_transpose_and_segsum(dev):
    from opendwm_amd import train_ops as T
    x = _rand((300, 136), dev, 1)
    xt = T.transpose(x)
    assert xt.shape == (136, 320)
    assert torch.equal(xt[:, :300], x.t()) and torch.count_nonzero(xt[:, 300:]) == 0
    v = _rand((1000, 264), dev, 2)[:, :256]                       # strided view
    assert torch.equal(T.transpose(v)[:, :1000], v.t())
    for shape in ((1003, 1544), (77, 8), (129, 4096), (640, 100)):  # vector path (cols % 8 == 0) and the scalar fallback
        t = _rand(shape, dev, 5)
        tt = T.transpose(t)
        assert torch.equal(tt[:, :shape[0]], t.t()) and torch.count_nonzero(tt[:, shape[0]:]) == 0
    a, b = _rand((1001, 1536), dev, 3), _rand((1001, 1536), dev, 4)
    s = T.segsum(a)
    assert rel_err(s[0], a.float().sum(0)) < TOL_REDUCE
    s2 = T.segsum(a, b, rows_per_group=154)
    ref = torch.stack([(a.float() * b.float())[i * 154:(i + 1) * 154].sum(0) for i in range(7)])
    e = rel_err(s2, ref)
    _log("segsum", rel=e)
    assert s2.shape == (7, 1536) and e < TOL_REDUCE


@pytest.mark.parametrize("act", ["gelu_tanh", "silu"])
def test_activation_fwd_bwd(dev, act):
    from opendwm_amd import ops, train_ops as T
    code = ops.ACT_GELU_TANH if act == "gelu_tanh" else ops.ACT_SILU
    f = (lambda t: F.gelu(t, approximate="tanh")) if act == "gelu_tanh" else F.silu
    x, dy = _rand((512, 1024), dev, 1, 2.0), _rand((512, 1024), dev, 2)
    xr = x.float().requires_grad_(True)
    y = f(xr)
    y.backward(dy.float())
    e1, e2 = rel_err(T.act_fwd(x, code), y), rel_err(T.act_bwd(x, dy, code), xr.grad)
    _log("act_fwd_bwd", act=act, fwd=e1, bwd=e2)
    assert e1 < TOL_KERNEL and e2 < TOL_KERNEL


def test_geglu_fwd_bwd(dev):
    from opendwm_amd import train_ops as T
    u, dg = _rand((300, 2 * 1024), dev, 1, 1.5), _rand((300, 1024), dev, 2)
    ur = u.float().requires_grad_(True)
    hv, gt = ur.chunk(2, -1)
    g = hv * F.gelu(gt)
    g.backward(dg.float())
    e1, e2 = rel_err(T.geglu_fwd(u), g), rel_err(T.geglu_bwd(u, dg), ur.grad)
    _log("geglu_fwd_bwd", fwd=e1, bwd=e2)
    assert e1 < TOL_KERNEL and e2 < TOL_KERNEL


def test_rowcombine(dev):
    from opendwm_amd import train_ops as T
    rows, n, rpg = 600, 512, 100
    a, b, gate = _rand((rows, n), dev, 1), _rand((rows, n), dev, 2), _rand((6, n), dev, 3)
    ca, cb = torch.rand(3, device=dev), torch.rand(3, device=dev)
    idx = torch.arange(rows, device=dev)
    ref = a.float() * gate.float()[idx // rpg] * ca[idx // 200][:, None] + b.float() * cb[idx // 200][:, None]
    out = T.rowcombine(a, gate_a=gate, rows_per_gate_a=rpg, coef_a=ca, rows_per_coef_a=200, b=b, coef_b=cb, rows_per_coef_b=200)
    assert rel_err(out, ref) < TOL_KERNEL
    assert rel_err(T.rowcombine(a, b=b), a.float() + b.float()) < TOL_KERNEL
    assert rel_err(T.rowcombine(a, gate_a=gate, rows_per_gate_a=rpg), a.float() * gate.float()[idx // rpg]) < TOL_KERNEL


@pytest.mark.parametrize("D,mode", [(1536, "mod"), (1536, "mod2"), (1536, "affine"), (1536, "affine_add"), (320, "affine")])
def test_layernorm_bwd(dev, D, mode):
    from opendwm_amd import ops, train_ops as T
    I, N = 5, 77
    rows = I * N
    x, dy, dy2 = _rand((rows, D), dev, 1, 1.5), _rand((rows, D), dev, 2), _rand((rows, D), dev, 3)
    idx = torch.arange(rows, device=dev) // N
    xr = x.float().requires_grad_(True)
    if mode in ("mod", "mod2"):
        sc, sh, sc2, sh2 = (_rand((I, D), dev, s, 0.3) for s in (4, 5, 6, 7))
        scr, shr, sc2r, sh2r = (t.float().requires_grad_(True) for t in (sc, sh, sc2, sh2))
        xh = F.layer_norm(xr, (D,), eps=1e-6)
        y = xh * (1 + scr[idx]) + shr[idx]
        loss = (y * dy.float()).sum()
        if mode == "mod2":
            loss = loss + ((xh * (1 + sc2r[idx]) + sh2r[idx]) * dy2.float()).sum()
        loss.backward()
        dg, db = torch.zeros(I, D, device=dev), torch.zeros(I, D, device=dev)
        dg2, db2 = torch.zeros(I, D, device=dev), torch.zeros(I, D, device=dev)
        dx = T.layernorm_bwd(x, dy, eps=1e-6, scale=sc, scale2=sc2 if mode == "mod2" else None, rows_per_mod=N,
                             dy2=dy2 if mode == "mod2" else None, dgamma=dg, dbeta=db,
                             dgamma2=dg2 if mode == "mod2" else None, dbeta2=db2 if mode == "mod2" else None,
                             grad_per_group=True)
        errs = dict(dx=rel_err(dx, xr.grad), dscale=rel_err(dg, scr.grad), dshift=rel_err(db, shr.grad))
        if mode == "mod2":
            errs.update(dscale2=rel_err(dg2, sc2r.grad), dshift2=rel_err(db2, sh2r.grad))
    else:
        w, b = _rand((D,), dev, 4, 0.2) + 1, _rand((D,), dev, 5, 0.2)
        wr, br = w.float().requires_grad_(True), b.float().requires_grad_(True)
        emb = _rand((I, D), dev, 6) if mode == "affine_add" else None
        xin = xr
        if emb is not None:
            embr = emb.float().requires_grad_(True)
            ssum = xr + embr[idx]
            xin = ssum.detach().to(bf16).float() + (ssum - ssum.detach())   # forward rounds the sum to bf16 (straight-through)
        y = F.layer_norm(xin, (D,), wr, br, eps=1e-5)
        (y * dy.float()).sum().backward()
        dg, db = torch.zeros(1, D, device=dev), torch.zeros(1, D, device=dev)
        pre = _rand((rows, D), dev, 8)
        dx = pre.clone()
        T.layernorm_bwd(x, dy, eps=1e-5, dx=dx, accumulate=True, weight=w, addvec=emb, rows_per_add=N, dgamma=dg, dbeta=db)
        errs = dict(dx=rel_err(dx, xr.grad + pre.float()), dw=rel_err(dg[0], wr.grad), db=rel_err(db[0], br.grad))
        # consistency with the forward kernel's output
        yk = ops.layernorm(x, eps=1e-5, weight=w, bias=b, addvec=emb, rows_per_add=N,
                           xsum=torch.empty_like(x) if emb is not None else None)
        errs["fwd"] = rel_err(yk, y)
    _log("layernorm_bwd", D=D, mode=mode, **errs)
    assert all(v < (TOL_REDUCE if k.startswith("ds") or k in ("dw", "db") else TOL_KERNEL) for k, v in errs.items()), errs


def test_rmsnorm_heads_fwd_bwd(dev):
    from opendwm_amd import train_ops as T
    rows, heads = 333, 24
    ncols = 2 * heads * 64
    x, dy = _rand((rows, ncols + 64), dev, 1, 1.3)[:, :ncols], _rand((rows, ncols), dev, 2)
    wq, wk = _rand((64,), dev, 3, 0.2) + 1, _rand((64,), dev, 4, 0.2) + 1
    w = torch.cat([wq.repeat(heads), wk.repeat(heads)]).contiguous()
    xr, wr = x.float().requires_grad_(True), w.float().requires_grad_(True)
    xh = xr.view(rows, 2 * heads, 64)
    y = (xh * torch.rsqrt(xh.pow(2).mean(-1, keepdim=True) + 1e-6)).view(rows, ncols) * wr
    y.backward(dy.float())
    xk = x.clone()
    rinv = T.rmsnorm_heads_train_(xk, w, 1e-6)
    e_f = rel_err(xk, y)
    dw = torch.zeros(ncols, device=dev)
    dx = T.rmsnorm_heads_bwd_(xk, rinv, w, dy.clone(), dw)
    e_x, e_w = rel_err(dx, xr.grad), rel_err(dw, wr.grad)
    _log("rmsnorm_heads_bwd", fwd=e_f, dx=e_x, dw=e_w)
    # dx goes through y / w (bf16 y): one extra rounding in the normalised activations
    assert e_f < TOL_KERNEL and e_x < 1.5e-2 and e_w < 1e-2


@pytest.mark.parametrize("M,N,K", [(896, 1536, 1536), (154 * 3, 4608, 1536), (96, 1536, 256)])
def test_linear_backward(dev, M, N, K):
    from opendwm_amd import ops, train_ops as T
    x, w, b, dy = _rand((M, K), dev, 1), _rand((N, K), dev, 2, K ** -0.5), _rand((N,), dev, 3), _rand((M, N), dev, 4)
    xr, wr, br = x.float().requires_grad_(True), w.float().requires_grad_(True), b.float().requires_grad_(True)
    F.linear(xr, wr, br).backward(dy.float())
    dx = T.linear_dgrad(dy, T.transpose(w, rows_pad=N))
    dw, db = T.linear_wgrad(dy, x)
    e = dict(dx=rel_err(dx, xr.grad), dw=rel_err(dw, wr.grad), db=rel_err(db, br.grad))
    _log("linear_backward", M=M, N=N, K=K, **e)
    assert e["dx"] < TOL_KERNEL and e["dw"] < TOL_KERNEL and e["db"] < TOL_REDUCE


@pytest.mark.parametrize("M,N,C,ks", [(1024, 256, 256, 1), (2048, 1536, 1536, 4), (4096, 320, 2880, 8), (8192, 1536, 3072, 4), (1280, 72, 8, 2),
                                      (86016, 1536, 1536, 7)])
def test_gemm_tn_equals_transposed_path(dev, M, N, C, ks):
    """dwm_gemm_tn (both operands as they are, transposing LDS reads) against the path it replaces - two explicit transposes + the NT
    GEMM: the same products in the same order over the same K ranges, so BIT-identical; and against the fp32 product."""
    from opendwm_amd import ops, train_ops as T
    dy, x = _rand((M, N), dev, 1), _rand((M, C), dev, 2, M ** -0.5)
    got = T.gemm_tn(dy, x, split_k=ks)
    e = rel_err(got, dy.float().t() @ x.float())
    _log("gemm_tn", M=M, N=N, C=C, split_k=ks, rel=e)
    assert got.shape == (N, C) and e < TOL_KERNEL
    old = ops.gemm(T.transpose(dy), T.transpose(x), None, split_k=ks)
    assert torch.equal(got, old)
    auto = T.gemm_tn(dy, x)                          # automatic range count (fills the last round of workgroups): another summation order
    assert rel_err(auto, dy.float().t() @ x.float()) < TOL_KERNEL
    # strided operands (column slices of wider buffers)
    wide_dy, wide_x = _rand((M, N + 64), dev, 3), _rand((M, C + 8), dev, 4, M ** -0.5)
    got2 = T.gemm_tn(wide_dy[:, 64:], wide_x[:, :C], split_k=ks)
    assert rel_err(got2, wide_dy[:, 64:].float().t() @ wide_x[:, :C].float()) < TOL_KERNEL


@pytest.mark.parametrize("I,h,w,C,N", [(3, 16, 28, 128, 320), (2, 5, 7, 64, 72), (6, 32, 56, 320, 320)])
def test_conv_wgrad_tn_vs_autograd(dev, I, h, w, C, N):
    """the 3x3 weight gradient as ONE dwm_gemm_tn launch over the padded grid (dy scattered onto it, nine row shifts) against fp32
    autograd of F.conv2d and against the per-tap path it replaces"""
    from opendwm_amd import ops, train_ops as T
    grid = ops.PaddedGrid(I, h, w)
    x = _rand((I, C, h, w), dev, 1)
    dy = _rand((I, N, h, w), dev, 2, (I * h * w) ** -0.5)
    wt = torch.zeros((N, C, 3, 3), device=dev, requires_grad=True)
    F.conv2d(x.float(), wt, None, padding=1).backward(dy.float())
    ref = wt.grad.permute(0, 2, 3, 1).reshape(N, 9 * C)                   # tap-major [N, 9*C]
    idx = grid.interior_index().to(dev)
    xp = torch.zeros((grid.rows, C), dtype=bf16, device=dev)
    xp[idx] = x.permute(0, 2, 3, 1).reshape(-1, C)
    dyc = dy.permute(0, 2, 3, 1).reshape(-1, N).contiguous()
    got = T.conv_wgrad(dyc, xp, idx, grid.tap_shifts())
    e = rel_err(got, ref)
    tn0 = T.WGRAD_TN
    try:
        T.WGRAD_TN = False
        old = T.conv_wgrad(dyc, xp, idx, grid.tap_shifts())
    finally:
        T.WGRAD_TN = tn0
    _log("conv_wgrad_tn", I=I, h=h, w=w, C=C, N=N, rel=e, rel_old_path=rel_err(old, ref), rel_vs_old=rel_err(got, old))
    assert e < TOL_KERNEL and rel_err(old, ref) < TOL_KERNEL


def test_adamw_matches_torch(dev):
    from opendwm_amd import train_ops as T
    n = 100_003
    g0 = torch.Generator(device="cpu").manual_seed(0)
    p = torch.randn(n, generator=g0).to(dev)
    ref = torch.nn.Parameter(p.clone())
    opt = torch.optim.AdamW([ref], lr=3e-4, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.01)
    m, v, pb = torch.zeros_like(p), torch.zeros_like(p), torch.empty(n, dtype=bf16, device=dev)
    for step in range(1, 4):
        g = torch.randn(n, generator=g0).to(dev)
        ref.grad = g.clone()
        opt.step()
        T.adamw_(p, g, m, v, pb, lr=3e-4, beta1=0.9, beta2=0.95, eps=1e-8, weight_decay=0.01, step=step)
    assert rel_err(p, ref.detach()) < 1e-6
    assert torch.equal(pb, p.to(bf16))
    out = T.cast_f32(pb)
    assert torch.equal(out, pb.float())


# ------------------------------------------------------------------------------- attention backward
def _attn_ref_autograd(qkv, cqkv, rows, heads, do, cdo, mask=None):
    """fp32 autograd reference: qkv [R, 3D] (+ context cqkv [P*L1, 3D]) -> grads wrt qkv, cqkv."""
    from oracle import ctsd_oracle as O
    P, L0 = rows.shape
    D = heads * 64
    x = qkv.float().requires_grad_(True)
    c = cqkv.float().requires_grad_(True) if cqkv is not None else None

    def gather(t, t1):
        g = t[rows.reshape(-1)].view(P, L0, heads, 64)
        if t1 is not None:
            g = torch.cat([g, t1.view(P, -1, heads, 64)], 1)
        return g.transpose(1, 2)
    Q = gather(x[:, :D], None if c is None else c[:, :D])
    K = gather(x[:, D:2 * D], None if c is None else c[:, D:2 * D])
    V = gather(x[:, 2 * D:], None if c is None else c[:, 2 * D:])
    o = O.sdpa(Q, K, V, None if mask is None else mask[:, None]).transpose(1, 2).reshape(P, -1, D)
    o0 = torch.zeros(qkv.shape[0], D, device=qkv.device).index_put((rows.reshape(-1),), o[:, :L0].reshape(-1, D))
    loss = (o0 * do.float()).sum()
    if c is not None:
        loss = loss + (o[:, L0:].reshape(-1, D) * cdo.float()).sum()
    loss.backward()
    return o0.detach(), x.grad, (None if c is None else c.grad)


@pytest.mark.parametrize("I,N,Lc,heads", [(2, 448, 154, 4), (2, 100, 0, 2), (2, 64, 10, 2), (3, 16, 0, 2), (1, 300, 3, 3)])
def test_attention_backward_joint(dev, I, N, Lc, heads):
    from opendwm_amd import ops
    D = heads * 64
    qkv, do = _rand((I * N, 3 * D), dev, 1), _rand((I * N, D), dev, 3)
    cqkv = _rand((I * Lc, 3 * D), dev, 2) if Lc else None
    cdo = _rand((I * Lc, D), dev, 4) if Lc else None
    out = torch.zeros((I * N, D), dtype=bf16, device=dev)
    cout = torch.zeros((I * Lc, D), dtype=bf16, device=dev) if Lc else None
    rm = ops.rowmap_identity(I, N)
    lse = torch.empty(I * heads * (N + Lc), dtype=torch.float32, device=dev)
    kw = dict(q1=cqkv[:, :D], k1=cqkv[:, D:2 * D], v1=cqkv[:, 2 * D:], out1=cout) if Lc else {}
    ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out, rm, heads, lse=lse, **kw)
    dqkv = torch.zeros_like(qkv)
    dcqkv = torch.zeros_like(cqkv) if Lc else None
    if Lc:
        kw.update(dout1=cdo, dq1=dcqkv[:, :D], dk1=dcqkv[:, D:2 * D], dv1=dcqkv[:, 2 * D:])
    ops.attention_bwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out, do, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:],
                      rm, heads, lse, **kw)
    o_ref, g_ref, cg_ref = _attn_ref_autograd(qkv, cqkv, rm.rows().to(dev), heads, do, cdo)
    e = dict(fwd=rel_err(out, o_ref), dq=rel_err(dqkv[:, :D], g_ref[:, :D]), dk=rel_err(dqkv[:, D:2 * D], g_ref[:, D:2 * D]),
             dv=rel_err(dqkv[:, 2 * D:], g_ref[:, 2 * D:]))
    if Lc:
        e.update(dq1=rel_err(dcqkv[:, :D], cg_ref[:, :D]), dk1=rel_err(dcqkv[:, D:2 * D], cg_ref[:, D:2 * D]),
                 dv1=rel_err(dcqkv[:, 2 * D:], cg_ref[:, 2 * D:]))
    _log("attention_bwd_joint", I=I, N=N, Lc=Lc, heads=heads, **e)
    # P and dS are rounded to bf16 before their second MFMA (as P is in the forward): 1e-2
    assert all(v < 1e-2 for v in e.values()), e


@pytest.mark.parametrize("kind", ["crossview_rowwise", "temporal_rowwise", "temporal_pointwise", "crossview_full"])
def test_attention_backward_rowmaps_and_masks(dev, kind):
    from opendwm_amd import ops
    from oracle import ctsd_oracle as O
    B, T, V, h, w, heads = 2, 5, 6, 3, 7, 2
    D = heads * 64
    rm = getattr(ops, "rowmap_" + kind)(B, T, V, h, w)
    R = B * T * V * h * w
    qkv, do = _rand((R, 3 * D), dev, 3), _rand((R, D), dev, 5)
    out = torch.zeros((R, D), dtype=bf16, device=dev)
    gmask = ref_mask = None
    if kind.startswith("crossview"):
        gmask = O.ring_crossview_mask(B, V).to(dev)
        gmask[1, 2, 5] = True
        p = torch.arange(rm.n_problems, device=dev)[:, None, None]
        l = torch.arange(rm.L0, device=dev)
        ref_mask = gmask[p // rm.p_per_mask, ((l // rm.group_size) % V)[None, :, None], ((l // rm.group_size) % V)[None, None, :]]
    lse = torch.empty(rm.n_problems * heads * rm.L0, dtype=torch.float32, device=dev)
    ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out, rm, heads, group_mask=gmask, lse=lse)
    dqkv = torch.zeros_like(qkv)
    ops.attention_bwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out, do, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:],
                      rm, heads, lse, group_mask=gmask)
    _, g_ref, _ = _attn_ref_autograd(qkv, None, rm.rows().to(dev), heads, do, None, mask=ref_mask)
    e = rel_err(dqkv, g_ref)
    _log("attention_bwd_rowmap", kind=kind, rel=e)
    assert e < 1e-2
    if ref_mask is not None:                                   # the same through the dense-mask mode
        d2 = torch.zeros_like(qkv)
        ops.attention_bwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out, do, d2[:, :D], d2[:, D:2 * D], d2[:, 2 * D:],
                          rm, heads, lse, dense_mask=ref_mask)
        assert rel_err(d2, g_ref) < 1e-2


# ------------------------------------------------------------------------------- blocks / model gradients
def _train_model(cfg, sd, dev):
    from opendwm_amd.dit import DiTCrossviewTemporalConditionModel
    m = DiTCrossviewTemporalConditionModel(**cfg)
    m.load_state_dict(sd)
    return m.to(dev).train()          # fp32 master parameters; bf16 shadows are made on first use


def _oracle_grads(sd, cfg, inp, wgt, dev, mixer_cond=None):
    """fp32 autograd through the oracle.  `mixer_cond` (a dict) receives, per AlphaBlender, the conditioning of its
    d(alpha) = sum dy * (x_spatial - x_temporal): sum |terms| / |sum terms| over the samples whose flag is off."""
    from oracle import ctsd_oracle as O
    sdo = {k: (v.to(dev).clone().requires_grad_(True) if v.is_floating_point() else v.to(dev)) for k, v in sd.items()}
    blend0, recs = O.alpha_blender, []

    def blender(sd_, p, a, b, image_only):
        out = blend0(sd_, p, a, b, image_only)
        rec = dict(p=p, d=(a - b).detach(), on=(~image_only).detach())
        out.register_hook(lambda gr, rec=rec: rec.__setitem__("dy", gr.detach()))
        recs.append(rec)
        return out
    O.alpha_blender = blender
    try:
        ref = O.dit_forward(sdo, cfg, **inp)
        (ref * wgt).sum().backward()
    finally:
        O.alpha_blender = blend0
    if mixer_cond is not None:
        for rec in recs:
            t = (rec["dy"] * rec["d"]).double()
            t = t * rec["on"].view(-1, *([1] * (t.dim() - 1))).to(t.dtype)
            mixer_cond[rec["p"] + ".mix_factor"] = (t.abs().sum() / t.sum().abs().clamp_min(1e-300)).item()
    return ref.detach(), {k: v.grad for k, v in sdo.items() if torch.is_tensor(v) and v.requires_grad}


def test_segsum_diff_kernel_exact_on_its_own_inputs(dev):
    """dwm_segsum_diff (per-group column sums of dy * (a - b): the d(alpha) of the AlphaBlender mixers) against the fp64 sum
    of the SAME bf16 inputs: a well conditioned case (every term positive: relative error <= 1e-4) and a heavily cancelling one
    (error measured against the sum of the absolute terms, which is what fp32 accumulation can promise)"""
    from opendwm_amd import train_ops as T
    g = torch.Generator().manual_seed(3)
    rows, cols, rpg = 4 * 448, 1536, 448
    a = torch.randn(rows, cols, generator=g).to(dev).to(bf16)
    b = torch.randn(rows, cols, generator=g).to(dev).to(bf16)
    dy_mag = torch.rand(rows, cols, generator=g).to(dev)
    res = {}
    for name, dy in (("aligned", (dy_mag * torch.sign(a.float() - b.float())).to(bf16)),
                     ("cancelling", (torch.randn(rows, cols, generator=g).to(dev) * 1e-2).to(bf16))):
        out = T.segsum_diff(dy, a, b, rows_per_group=rpg).double()
        t = (dy.double() * (a.double() - b.double())).view(rows // rpg, rpg, cols)
        exact, tabs = t.sum(1), t.abs().sum(1)
        res[name] = dict(rel_to_sum=((out - exact).abs().sum() / exact.abs().sum()).item(),
                         rel_to_abs_terms=((out - exact).abs().max() / tabs.max()).item(),
                         conditioning=(tabs.sum() / exact.sum().abs()).item())
    _log("segsum_diff_exactness", **res)
    assert res["aligned"]["rel_to_sum"] < 1e-4 and res["aligned"]["rel_to_abs_terms"] < 2e-6
    assert res["cancelling"]["rel_to_abs_terms"] < 2e-6


@pytest.mark.parametrize("tt", ["rowwise", "pointwise"])
def test_model_gradients_vs_oracle(dev, tt):
    """d(loss)/d(every parameter) of the HIP training path (checkpointed block Functions, bf16) against fp32 autograd
    through the oracle, on the small full-graph configuration; loss = <prediction, fixed random tensor>."""
    from oracle import ctsd_oracle as O
    from opendwm_amd import train
    from tests.common import small_config, small_inputs, to_dev
    cfg = small_config(temporal_attention_type=tt)
    sd = {k: v.to(bf16).float() for k, v in O.make_state_dict(cfg, 0).items()}
    inp = small_inputs(cfg, 0)
    inp = {k: (v.to(bf16).float() if v.is_floating_point() and k not in ("timestep", "added_time_ids") else v) for k, v in inp.items()}
    di = to_dev(inp, dev)
    g = torch.Generator(device="cpu").manual_seed(11)
    wgt = torch.randn(inp["sample"].shape, generator=g).to(dev)
    cond = {}
    ref, gref = _oracle_grads(sd, cfg, di, wgt, dev, mixer_cond=cond)

    m = _train_model(cfg, sd, dev)
    kw = dict(di)
    out = train.forward_train(m, kw.pop("sample"), kw.pop("timestep"), kw.pop("encoder_hidden_states"), kw.pop("pooled_projections"),
                              crossview_attention_mask=kw.get("crossview_attention_mask"), added_time_ids=kw.get("added_time_ids"))
    e_fwd = rel_err(out, ref)
    from tests.common import capture_segsum_diff, check_mixer_gradients
    with capture_segsum_diff() as seg_calls:
        (out.float() * wgt).sum().backward()
    errs, num, den, missing = {}, 0.0, 0.0, []
    for name, p in m.named_parameters():
        if name not in gref or gref[name] is None:
            continue
        if p.grad is None:
            missing.append(name)
            continue
        a, b = p.grad.double().cpu(), gref[name].double().cpu()
        errs[name] = ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
        num += float((a - b).pow(2).sum())
        den += float(b.pow(2).sum())
    glob = (num / den) ** 0.5
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    mixers = {n: dict(rel=errs[n], cond=cond[n]) for n in cond}
    # scalar mixer parameters: d(alpha) = <dy, h - block(h)> is ONE heavily cancelling sum over every activation of the block
    # (fp32 difference and accumulation in dwm_segsum_diff): the kernel is held against the fp64 sum of its own inputs (at any
    # conditioning), the gradient against the flat 5e-2 where the sum is reasonably conditioned
    wk = check_mixer_gradients(mixers, seg_calls, 5e-2)
    _log("model_gradients", temporal=tt, fwd=e_fwd, global_rel=glob, worst=worst, n_params=len(errs), missing=missing, mixers=mixers,
         segsum_kernel_vs_own_inputs=wk)
    assert not missing, missing
    assert e_fwd < 2e-2 and glob < 3e-2, (glob, worst)
    assert all(v < 0.15 for n, v in errs.items() if n not in cond), worst


def test_adamw_step_updates_shadows(dev):
    """one optimizer step through the HIP AdamW: matches torch.optim.AdamW and the next forward sees the new weights"""
    from oracle import ctsd_oracle as O
    from opendwm_amd import train
    from tests.common import small_config, small_inputs, to_dev
    cfg = small_config()
    sd = {k: v.to(bf16).float() for k, v in O.make_state_dict(cfg, 0).items()}
    di = to_dev(small_inputs(cfg, 0), dev)
    m = _train_model(cfg, sd, dev)
    opt = train.AdamW(m.parameters(), lr=1e-3, betas=(0.9, 0.99), weight_decay=0.01)

    def run():
        kw = dict(di)
        return train.forward_train(m, kw.pop("sample"), kw.pop("timestep"), kw.pop("encoder_hidden_states"), kw.pop("pooled_projections"),
                                   crossview_attention_mask=kw.get("crossview_attention_mask"), added_time_ids=kw.get("added_time_ids"))
    out0 = run()
    out0.float().pow(2).mean().backward()
    ref_p = {n: p.detach().clone() for n, p in m.named_parameters()}
    ref_g = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
    opt.step()
    # torch reference of the same update
    for n, p in m.named_parameters():
        if n not in ref_g:
            continue
        q = torch.nn.Parameter(ref_p[n].clone())
        q.grad = ref_g[n].float()
        torch.optim.AdamW([q], lr=1e-3, betas=(0.9, 0.99), weight_decay=0.01).step()
        assert rel_err(p.detach(), q.detach()) < 1e-5, n
    opt.zero_grad()
    out1 = run()
    assert not torch.equal(out1, out0)            # the bf16 shadows were refreshed by the optimizer kernel
    from opendwm_amd.blocks import STORE
    w = m.transformer_blocks[0].ff.net[2].weight
    assert torch.equal(STORE.bf(w), w.detach().to(bf16))


def test_train_step_loss_and_descent(dev):
    """CTSDTrainer: the loss of one batch equals the oracle's restatement of train_step, and a few optimizer steps on
    the same batch reduce it."""
    from oracle import ctsd_oracle as O
    from opendwm_amd.pipeline import CTSDTrainer
    from tests.common import small_config, small_inputs, to_dev
    cfg = small_config()
    sd = {k: v.to(bf16).float() for k, v in O.make_state_dict(cfg, 0).items()}
    inp = small_inputs(cfg, 0)
    inp = {k: (v.to(bf16).float() if v.is_floating_point() and k not in ("timestep", "added_time_ids") else v) for k, v in inp.items()}
    lat = inp.pop("sample")
    inp.pop("timestep")
    g = torch.Generator().manual_seed(5)
    noise = torch.randn(lat.shape, generator=g)
    idx = torch.tensor([250, 800])
    ref = O.train_loss(sd, cfg, lat, inp, idx, noise).item()
    m = _train_model(cfg, sd, dev)
    tr = CTSDTrainer(m, lr=2e-4, weight_decay=0.0)
    di = to_dev(inp, dev)
    l0 = tr.loss(lat.to(dev), di, timestep_indices=idx, noise=noise).item()
    _log("train_step_loss", ours=l0, oracle=ref)
    assert abs(l0 - ref) / ref < 2e-2
    tr.optimizer.zero_grad()
    losses = [tr.train_step(lat.to(dev), di, timestep_indices=idx, noise=noise).item() for _ in range(4)]
    _log("train_step_descent", losses=losses)
    assert losses[-1] < losses[0]


def test_train_step_grad_scaler_mode(dev):
    """training_config["enable_grad_scaler"] (set by every shipped CTSD training config; ctsd.py:1040-1048, 1401-1432): the loss is
    scaled before the backward, the gradients are unscaled before the clip, a step whose gradients hold inf / nan is skipped and
    halves the scale.  The backward is linear in the upstream gradient and the scale is a power of two, so a scaled step must
    land on the weights of the unscaled one (up to the few fp32 sums whose order autograd may change: 1e-5); a poisoned batch must
    leave every weight bit-identical."""
    from oracle import ctsd_oracle as O
    from opendwm_amd.pipeline import CTSDTrainer
    from tests.common import small_config, small_inputs, to_dev
    cfg = small_config()
    sd = {k: v.to(bf16).float() for k, v in O.make_state_dict(cfg, 0).items()}
    inp = small_inputs(cfg, 0)
    inp = {k: (v.to(bf16).float() if v.is_floating_point() and k not in ("timestep", "added_time_ids") else v) for k, v in inp.items()}
    lat = inp.pop("sample")
    inp.pop("timestep")
    noise = torch.randn(lat.shape, generator=torch.Generator().manual_seed(5))
    idx = torch.tensor([250, 800])
    di = to_dev(inp, dev)
    tc = {"max_norm_for_grad_clip": 1.0}
    plain = CTSDTrainer(_train_model(cfg, sd, dev), lr=2e-4, weight_decay=0.0, training_config=dict(tc))
    scaled = CTSDTrainer(_train_model(cfg, sd, dev), lr=2e-4, weight_decay=0.0, training_config=dict(tc, enable_grad_scaler=True))
    assert plain.grad_scaler is None and scaled.grad_scaler is not None and scaled.grad_scaler.get_scale() == 65536.0
    l0 = plain.train_step(lat.to(dev), di, timestep_indices=idx, noise=noise).item()
    l1 = scaled.train_step(lat.to(dev), di, timestep_indices=idx, noise=noise).item()
    worst = max(((a.detach() - b.detach()).abs().max() / b.detach().abs().max().clamp_min(1e-12)).item()
                for a, b in zip(scaled.model.parameters(), plain.model.parameters()))
    moved = any(not torch.equal(p.detach().cpu(), sd[k].to(p.dtype)) for k, p in scaled.model.named_parameters() if k in sd)
    assert abs(l0 - l1) <= 1e-6 * abs(l0) and moved and worst < 1e-5 and scaled.grad_scaler.get_scale() == 65536.0
    before = [p.detach().clone() for p in scaled.model.parameters()]
    t_before = scaled.optimizer.t
    bad = lat.clone()
    bad[0, 0, 0, 0, 0, 0] = float("inf")
    scaled.train_step(bad.to(dev), di, timestep_indices=idx, noise=noise)
    same = all(torch.equal(a, b.detach()) for a, b in zip(before, scaled.model.parameters()))
    _log("grad_scaler", loss_plain=l0, loss_scaled=l1, worst_param_rel=worst, skipped_step_kept_weights=same,
         scale_after_skip=scaled.grad_scaler.get_scale())
    assert same and scaled.optimizer.t == t_before and scaled.grad_scaler.get_scale() == 32768.0
    assert all(p.grad is None or not p.grad.any() for p in scaled.model.parameters())        # zero_grad ran
    l2 = scaled.train_step(lat.to(dev), di, timestep_indices=idx, noise=noise).item()        # and the next clean step trains on
    assert l2 == l2 and scaled.optimizer.t == t_before + 1


@pytest.mark.parametrize("style", ["diffusion_forcing", "ctsd"])
def test_trainer_task_styles_vs_oracle(dev, style):
    """CTSDTrainer.loss in the two non-trivial training styles (ctsd.py:619-741, 1232-1237, 1363-1367): per-frame timesteps
    and the image-task / reference-augmentation mix of the diffusion-forcing checkpoints, or clean reference frames at
    timestep 0 excluded from the loss.  Reference value: the same host draws (make_training_pair / make_input_for_prediction
    are pinned against the executed reference code in tests/test_reference_fixtures_cpu.py) through the fp32 oracle forward."""
    from oracle import ctsd_oracle as O
    from opendwm_amd.pipeline import CTSDTrainer, make_input_for_prediction
    from tests.common import small_config, small_inputs, to_dev
    cfg = small_config()
    sd = {k: v.to(bf16).float() for k, v in O.make_state_dict(cfg, 0).items()}
    inp = small_inputs(cfg, 0)
    inp = {k: (v.to(bf16).float() if v.is_floating_point() and k not in ("timestep", "added_time_ids") else v) for k, v in inp.items()}
    lat = inp.pop("sample")
    inp.pop("timestep")
    inp.pop("disable_temporal", None)                      # set by the task mixer
    B, T, V = lat.shape[:3]
    common = {"frame_prediction_style": style}
    tcfg = {"diffusion_forcing": {"image_generation_ratio": 0.5, "reference_frame_scale_std": 0.05, "reference_frame_offset_std": 0.05},
            "ctsd": {"all_reference_visible_ratio": 0.5, "reference_visible_rate": 0.7, "disable_reference_frame_loss": True}}[style]
    rlc = 0 if style == "diffusion_forcing" else 2
    noise = torch.randn(lat.shape, generator=torch.Generator().manual_seed(5))
    idx = torch.tensor([[250, 800, 40], [600, 10, 990]]) if style == "diffusion_forcing" else torch.tensor([250, 800])
    m = _train_model(cfg, sd, dev)
    tr = CTSDTrainer(m, lr=2e-4, weight_decay=0.0, common_config=common, training_config=tcfg, reference_latent_count=rlc)
    # reference: the same pieces on the host, fp32 oracle forward
    noisy, ts, sig, _ = tr.make_training_pair(lat, timestep_indices=idx, noise=noise)
    made, mts, extra, ind = make_input_for_prediction(noisy, lat, ts, tcfg, common, torch.Generator().manual_seed(9), rlc)
    pred = O.dit_forward(sd, cfg, made.to(bf16).float(), mts, **dict(inp, **extra))
    x0, target = pred * (-sig) + made, lat
    if tcfg.get("disable_reference_frame_loss"):
        keep = ~ind.view(B, T, V, 1, 1, 1)
        x0, target = x0 * keep, target * keep
    ref = torch.nn.functional.mse_loss(x0, target).item()
    ours = tr.loss(lat.to(dev), to_dev(inp, dev), generator=torch.Generator().manual_seed(9), timestep_indices=idx, noise=noise)
    _log("train_task_style", style=style, ours=ours.item(), oracle=ref, reference_frames=int(ind.sum()))
    assert abs(ours.item() - ref) / ref < 2e-2
    ours.backward()                                        # the gradient path through the task-mixed input works
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters() if p.requires_grad)


def _ddp_worker(rank, world, port, cfg, sd, inp, wgt, path):
    import torch.distributed as dist
    from opendwm_amd import train
    dev = torch.device("cuda:0")                       # both ranks share the one GPU of the test box (gloo moves the buckets)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        m = _train_model(cfg, sd, dev)
        ddp = torch.nn.parallel.DistributedDataParallel(m)
        mine = {k: (v[rank:rank + 1].to(dev) if torch.is_tensor(v) else v) for k, v in inp.items()}
        out, _, _ = ddp(mine.pop("sample"), mine.pop("timestep"), **mine)
        (out[0].float() * wgt[rank:rank + 1].to(dev)).sum().backward()
        if rank == 0:
            torch.save({n: p.grad.detach().cpu() for n, p in m.named_parameters() if p.grad is not None}, path)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_ddp_two_ranks_average_gradients(dev):
    """DistributedDataParallel over the block Functions: two ranks (one batch element each, gloo) end up with the
    mean of their gradients == half the single-process full-batch gradient."""
    import torch.multiprocessing as mp
    from oracle import ctsd_oracle as O
    from tests.common import small_config, small_inputs, to_dev
    cfg = small_config()
    sd = {k: v.to(bf16).float() for k, v in O.make_state_dict(cfg, 0).items()}
    inp = small_inputs(cfg, 0)
    inp = {k: (v.to(bf16) if v.is_floating_point() and k not in ("timestep", "added_time_ids") else v) for k, v in inp.items()}
    g = torch.Generator().manual_seed(7)
    wgt = torch.randn(inp["sample"].shape, generator=g)
    # single process, full batch
    m = _train_model(cfg, sd, dev)
    di = to_dev(inp, dev)
    out, _, _ = m(di.pop("sample"), di.pop("timestep"), **di)
    (out[0].float() * wgt.to(dev)).sum().backward()
    full = {n: p.grad.detach().cpu() for n, p in m.named_parameters() if p.grad is not None}
    import tempfile
    ctx = mp.get_context("spawn")
    port = 29500 + os.getpid() % 2000
    path = os.path.join(tempfile.mkdtemp(), "ddp_grads.pt")
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, cfg, sd, inp, wgt, path)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    got = torch.load(path)
    num = sum(float((2 * got[n].double() - full[n].double()).pow(2).sum()) for n in full)
    den = sum(float(full[n].double().pow(2).sum()) for n in full)
    e = (num / den) ** 0.5
    _log("ddp_two_ranks", rel=e, n_params=len(full))
    assert set(got) == set(full) and e < 5e-3


def test_model_gradients_with_layout_adapter_vs_oracle(dev):
    """the layout branch in training (condition_image_tensor -> ImageAdapter -> residuals added before the first
    blocks, crossview_temporal_dit.py:459-462,491-494): gradients of every adapter parameter (1x1 / 3x3 convolutions,
    zero convs) and of the rest of the model against fp32 autograd through the oracle"""
    from oracle import ctsd_oracle as O
    from opendwm_amd import train
    from tests.common import small_config, small_inputs, to_dev
    acfg = dict(in_channels=6, channels=[128, 128, 128], is_downblocks=[True, False, False], num_res_blocks=2,
                downscale_factor=8, use_zero_convs=True)
    cfg = small_config(condition_image_adapter_config=acfg)
    sd = {k: v.to(bf16).float() for k, v in O.make_state_dict(cfg, 0).items()}
    inp = small_inputs(cfg, 0)
    inp["condition_image_tensor"] = torch.rand(2, 3, 3, 6, 64, 96, generator=torch.Generator().manual_seed(5))
    inp = {k: (v.to(bf16).float() if v.is_floating_point() and k not in ("timestep", "added_time_ids") else v) for k, v in inp.items()}
    di = to_dev(inp, dev)
    wgt = torch.randn(inp["sample"].shape, generator=torch.Generator().manual_seed(11)).to(dev)
    ref, gref = _oracle_grads(sd, cfg, di, wgt, dev)
    m = _train_model(cfg, sd, dev)
    kw = dict(di)
    out = m(kw.pop("sample"), kw.pop("timestep"), **kw)[0][0]              # the reference entry point, train mode
    assert out.grad_fn is not None
    e_fwd = rel_err(out, ref)
    (out.float() * wgt).sum().backward()
    errs, num, den, missing = {}, 0.0, 0.0, []
    for name, p in m.named_parameters():
        if name not in gref or gref[name] is None:
            continue
        if p.grad is None:
            missing.append(name)
            continue
        a, b = p.grad.double().cpu(), gref[name].double().cpu()
        errs[name] = ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
        if name.startswith("condition_image_adapter"):
            num += float((a - b).pow(2).sum())
            den += float(b.pow(2).sum())
    glob = (num / den) ** 0.5
    ad = {n: v for n, v in errs.items() if n.startswith("condition_image_adapter")}
    worst = sorted(ad.items(), key=lambda kv: -kv[1])[:5]
    _log("adapter_gradients", fwd=e_fwd, adapter_global_rel=glob, worst=worst, n_adapter_params=len(ad), missing=missing)
    assert not missing, missing
    assert len(ad) == 32                          # in_conv (2) + 3 x 2 resnets x 4 + 3 zero convs x 2
    # measured 2.8e-2: the adapter sits below the whole bf16 backward of the model, and its 3x3 convolutions see the
    # ReLU mask of bf16 pre-activations (sign flips of near-zero entries); bound 4e-2 on the Frobenius norm over all
    # adapter gradients, 15 % on any single tensor
    assert e_fwd < 2e-2 and glob < 4e-2, (glob, worst)
    assert all(v < 0.15 for v in ad.values()), worst


def test_adapter_weights_are_repacked_after_optimizer_step(dev):
    """two optimizer steps with the layout ImageAdapter: the packed 3x3 / 1x1 weights of its resnets must follow the
    masters (they were cached once and kept the step-0 values).  After the steps the training-mode forward, the eval
    forward AND the layout residual cache must all see the updated adapter: the forward equals the oracle evaluated
    with the model's CURRENT parameters, and differs from the oracle with the initial ones."""
    from oracle import ctsd_oracle as O
    from opendwm_amd.pipeline import CTSDTrainer
    from tests.common import small_config, small_inputs, to_dev
    acfg = dict(in_channels=6, channels=[128, 128, 128], is_downblocks=[True, False, False], num_res_blocks=2,
                downscale_factor=8, use_zero_convs=True)
    cfg = small_config(condition_image_adapter_config=acfg)
    sd0 = {k: v.to(bf16).float() for k, v in O.make_state_dict(cfg, 0).items()}
    inp = small_inputs(cfg, 0)
    inp["condition_image_tensor"] = torch.rand(2, 3, 3, 6, 64, 96, generator=torch.Generator().manual_seed(5))
    inp = {k: (v.to(bf16).float() if v.is_floating_point() and k not in ("timestep", "added_time_ids") else v) for k, v in inp.items()}
    lat = inp.pop("sample")
    ts = inp.pop("timestep")
    m = _train_model(cfg, sd0, dev)
    # only the adapter resnets train, with a large step: a stale packed copy cannot hide behind other parameters
    m.requires_grad_(False)
    for blk in m.condition_image_adapter.body:
        blk.resnets.requires_grad_(True)
    tr = CTSDTrainer(m, lr=3e-2, weight_decay=0.0)
    di = to_dev(inp, dev)
    m.eval()
    with torch.no_grad():
        before = m(lat.to(dev).to(bf16), ts.to(dev), **di)[0][0].float()     # fills the residual cache with step-0 weights
    m.train()
    idx, noise = torch.tensor([250, 800]), torch.randn(lat.shape, generator=torch.Generator().manual_seed(5))
    for _ in range(2):
        tr.train_step(lat.to(dev), di, timestep_indices=idx, noise=noise)
    sd_now = {k: v.detach().float().cpu().to(bf16).float() for k, v in m.state_dict().items()}
    moved = max((sd_now[k] - sd0[k]).abs().max().item() for k in sd0 if ".resnets." in k)
    assert moved > 1e-2, moved
    xin = lat.to(bf16).float()
    ref_now = O.dit_forward(sd_now, cfg, xin, ts, **inp)
    ref_old = O.dit_forward(sd0, cfg, xin, ts, **inp)
    m.eval()
    with torch.no_grad():
        after = m(lat.to(dev).to(bf16), ts.to(dev), **di)[0][0].float()      # same condition tensor object: cache key must miss
    m.train()
    after_train = m(lat.to(dev).to(bf16), ts.to(dev), **di)[0][0].detach().float()
    e_now, e_old, e_train = rel_err(after, ref_now), rel_err(after, ref_old), rel_err(after_train, ref_now)
    _log("adapter_two_step", rel_vs_current_weights=e_now, rel_vs_initial_weights=e_old, train_mode=e_train,
         weights_moved=moved, changed=rel_err(after, before))
    assert e_now < 2e-2 and e_train < 2e-2, (e_now, e_train)
    assert e_old > 3 * e_now, (e_old, e_now)


def test_model_gradients_explicit_perspective_vs_oracle(dev):
    """training with perspective_modeling_type="explicit" (the UniMLVG training configs, configs/ctsd/unimlvg/*): per-token
    embedding = index embedding + RayEncoder.proj(ray features) in front of every cross-view / temporal block
    (crossview_temporal_dit.py:440-458, 528-568) - gradients of `rayencoder.proj.weight`, of the index-embedding MLPs and of
    everything else against fp32 autograd through the oracle"""
    from oracle import ctsd_oracle as O
    from tests.common import small_config, small_inputs, to_dev
    GOLDEN = os.path.join(ROOT, "tests", "golden")
    fx = torch.load(os.path.join(GOLDEN, "reference_forward.pt"))["explicit"]
    cfg = small_config(perspective_modeling_type="explicit")
    sd = {k: v.to(bf16).float() for k, v in O.make_state_dict(cfg, 0).items()}
    inp = small_inputs(cfg, 0)
    inp.pop("added_time_ids")
    inp = {k: (v.to(bf16).float() if v.is_floating_point() and k not in ("timestep",) else v) for k, v in inp.items()}
    inp.update({k: fx[k] for k in ("camera_intrinsics_norm", "camera2referego")})
    di = to_dev(inp, dev)
    wgt = torch.randn(inp["sample"].shape, generator=torch.Generator().manual_seed(11)).to(dev)
    cond = {}
    ref, gref = _oracle_grads(sd, cfg, di, wgt, dev, mixer_cond=cond)
    m = _train_model(cfg, sd, dev)
    kw = dict(di)
    out = m(kw.pop("sample"), kw.pop("timestep"), **kw)[0][0]
    assert out.grad_fn is not None
    e_fwd = rel_err(out, ref)
    (out.float() * wgt).sum().backward()
    errs, num, den, missing = {}, 0.0, 0.0, []
    for name, p in m.named_parameters():
        if name not in gref or gref[name] is None:
            continue
        if p.grad is None:
            missing.append(name)
            continue
        a, b = p.grad.double().cpu(), gref[name].double().cpu()
        errs[name] = ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
        num += float((a - b).pow(2).sum())
        den += float(b.pow(2).sum())
    glob = (num / den) ** 0.5
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    _log("model_gradients_explicit", fwd=e_fwd, global_rel=glob, rayencoder=errs.get("rayencoder.proj.weight"), worst=worst, missing=missing)
    assert not missing and "rayencoder.proj.weight" in errs, (missing, list(errs)[:5])
    assert e_fwd < 2e-2 and glob < 3e-2, (glob, worst)
    assert errs["rayencoder.proj.weight"] < 5e-2
    assert all(v < 0.15 for n, v in errs.items() if n not in cond), worst
