"""debug aid: back-to-back launches (no host synchronisation between them, as the tests and the model issue them) against launches with
a synchronisation in between: which launch differs, where (problem / head / row / dim), by how much, and which side is closer to fp32?"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from opendwm_amd import ops
from tests.test_hip_gpu import _rand
dev = torch.device("cuda:0")
bf16 = torch.bfloat16
VAR = int(sys.argv[1], 0) if len(sys.argv) > 1 else 0
for (I, N, Lc, heads, hs) in [(150, 256, 40, 4, 2), (192, 448, 154, 24, 6)]:
    D = heads * 64
    L = N + Lc
    qkv = _rand((I * N, 3 * D), dev, 21)
    cqkv = _rand((I * Lc, 3 * D), dev, 22) if Lc else None
    rm = ops.rowmap_identity(I, N)

    def run(variant, sync):
        out = torch.full((I * N, D), float("nan"), dtype=bf16, device=dev)
        cout = torch.full((I * Lc, D), float("nan"), dtype=bf16, device=dev) if Lc else None
        kw = dict(q1=cqkv[:, :D], k1=cqkv[:, D:2 * D], v1=cqkv[:, 2 * D:], out1=cout) if Lc else {}
        ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out, rm, heads, variant=variant, **kw)
        if sync:
            torch.cuda.synchronize()
        return out, cout

    def cat(r):
        return torch.cat([r[0].view(I, N, heads, 64), r[1].view(I, Lc, heads, 64)], 1) if Lc else r[0].view(I, N, heads, 64)
    var = VAR | (hs << 8)
    torch.cuda.synchronize()
    ref = cat(run(var, True))
    ref2 = cat(run(var, True))
    torch.cuda.synchronize()
    rs = [run(var, False) for _ in range(6)]
    torch.cuda.synchronize()
    rs = [cat(r) for r in rs]
    print((I, N, Lc, heads, hs), "variant", hex(var), "synced launches equal:", bool(torch.equal(ref, ref2)),
          "| back-to-back launches equal to the synced one:", [bool(torch.equal(r, ref)) for r in rs], flush=True)
    for i, r in enumerate(rs):
        ne = r != ref
        if not ne.any():
            continue
        idx = ne.nonzero()
        print(f"  launch {i}: {int(ne.sum())} elements differ; max abs {float((r.float() - ref.float()).abs().max()):.3e};",
              "problems", idx[:, 0].unique().numel(), "of", I, "| heads", idx[:, 2].unique().tolist(), "| rows: min", int(idx[:, 1].min()), "max", int(idx[:, 1].max()),
              "| row tiles (32)", torch.bincount(idx[:, 1] // 32, minlength=(L + 31) // 32).tolist(), flush=True)
        # items of the persistent workgroups: item = problem * (heads / hs) + head group; workgroup = item % 256, position = item // 256
        item = idx[:, 0] * (heads // hs) + idx[:, 2] // hs
        print("   workgroups touched", (item % 256).unique().numel(), "| position of the item in its workgroup's walk", torch.bincount(item // 256).tolist(),
              "| head within the item", torch.bincount(idx[:, 2] % hs, minlength=hs).tolist())
        # rows: are whole rows (all 64 dims) perturbed?
        rows = (ne.sum(-1) > 0)
        print("   rows touched", int(rows.sum()), "| differing elements per touched row: mean", float(ne.sum(-1)[rows].float().mean()))
        # which side is closer to an fp32 softmax attention of the same bf16 inputs? (problems with differences only, first 4)
        for p in idx[:, 0].unique()[:4].tolist():
            f = qkv[p * N:(p + 1) * N].float(); cf = cqkv[p * Lc:(p + 1) * Lc].float() if Lc else None
            q = torch.cat([f[:, :D], cf[:, :D]], 0) if Lc else f[:, :D]
            k = torch.cat([f[:, D:2 * D], cf[:, D:2 * D]], 0) if Lc else f[:, D:2 * D]
            v = torch.cat([f[:, 2 * D:], cf[:, 2 * D:]], 0) if Lc else f[:, 2 * D:]
            qh, kh, vh = (t.view(L, heads, 64).transpose(0, 1).double() for t in (q, k, v))
            o = (torch.softmax(qh @ kh.transpose(1, 2) * 0.125, -1) @ vh).transpose(0, 1)          # [L, heads, 64]
            m = ne[p]
            ea = (r[p].double() - o)[m].abs(); eb = (ref[p].double() - o)[m].abs()
            print(f"   problem {p}: {int(m.sum())} differing elements; |launch - fp64| mean {float(ea.mean()):.3e} max {float(ea.max()):.3e};"
                  f" |synced - fp64| mean {float(eb.mean()):.3e} max {float(eb.max()):.3e}; |launch - synced| max {float((r[p].float() - ref[p].float())[m].abs().max()):.3e}")
