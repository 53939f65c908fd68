import os, sys, json, torch
sys.path.insert(0, "/root/repo")
from opendwm_amd import ops
from scripts.microbench import timeit, rnd
dev = torch.device("cuda:0"); bf16 = torch.bfloat16
H, D = 24, 1536
B, T, V, h, w = 2, 16, 6, 16, 28
I, N, Lc = B * T * V, h * w, 154
qkv = rnd(I * N, 3 * D); cqkv = rnd(I * Lc, 3 * D)
out = torch.empty(I * N, D, device=dev, dtype=bf16); cout = torch.empty(I * Lc, D, device=dev, dtype=bf16)
rm0 = ops.rowmap_identity(I, N)
timeit(lambda: ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out, rm0, H), iters=300)
cases = {"joint L=602": (rm0, dict(q1=cqkv[:, :D], k1=cqkv[:, D:2 * D], v1=cqkv[:, 2 * D:], out1=cout), 602),
         "temporal L=448": (ops.rowmap_temporal_rowwise(B, T, V, h, w), {}, 448)}
for name, (rm, kw, L) in cases.items():
    fl = 4.0 * rm.n_problems * H * L * L * 64
    r = {"case": name}
    for nm, dbg in (("base", 0), ("nostore", 1), ("noDMA", 2), ("noexp", 4), ("nomax", 8), ("noexp_nomax", 12), ("nostore_noDMA", 3), ("all", 15)):
        var = 1 | ((dbg & 15) << 4)
        ms = timeit(lambda: ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out, rm, H, variant=var, **kw))
        r[nm] = round(fl / ms / 1e9)
    print(json.dumps(r), flush=True)
