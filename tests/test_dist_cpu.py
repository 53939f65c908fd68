"""world_size-2 gloo test of the replica launch path (runs on CPU): rank/env parsing, sample
sharding, barrier-bracketed timing with MAX-over-ranks, and that each rank's denoise loop is an
independent replica (the oracle stands in for the device kernels here)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from opendwm_amd import dist as D


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    r, lr, w = D.init("gloo")
    assert (r, lr, w) == (rank, rank, world)
    mine = D.shard_samples(5, r, w)

    from oracle import ctsd_oracle as O
    from tests.common import small_config, small_inputs
    cfg = small_config()
    sd = O.make_state_dict(cfg, 0)
    outs = {}

    def step(i):
        # every rank advances ITS samples by one step; no data-path collective
        import time
        for sidx in mine:
            inp = small_inputs(cfg, seed=sidx, T=1, V=2, text_len=4)
            cond = {k: v for k, v in inp.items() if k not in ("sample", "timestep")}
            g = torch.Generator().manual_seed(sidx)
            lat = torch.randn(1, 1, 2, 16, 8, 12, generator=g)
            outs[sidx] = O.denoise(sd, cfg, lat, cond, steps=2, guidance_scale=4.0, stop=1)
        if rank == 1:
            time.sleep(0.2)          # the slow rank must define the reported time

    dt = D.timed_steps(step, steps=1, warmup=0)
    q.put((rank, mine, dt, {k: v.double().sum().item() for k, v in outs.items()}))
    D.shutdown()


def test_two_rank_replicas_gloo():
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, s0, t0, o0), (r1, s1, t1, o1) = res
    assert s0 == [0, 1, 2] and s1 == [3, 4]                 # disjoint, complete
    assert abs(t0 - t1) < 1e-9 and t0 >= 0.2                # MAX over ranks, identical on both
    assert set(o0) == {0, 1, 2} and set(o1) == {3, 4}
    assert len({round(v, 6) for v in list(o0.values()) + list(o1.values())}) == 5   # independent samples


def test_shard_samples_properties():
    for n in (0, 1, 7, 8, 9, 64):
        for w in (1, 2, 3, 8):
            parts = [D.shard_samples(n, r, w) for r in range(w)]
            flat = [i for p in parts for i in p]
            assert flat == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


# ---------------------------------------------------------------- intra-sample (frame) sharding, opendwm_amd.sharding
def _frame_shard_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    D.init("gloo")
    from opendwm_amd.sharding import FrameShard
    from oracle import ctsd_oracle as O
    from tests.common import small_config
    fs = FrameShard()
    B, T, V, height, width = 2, 4, 2, 4, 3
    res = {}
    # 1. the exchange itself: rank r must end up with rows [r*hl, (r+1)*hl) of EVERY frame, and back
    Dm = 5
    g = torch.Generator().manual_seed(0)
    full = torch.randn(B, T, V, height, width, Dm, generator=g)
    t0, t1 = fs.frame_range(T)
    hl = height // world
    mine = full[:, t0:t1].reshape(-1, Dm).contiguous()
    hx = fs.frames_to_rows(mine, B, t1 - t0, V, height, width)
    res["rows_ok"] = bool(torch.equal(hx.view(B, T, V, hl, width, Dm), full[:, :, :, rank * hl:(rank + 1) * hl]))
    back = fs.rows_to_frames(hx, B, t1 - t0, V, height, width)
    res["round_trip"] = bool(torch.equal(back, mine))
    res["gather"] = bool(torch.equal(fs.gather_frames(full[:, t0:t1].contiguous(), 1), full))
    # 2. a temporal block + mixer on the re-sharded rows == the same block on the whole sample (oracle as the row compute)
    for typ in ("rowwise", "pointwise"):
        cfg = small_config(temporal_attention_type=typ)
        sd = O.make_state_dict(cfg, 0)
        C = cfg["num_attention_heads"] * cfg["attention_head_dim"]
        h = torch.randn(B * T * V, height * width, C, generator=g)
        emb = torch.randn(B * T * V, 1, C, generator=g) * 0.3
        dis = torch.tensor([False, True])
        whole = O.temporal_block_and_mix(sd, cfg, 0, h, emb, B, T, V, width, dis).view(B, T, V, height * width, C)
        loc = h.view(B, T, V, height * width, C)[:, t0:t1].reshape(-1, C).contiguous()
        hx = fs.frames_to_rows(loc, B, t1 - t0, V, height, width)
        y = O.temporal_block_and_mix(sd, cfg, 0, hx.view(B * T * V, hl * width, C), emb, B, T, V, width, dis)
        out = fs.rows_to_frames(y.reshape(-1, C).contiguous(), B, t1 - t0, V, height, width)
        res[typ] = float((out.view(B, t1 - t0, V, height * width, C) - whole[:, t0:t1]).abs().max())
    try:
        fs.check(height, "full")
        res["full_rejected"] = False
    except NotImplementedError:
        res["full_rejected"] = True
    q.put((rank, res))
    D.shutdown()


@pytest.mark.parametrize("world", [2, 4])
def test_frame_shard_exchange_and_temporal_block_gloo(world):
    """frames of one sample on two / four ranks.  The all-to-all re-shard (my frames, all token rows) <->
    (all frames, my token rows) is exact, and a temporal block + mixer run on the re-sharded rows reproduces the
    unsharded block (fp32 oracle compute; 1e-5 abs: only the summation order inside torch kernels may differ)."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_frame_shard_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        assert res[r]["rows_ok"] and res[r]["round_trip"] and res[r]["gather"] and res[r]["full_rejected"], res[r]
        assert res[r]["rowwise"] < 1e-5 and res[r]["pointwise"] < 1e-5, res[r]


# ---------------------------------------------------------------- CTSDDenoiser host logic, one process and frame-sharded,
# against the REAL inference_pipeline (tests/golden/reference_drivers.pt).  The two HIP ops the loop calls are replaced by
# torch stand-ins HERE (test-only: the product has no such fallback), the model by the fixtures' cheap per-frame denoiser.
def _install_fake_ops(P):
    import types
    bf16 = torch.bfloat16

    def cfg_euler_step(pred, latents, guidance, dsigma, model_in=None, group_elems=0):
        u, c = pred.float().chunk(2)
        d = dsigma.reshape(*dsigma.shape, *([1] * (latents.dim() - dsigma.dim()))) if torch.is_tensor(dsigma) else dsigma
        latents += d * (u + guidance * (c - u))
        if model_in is not None:
            B = latents.shape[0]
            model_in[:B].copy_(latents)
            model_in[B:].copy_(latents)
    P.ops = types.SimpleNamespace(cast_bf16=lambda t: t.to(bf16), cfg_euler_step=cfg_euler_step)


class _FrameModel(torch.nn.Module):
    """fake_pred of make_reference_driver_fixtures.py: per-frame, so it needs no exchange between frame shards"""
    frame_shard = None

    def forward(self, x, ts, c=None, scale=None, **kw):
        return [((0.1 * x.float() + 1e-4 * ts.float()[..., None, None, None] + 0.01 * c.float()[..., None, None, None]) * scale)
                .to(torch.bfloat16)], None, None


def _denoiser_worker(rank, world, port, q, split="frames"):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    group = None
    if world > 1:
        D.init("gloo")
        import torch.distributed as dist
        group = dist.group.WORLD
    import opendwm_amd.pipeline as P
    _install_fake_ops(P)
    fx = torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_drivers.pt"))
    res = {}
    for mode, d in fx["inference_pipeline"].items():
        kw = dict(d["kwargs"])
        start, stop, take = kw.pop("start_timestep", 0), kw.pop("stop_timestep", None), kw.pop("take_time", 0)
        df = mode.startswith("diffusion_forcing")
        noise = torch.randn(tuple(d["shape"]), generator=torch.Generator().manual_seed(d["seed"]))
        den = P.CTSDDenoiser(_FrameModel(), guidance_scale=fx["guidance"], inference_steps=d["steps"],
                             **({"frame_group": group} if split == "frames" else {"cfg_group": group}))
        cond = {k: v for k, v in d["batch"].items() if k != "pts"}
        out = den.run(noise, cond, stop=stop, start=start, diffusion_forcing=df, take_time=take, **kw)
        res[mode] = float(((out - d["latents"]).norm() / d["latents"].norm()))
    q.put((rank, res))
    if world > 1:
        D.shutdown()


@pytest.mark.parametrize("world,split", [(1, "frames"), (2, "frames"), (2, "cfg")])
def test_ctsd_denoiser_host_logic_vs_reference_pipeline(world, split):
    """pipeline.CTSDDenoiser - step inputs per frame, reference-frame injection, diffusion-forcing windows, CFG + Euler call
    order, and with two ranks the frame-shard slicing / gather or the CFG split (halves of the guidance batch on the two ranks, one
    all-gather of the prediction per step) - against the latents of the REAL inference_pipeline in its
    four modes.  bf16 model input and prediction (as on the device): 1e-2 relative."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_denoiser_worker, args=(r, world, port, q, split)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        assert set(res[r]) == {"full", "reference_frames", "diffusion_forcing", "diffusion_forcing_warmup"}
        assert all(e < 1e-2 for e in res[r].values()), res[r]


# ---------------------------------------------------------------- bench.py started the way the driver starts it
def test_bench_self_launches_ranks_when_started_plainly():
    """`python bench.py --gpus 2 ...` with no WORLD_SIZE in the environment must start its own ranks (torch.distributed.run,
    127.0.0.1 rendezvous) and print ONE JSON line from rank 0; --debug-cpu-launch swaps RCCL + kernels for gloo + a stand-in
    step so this runs without a GPU.  The torchrun-launched form must keep working too."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--debug-cpu-launch"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1
    assert line["ms_per_step"] >= 20.0                       # the slow rank's sleep: MAX over ranks
    assert abs(line["value"] - 2 * 3 / (line["ms_per_step"] * 3e-3)) < 1e-6 * line["value"]   # whole-job aggregate
    # the preflight every N > 1 launch runs before its timed region (checked all-reduce, distinct ranks) and the
    # gradient-exchange probe of the --train line
    pre = line["preflight"]
    assert pre["world_size"] == 2 and pre["backend"] == "gloo" and pre["allreduce_sum_ok"] is True and pre["allreduce_ms"] > 0
    assert line["allreduce_ms"] > 0
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2",
                        "--warmup", "0", "--debug-cpu-launch"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert len([ln for ln in r.stdout.splitlines() if ln.startswith("{")]) == 1


def test_bench_self_launch_eight_ranks():
    """the driver's first `--gpus 8` run must not die on launch plumbing: the same self-launch with EIGHT ranks (gloo stand-in) -
    rendezvous, preflight with a checked all-reduce over 8 ranks, MAX over ranks, one JSON line"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
                        "--debug-cpu-launch"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["preflight"]["world_size"] == 8 and line["preflight"]["allreduce_sum_ok"] is True
    assert abs(line["value"] - 8 * 2 / (line["ms_per_step"] * 2e-3)) < 1e-6 * line["value"]


# ---------------------------------------------------------------- train step: accumulation + no_sync + bf16 buckets + LR schedule
def _adamw_cpu(p, g, m, v, p_bf16, *, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    """what dwm_adamw computes (torch.optim.AdamW's update), for the CPU ranks of the test below"""
    g = g * grad_scale
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    p.mul_(1 - lr * weight_decay)
    p.addcdiv_(m / (1 - beta1 ** step), (v / (1 - beta2 ** step)).sqrt() + eps, value=-lr)


class _StandInDenoiser(torch.nn.Module):
    """forward signature / 3-tuple return of the model; `frozen` never trains (freezing_pattern)"""

    def __init__(self):
        super().__init__()
        torch.manual_seed(11)
        self.frozen = torch.nn.Linear(4, 4)
        self.mix = torch.nn.Linear(4, 4)
        self.scale = torch.nn.Parameter(torch.tensor([0.5, -0.25, 0.1, 0.3]))

    def forward(self, x, timestep, c=None, **kw):
        h = self.mix(self.frozen(x.float().movedim(3, -1))).movedim(-1, 3)
        return [h * self.scale.view(1, 1, 1, 4, 1, 1) + 1e-3 * timestep[..., None, None, None]], None, None


def _trainer_worker(rank, world, port, comm, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    D.init("gloo")
    import copy
    from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
    from opendwm_amd import train_ops
    from opendwm_amd.pipeline import CTSDTrainer
    train_ops.adamw_ = _adamw_cpu                      # the HIP kernel needs a GPU; the host logic around it is what runs here
    train_ops.adamw_multi_ = lambda ps, gs, ms, vs, shs, **kw: [_adamw_cpu(*t, **kw) for t in zip(ps, gs, ms, vs, shs)]
    calls = []
    hook0 = default_hooks.bf16_compress_hook

    def counting_hook(state, bucket):
        calls.append(bucket.buffer().numel())
        return hook0(state, bucket)
    default_hooks.bf16_compress_hook = counting_hook
    tc = {"gradient_accumulation_steps": 2, "max_norm_for_grad_clip": 0.5, "freezing_pattern": "^frozen$"}
    mk_sched = lambda opt: torch.optim.lr_scheduler.LambdaLR(opt, lambda s: 1.0 / (1 + s))
    model = _StandInDenoiser()
    ref_model = copy.deepcopy(model)
    tr = CTSDTrainer(model, lr=1e-2, betas=(0.9, 0.95), weight_decay=0.01, ddp=True, training_config=tc, lr_scheduler=mk_sched,
                     ddp_comm_dtype=comm)
    assert tr.frozen_modules == ["frozen"] and len(tr.optimizer.param_groups[0]["params"]) == 5    # frozen ones stay listed

    def data(r, step):
        g = torch.Generator().manual_seed(1000 * r + step)
        return torch.randn(1, 2, 2, 4, 4, 6, generator=g), torch.Generator().manual_seed(77 + 1000 * r + step)

    for step in range(4):
        lat, gen = data(rank, step)
        tr.train_step(lat, {}, generator=gen)
    # expectation, computed locally from BOTH ranks' data: mean over ranks of the gradient summed over the two micro-steps,
    # clipped, torch.optim.AdamW, the LR schedule advanced once per call (ctsd.py:1401-1435)
    rt = CTSDTrainer(ref_model, training_config={"freezing_pattern": "^frozen$"})
    ropt = torch.optim.AdamW(ref_model.parameters(), lr=1e-2, betas=(0.9, 0.95), weight_decay=0.01)
    rsched = mk_sched(ropt)
    for ostep in range(2):
        acc = None
        for r in range(world):
            for micro in range(2):
                lat, gen = data(r, 2 * ostep + micro)
                ref_model.zero_grad()
                rt.loss(lat, {}, generator=gen).backward()
                gs = [None if p.grad is None else p.grad.clone() for p in ref_model.parameters()]
                acc = gs if acc is None else [a if g is None else a + g for a, g in zip(acc, gs)]
        for p, a in zip(ref_model.parameters(), acc):
            p.grad = None if a is None else a / world
        torch.nn.utils.clip_grad_norm_(ref_model.parameters(), 0.5)
        rsched.step()                                  # the micro-step call
        ropt.step()
        ropt.zero_grad()
        rsched.step()
    err = max((p - r_).abs().max().item() for p, r_ in zip(model.parameters(), ref_model.parameters()))
    osd = tr.optimizer.state_dict()
    q.put((rank, err, len(calls), tr.optimizer.lr, ropt.param_groups[0]["lr"], sorted(osd["state"]), tr.optimizer.t,
           [p.detach().flatten().tolist() for p in model.parameters()]))
    D.shutdown()


@pytest.mark.parametrize("comm", [torch.bfloat16, None])
def test_trainer_accumulation_no_sync_bf16_buckets_lr_schedule_two_ranks(comm):
    """CTSDTrainer over DDP on two gloo ranks with a stand-in model: gradient_accumulation_steps = 2 (one bucket all-reduce per
    OPTIMIZER step - the micro-step runs under no_sync), bf16 buckets (bf16_compress_hook) or fp32, gradient clipping, a
    frozen module that stays in the optimizer's parameter list, and an LR scheduler stepped every call."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_trainer_worker, args=(r, 2, port, comm, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, err, ncalls, lr, ref_lr, state_keys, t, params in res:
        assert err < (2e-4 if comm is not None else 1e-6), err       # bf16 buckets round the gradient to 8 bits
        assert ncalls == (2 if comm is not None else 0)               # one bucket, two optimizer steps: NOT four
        assert abs(lr - ref_lr) < 1e-12 and abs(lr - 1e-2 / 5) < 1e-12   # scheduler stepped on all four calls
        assert state_keys == [0, 3, 4] and t == 2                    # frozen parameters 1, 2 (own parameters come first): listed, no state
    assert res[0][-1] == res[1][-1]                  # replicas stay identical


def test_preflight_device_check_is_per_host():
    """16 ranks on 2 nodes x 8 GPUs use every device index twice (once per host): fine; two ranks of one host on one GPU: not"""
    from opendwm_amd import dist as D
    D.check_distinct_devices([(h, i, 1000 + 8 * h + i) for h in (11, 22) for i in range(8)])
    D.check_distinct_devices([(11, 0, 1)])
    with pytest.raises(RuntimeError, match="share a GPU"):
        D.check_distinct_devices([(11, 0, 1), (11, 1, 2), (11, 1, 3), (22, 1, 4)])
    # with the PCI identity (host, pci, index, pid): ranks that each see only their own GPU (index 0 everywhere, different
    # addresses) pass; so do partitions of one GPU (same address, different indices); the same address and index does not
    D.check_distinct_devices([(11, 0x100 * b, 0, 50 + b) for b in range(8)])
    D.check_distinct_devices([(11, 0x500, i, 60 + i) for i in range(4)])
    with pytest.raises(RuntimeError, match="share a GPU"):
        D.check_distinct_devices([(11, 0x500, 0, 1), (11, 0x600, 0, 2), (11, 0x500, 0, 3)])
    # no PCI identity in the build: ranks that see all GPUs of the node are still checked by (host, index); ranks that see one
    # device each cannot be told apart - skipped (False), never an abort of a correct launch
    assert D.check_distinct_devices([(11, -1, i, 70 + i) for i in range(8)], [8] * 8) is True
    with pytest.raises(RuntimeError, match="share a GPU"):
        D.check_distinct_devices([(11, -1, 0, 1), (11, -1, 1, 2), (11, -1, 1, 3)], [8, 8, 8])
    with pytest.warns(UserWarning):
        assert D.check_distinct_devices([(11, -1, 0, 80 + i) for i in range(8)], [1] * 8) is False
    assert D.pci_string(0x00012a00) == "0001:2a:00" and D.pci_string(-1) is None
