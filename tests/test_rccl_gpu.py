"""GPU leg: the RCCL code paths on the ONE GPU of the test box (backend "nccl" = RCCL on ROCm, world size 1).

Multi-GPU runs are the driver's; what can be done here is to make sure that none of the collective code is unexecuted
code: a one-rank RCCL communicator carries the DDP gradient all-reduce of the train step (ctsd.py:1051-1054), the device
`all_to_all_single` of the frame shards (opendwm_amd/sharding.py; under gloo the tensors are staged through the host instead),
the latents all-gather, and the preflight all-reduce of bench.py.  With one rank every collective is the identity, so the
results must equal the collective-free path bit for bit."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

from oracle import ctsd_oracle as O
from tests.common import small_config, small_inputs, to_dev

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
bf16 = torch.bfloat16


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _log(name, **kv):
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "gpu_parity.log"), "a") as f:
        f.write(json.dumps({"test": name, **kv}) + "\n")
    print(name, kv)


@pytest.fixture(scope="module")
def rccl():
    """a one-rank process group over RCCL on cuda:0, torn down after the module"""
    if not torch.cuda.is_available():
        pytest.fail("the gpu-marked tests need a HIP device (torch.cuda.is_available() is False)")
    import torch.distributed as dist
    from opendwm_amd import _lib
    _lib.load()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    assert not dist.is_initialized()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1, device_id=dev)
    yield dev
    dist.destroy_process_group()


def test_preflight_and_allreduce_over_rccl(rccl):
    import torch.distributed as dist
    from opendwm_amd import dist as D
    assert dist.get_backend() == "nccl"
    pre = D.preflight(rccl, mbytes=16)
    _log("rccl_preflight", **{k: v for k, v in pre.items()})
    assert pre["backend"] == "nccl" and pre["rccl_version"] and pre["allreduce_sum_ok"] and pre["distinct_devices"] and pre["hosts"] == 1
    assert D.measure_allreduce(64 << 20, rccl, torch.float32) > 0.0
    assert D.max_over_ranks(1.25, rccl) == 1.25


def _train_model(cfg, sd, dev):
    from opendwm_amd.dit import DiTCrossviewTemporalConditionModel
    m = DiTCrossviewTemporalConditionModel(**cfg)
    m.load_state_dict(sd)
    return m.to(dev).train()


def test_ddp_train_step_over_rccl_equals_plain_step(rccl):
    """CTSDTrainer(ddp=True) on a one-rank RCCL group: DistributedDataParallel's reducer hooks fire per block Function, the
    buckets are all-reduced on the device by RCCL (mean over one rank = identity), AdamW steps on the bucket views - the losses
    and the parameters after two optimizer steps must equal the plain trainer's (bit for bit wherever the backward kernels
    are order-deterministic)"""
    from opendwm_amd.pipeline import CTSDTrainer
    dev = rccl
    cfg = small_config()
    sd = {k: v.to(bf16).float() for k, v in O.make_state_dict(cfg, 0).items()}
    inp = small_inputs(cfg, 0)
    inp = {k: (v.to(bf16).float() if v.is_floating_point() and k not in ("timestep", "added_time_ids") else v) for k, v in inp.items()}
    lat = inp.pop("sample").to(dev)
    inp.pop("timestep")
    noise = torch.randn(lat.shape, generator=torch.Generator().manual_seed(5))
    idx = torch.tensor([250, 800])
    results = {}
    for name, ddp in (("plain", False), ("ddp", True)):
        m = _train_model(cfg, sd, dev)
        tr = CTSDTrainer(m, lr=2e-4, weight_decay=0.01, ddp=ddp)
        di = to_dev(inp, dev)
        losses = [tr.train_step(lat, di, timestep_indices=idx, noise=noise).item() for _ in range(2)]
        if ddp:
            assert isinstance(tr.wrapper, torch.nn.parallel.DistributedDataParallel)
            assert tr.wrapper.process_group is not None and torch.distributed.get_backend(tr.wrapper.process_group) == "nccl"
        results[name] = (losses, {n: p.detach().clone() for n, p in m.named_parameters()})
        del tr, m
    lp, pp = results["plain"]
    ld, pd = results["ddp"]
    diff = [n for n in pp if not torch.equal(pp[n], pd[n])]
    worst = max(((pp[n].double() - pd[n].double()).norm() / pp[n].double().norm().clamp_min(1e-30)).item() for n in pp)
    _log("ddp_over_rccl_one_rank", losses_plain=lp, losses_ddp=ld, params=len(pp), params_different=len(diff), worst_rel=worst)
    # bit-identical except where the backward itself is not: bias / modulation gradients are segmented column sums finished by fp32
    # atomics (train.hip), whose order changes from run to run - those parameters (8 and 17 of the 240 in two runs of round 4)
    # agree to accumulation accuracy (7.5e-9 observed); what the test holds is the losses, the accuracy, and that it is only those
    assert lp == ld and len(diff) <= len(pp) // 4 and worst < 1e-6, (diff[:5], worst)
    assert all(n.endswith(("bias", "weight")) for n in diff)


@pytest.mark.parametrize("temporal,mode", [("rowwise", "full"), ("pointwise", "diffusion_forcing")])
def test_frame_shard_over_rccl_equals_unsharded(rccl, temporal, mode):
    """FrameShard with R = 1 over RCCL: the re-sharding around every temporal block goes through the DEVICE
    all_to_all_single (sharding.all_to_all_chunks' RCCL branch) and result() through the device all-gather; one rank holds
    all frames, so the latents must equal the unsharded denoiser's bit for bit"""
    import torch.distributed as dist
    from opendwm_amd import sharding
    from opendwm_amd.dit import DiTCrossviewTemporalConditionModel
    from opendwm_amd.pipeline import CTSDDenoiser
    dev = rccl
    cfg = small_config(temporal_attention_type=temporal)
    sd = {k: v.to(bf16).float() for k, v in O.make_state_dict(cfg, 0).items()}

    def model():
        m = DiTCrossviewTemporalConditionModel(**cfg)
        m.load_state_dict(sd, strict=True)
        return m.to(dev).to(bf16).eval()
    inp = small_inputs(cfg, 0, T=4)
    cond = {k: v for k, v in inp.items() if k not in ("sample", "timestep")}
    lat = torch.randn(1, 4, 3, 16, 8, 12, generator=torch.Generator().manual_seed(13))
    img = torch.randn(1, 4, 3, 16, 8, 12, generator=torch.Generator().manual_seed(14))
    kw = {"full": {}, "diffusion_forcing": dict(image_latents=img, diffusion_forcing=True, take_time=0)}[mode]
    kwd = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in kw.items()}
    calls = {"n": 0}
    orig = sharding.dist.all_to_all_single

    def counting(*a, **k):
        assert a[0].is_cuda and a[1].is_cuda            # device tensors straight into RCCL: no host staging
        calls["n"] += 1
        return orig(*a, **k)
    sharding.dist.all_to_all_single = counting
    try:
        single = CTSDDenoiser(model(), guidance_scale=4.0, inference_steps=4).run(lat.to(dev), to_dev(cond, dev), stop=3, **kwd)
        assert calls["n"] == 0
        sharded = CTSDDenoiser(model(), guidance_scale=4.0, inference_steps=4, frame_group=dist.group.WORLD).run(
            lat.to(dev), to_dev(cond, dev), stop=3, **kwd)
    finally:
        sharding.dist.all_to_all_single = orig
    _log("frame_shard_over_rccl_one_rank", temporal=temporal, mode=mode, all_to_all_calls=calls["n"], equal=bool(torch.equal(single, sharded)))
    assert calls["n"] > 0 and torch.equal(single, sharded)


def _run_bench(*flags):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", *flags], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_preflight_line_over_rccl():
    """`bench.py --gpus 1 --preflight`: the line carries the RCCL facts of a checked one-rank all-reduce (the same code a
    torch.distributed.run launch executes before its timed region)"""
    line = _run_bench("--preflight", "--steps", "1", "--warmup", "1", "--layers", "2", "--no-cpu-baseline", "--no-text-only-leg")
    pre = line["preflight"]
    _log("bench_preflight_rccl", **pre)
    assert pre["backend"] == "nccl" and pre["rccl_version"] and pre["allreduce_sum_ok"] and pre["world_size"] == 1
    assert line["n_gpus"] == 1 and line["config"]["finite"]


def test_bench_train_ddp_fields_over_rccl():
    """`bench.py --train --gpus 1 --preflight`: the train step under DistributedDataParallel on a one-rank RCCL group, with the
    gradient-exchange probes (`config.ddp`) populated"""
    line = _run_bench("--train", "--preflight", "--steps", "1", "--warmup", "1", "--layers", "2")
    ddp = line["config"]["ddp"]
    _log("bench_train_ddp_rccl", **ddp)
    assert ddp is not None and ddp["allreduce_bytes"] > 0 and ddp["allreduce_ms"] > 0.0
    assert ddp["forward_backward_ms_with_gradient_sync"] > 0.0 and ddp["forward_backward_ms_without_gradient_sync"] > 0.0
    assert line["config"]["preflight"]["backend"] == "nccl" and line["config"]["finite"]
    # the bucket timeline behind the modelled 8-GPU exchange: every bucket stamped inside the backward, all gradient elements covered
    tl = ddp["bucket_timeline"]
    assert "error" not in tl, tl
    assert tl["buckets"] >= 1 and tl["elements"] * 4 == ddp["allreduce_bytes"]
    assert all(0.0 <= t <= tl["backward_ms"] * 1.05 for t in tl["ready_ms_after_backward_start"]), tl
    assert tl["fp32_ring_2_links"]["exposed_ms"] >= tl["bf16_ring_2_links"]["exposed_ms"] >= 0.0
