"""world_size-2 gloo test of the replica launch path (runs on CPU): rank/env parsing, sample
sharding, barrier-bracketed timing with MAX-over-ranks, and that each rank's denoise loop is an
independent replica (the oracle stands in for the device kernels here)."""
import os
import socket

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
