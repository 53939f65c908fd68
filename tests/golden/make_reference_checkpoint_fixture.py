"""Checkpoint files written by EXECUTING the reference's save path - runs only where /root/reference exists.

The REAL CrossviewTemporalSD.save_checkpoint (src/dwm/pipelines/ctsd.py:1134-1155) with the REAL
dwm.distributed.distributed_save_optimizer_state (src/dwm/distributed.py:7-40) on a two-layer stand-in model after three
torch.optim.AdamW steps; the committed files are what `CTSDTrainer.load_checkpoint` must be able to resume from
(tests/test_reference_fixtures_cpu.py), and `reference_checkpoint/expected.pt` holds the parameters after one more step.

usage: python tests/golden/make_reference_checkpoint_fixture.py  ->  tests/golden/reference_checkpoint/
"""
import os
import shutil
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests.golden.make_reference_driver_fixtures import _Finder          # noqa: E402


def tiny_model():
    torch.manual_seed(3)
    return torch.nn.Sequential(torch.nn.Linear(4, 8), torch.nn.SiLU(), torch.nn.Linear(8, 4))


def batch(i):
    g = torch.Generator().manual_seed(100 + i)
    return torch.randn(5, 4, generator=g), torch.randn(5, 4, generator=g)


FREEZING_PATTERN = "^0$"        # the first Linear of tiny_model(): a training_config["freezing_pattern"]


def main():
    sys.meta_path.insert(0, _Finder())
    sys.path.insert(0, "/root/reference/src")
    import dwm.pipelines.ctsd as C
    make(C, "reference_checkpoint", None)
    # the warm-up configs freeze modules (ctsd.py:1014-1022) BEFORE the optimizer is built from all parameters (:1089-1092):
    # the saved state is sparse over the full parameter list
    make(C, "reference_checkpoint_frozen", FREEZING_PATTERN)


def make(C, name, freezing_pattern):
    import re
    out = os.path.join(HERE, name)
    shutil.rmtree(out, ignore_errors=True)
    p = object.__new__(C.CrossviewTemporalSD)
    p.model = tiny_model()
    p.model_wrapper = p.model
    p.should_save = True
    if freezing_pattern is not None:          # the loop of ctsd.py:1014-1022
        pattern = re.compile(freezing_pattern)
        for mname, module in p.model.named_modules():
            if pattern.match(mname) is not None:
                module.requires_grad_(False)
    p.optimizer = torch.optim.AdamW(p.model_wrapper.parameters(), lr=1e-2, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.05)
    for i in range(3):
        x, y = batch(i)
        torch.nn.functional.mse_loss(p.model(x), y).backward()
        p.optimizer.step()
        p.optimizer.zero_grad()
    C.CrossviewTemporalSD.save_checkpoint(p, out, 3)
    x, y = batch(3)                                   # one more step: what a resumed run must reproduce
    torch.nn.functional.mse_loss(p.model(x), y).backward()
    grads = [None if q.grad is None else q.grad.clone() for q in p.model.parameters()]
    p.optimizer.step()
    torch.save({"params_after_step_4": [q.detach().clone() for q in p.model.parameters()], "grads_step_4": grads},
               os.path.join(out, "expected.pt"))
    print("wrote", sorted(os.path.relpath(os.path.join(d, f), out) for d, _, fs in os.walk(out) for f in fs))


if __name__ == "__main__":
    main()
