"""scripts/pin_with_diffusers.py is the one command that closes SURVEY.md §8c's "parity unpinned for the diffusers leaves" on a box that
has diffusers 0.31.0 - and it cannot run in this image (no diffusers).  So that it does not rot unseen, this test runs it in a fresh
interpreter behind a stand-in `diffusers` package whose classes accept any constructor arguments, take `.to()` / `load_state_dict()`
(recording that they got keys) and raise `StandInReached` the moment arithmetic is asked of them: every check must get exactly that
far - through the oracle imports, the small configurations, the seeded state dicts and their key slices - or be skipped for a missing
import (`dwm`, the reference checkout, is not on the path here).  Any other exception is a defect of the script."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STUB_INIT = '''
import sys, types
__version__ = "0.31.0"


class StandInReached(RuntimeError):
    pass


class _StandIn:
    """accepts any construction; refuses arithmetic"""
    def __init__(self, *a, **k):
        self.loaded = None
        for name in ("alphas_cumprod", "final_alpha_cumprod", "sigmas", "timesteps"):
            setattr(self, name, None)

    def to(self, *a, **k):
        return self

    def eval(self):
        return self

    def load_state_dict(self, sd, strict=True):
        assert len(sd) > 0, "an empty state-dict slice was handed to a module (wrong prefix?)"
        self.loaded = list(sd)
        return [], []

    def __call__(self, *a, **k):
        raise StandInReached(type(self).__name__)

    def __getattr__(self, name):                       # any method (encode, decode, step, set_timesteps, add_noise, ...)
        if name.startswith("__"):
            raise AttributeError(name)
        def method(*a, **k):
            raise StandInReached(f"{type(self).__name__}.{name}")
        return method


def _cls(name):
    return type(name, (_StandIn,), {})


def __getattr__(name):
    if name.startswith("__"):
        raise AttributeError(name)
    return _cls(name)


class _Mod(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _cls(name)


for _m in ("diffusers.models", "diffusers.models.attention", "diffusers.models.embeddings", "diffusers.models.normalization",
           "diffusers.schedulers"):
    sys.modules[_m] = _Mod(_m)
'''


def test_pin_script_runs_up_to_the_first_diffusers_call(tmp_path):
    pkg = tmp_path / "diffusers"
    pkg.mkdir()
    (pkg / "__init__.py").write_text(textwrap.dedent(STUB_INIT))
    out = tmp_path / "pin.json"
    env = dict(os.environ, PYTHONPATH=str(tmp_path) + os.pathsep + ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "pin_with_diffusers.py"), "--out", str(out)], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 1, (r.returncode, r.stdout[-1500:], r.stderr[-1500:])       # "failed" checks: the stand-in refused arithmetic
    res = json.load(open(out))
    assert res["diffusers"] == "0.31.0" and len(res["results"]) >= 9
    reached = 0
    for c in res["results"]:
        if c["status"] == "skipped":
            assert "dwm" in c["reason"], c                 # only the reference checkout may be missing
            continue
        assert c["status"] == "ERROR" and c["reason"].startswith("StandInReached"), c
        reached += 1
    assert reached >= 7, res["results"]
