"""The drop-in classes as diffusers types ("missing" item 4).

The reference switches on `isinstance(model, diffusers.UNetSpatioTemporalConditionModel)` / `isinstance(model,
diffusers.SD3Transformer2DModel)` (src/dwm/pipelines/ctsd.py:186,205,450,888,896,967,977,1240,1255,1308-1311,1359), so the
drop-in classes derive from those types WHEN `diffusers` imports (opendwm_amd/dit.py, unet.py: `_Base`) and bypass their
constructors (`nn.Module.__init__`: diffusers' own __init__ would build its own blocks).  `diffusers` is not installed in this
image, so that branch never runs in the rest of the suite.  Here it runs in a fresh interpreter behind a stand-in `diffusers`
package whose model types behave like diffusers 0.31.0's where it matters for this path:

  * `ModelMixin(torch.nn.Module)` with the `__getattr__` that looks attributes up in `_internal_dict` first (so a subclass
    that never ran `register_to_config` must not trip over a missing `_internal_dict`), `ConfigMixin.config` as a property;
  * `SD3Transformer2DModel.__init__` / `UNetSpatioTemporalConditionModel.__init__` RAISE: the drop-in must not call them.

Checked: the isinstance switches of ctsd.py (restated in their if / elif order) take the SD 3 branch for the DiT class and the
UNet branch for the UNet class; the state-dict keys are the ones the classes have WITHOUT diffusers (= the reference's, checked
in test_oracle_cpu.py); `.config.patch_size` (crossview_temporal_dit.py:409), `enable_gradient_checkpointing()` (ctsd.py:868),
`.to(dtype=...)` (:867), `named_modules()` / `parameters()` / `load_state_dict` (:881-1029) work on the derived class.
What stays unexercised: the real ModelMixin (save_pretrained / from_pretrained / hub plumbing), which this path does not call.
"""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STUB = '''
import torch


class ConfigMixin:
    config_name = "config.json"

    @property
    def config(self):
        return self._internal_dict


class ModelMixin(torch.nn.Module):
    _supports_gradient_checkpointing = False

    def __init__(self):
        super().__init__()

    def __getattr__(self, name):                       # diffusers 0.31.0 modeling_utils.ModelMixin.__getattr__
        is_in_config = "_internal_dict" in self.__dict__ and hasattr(self.__dict__["_internal_dict"], name)
        is_attribute = name in self.__dict__
        if is_in_config and not is_attribute:
            return self._internal_dict[name]
        return super().__getattr__(name)

    def enable_gradient_checkpointing(self):
        if not self._supports_gradient_checkpointing:
            raise ValueError(f"{self.__class__.__name__} does not support gradient checkpointing.")


class SD3Transformer2DModel(ModelMixin, ConfigMixin):
    _supports_gradient_checkpointing = True

    def __init__(self, *a, **kw):
        raise AssertionError("the drop-in must not run diffusers.SD3Transformer2DModel.__init__ (it builds its own blocks)")


class UNetSpatioTemporalConditionModel(ModelMixin, ConfigMixin):
    _supports_gradient_checkpointing = True

    def __init__(self, *a, **kw):
        raise AssertionError("the drop-in must not run diffusers.UNetSpatioTemporalConditionModel.__init__")
'''

CHILD = '''
import json, sys, torch
import diffusers
from tests.common import small_config
from tests.test_oracle_cpu import _unet_small
from opendwm_amd import dit, unet

def branch(model):                                     # ctsd.py:186 / 205 (the same order at every switch site)
    if isinstance(model, diffusers.UNetSpatioTemporalConditionModel):
        return "unet"
    elif isinstance(model, diffusers.SD3Transformer2DModel):
        return "sd3"
    return "none"

out = {}
cfg = small_config()
m = dit.DiTCrossviewTemporalConditionModel(**cfg)
u = unet.UNetCrossviewTemporalConditionModel(**_unet_small())
out["bases"] = [dit._Base.__module__ + "." + dit._Base.__name__, unet._Base.__module__ + "." + unet._Base.__name__]
out["branch"] = [branch(m), branch(u)]
out["is_module"] = [isinstance(m, torch.nn.Module), isinstance(u, torch.nn.Module)]
out["patch_size"] = m.config.patch_size
m.enable_gradient_checkpointing(); u.enable_gradient_checkpointing()
out["ckpt"] = [bool(m.gradient_checkpointing), bool(u.gradient_checkpointing)]
m2 = m.to(dtype=torch.bfloat16)
out["to_dtype"] = str(next(m2.parameters()).dtype)
out["dit_keys"] = sorted(m.state_dict().keys())
out["unet_keys"] = sorted(u.state_dict().keys())
sd = {k: v.clone() for k, v in m.state_dict().items()}
missing, unexpected = m.load_state_dict(sd)
out["load"] = [list(missing), list(unexpected)]
out["n_modules"] = [sum(1 for _ in m.named_modules()), sum(1 for _ in u.named_modules())]
out["depth_net"] = [getattr(m, "depth_net", "absent") is None, u.depth_net is None]
try:                                                   # an unknown attribute still raises AttributeError (not a KeyError on _internal_dict)
    m.no_such_attribute
    out["missing_attr"] = "no error"
except AttributeError:
    out["missing_attr"] = "AttributeError"
print("DROPIN " + json.dumps(out))
'''


def test_dropin_classes_are_diffusers_types_when_diffusers_imports(tmp_path, small_cfg):
    pkg = tmp_path / "diffusers"
    pkg.mkdir()
    (pkg / "__init__.py").write_text(textwrap.dedent(STUB))
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([str(tmp_path), ROOT]))
    r = subprocess.run([sys.executable, "-c", textwrap.dedent(CHILD)], env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("DROPIN ")][0][7:])
    assert out["bases"] == ["diffusers.SD3Transformer2DModel", "diffusers.UNetSpatioTemporalConditionModel"]
    assert out["branch"] == ["sd3", "unet"] and out["is_module"] == [True, True]
    assert out["patch_size"] == small_cfg["patch_size"] and out["ckpt"] == [True, True]
    assert out["to_dtype"] == "torch.bfloat16" and out["load"] == [[], []] and out["missing_attr"] == "AttributeError"
    assert out["depth_net"] == [True, True]            # ctsd.py:1462 reads model.depth_net on both model families
    # the same keys as without diffusers (this process: plain nn.Module base)
    from opendwm_amd import dit, unet
    from tests.test_oracle_cpu import _unet_small
    assert dit._Base is __import__("torch").nn.Module
    assert out["dit_keys"] == sorted(dit.DiTCrossviewTemporalConditionModel(**small_cfg).state_dict().keys())
    assert out["unet_keys"] == sorted(unet.UNetCrossviewTemporalConditionModel(**_unet_small()).state_dict().keys())


def test_kernel_selection_is_mapped_on_the_host_side(monkeypatch):
    """The C library reads no environment (round 6): `opendwm_amd.ops` maps DWM_GEMM4W / gemm_4wave_scope to dwm_gemm_args.tile
    (0 = library's choice on the 8-wave kernels, 3 = the 4-wave kernels may serve the launch, 4 = their fast form only; an explicit
    tile 1 / 2 of the caller always wins) and DWM_ATTN_VARIANT / DWM_ATTN_RES4 to bits OR-ed into dwm_attn_args.variant."""
    import importlib
    from opendwm_amd import ops
    cases = {"": (0, 3), "0": (0, 0), "1": (3, 3), "f": (0, 4)}               # env -> (outside a scope, inside gemm_4wave_scope(True))
    for env, (outside, inside) in cases.items():
        monkeypatch.setattr(ops, "_G4W_ENV", env)
        assert ops._gemm_tile_request(0) == outside, env
        with ops.gemm_4wave_scope(True):
            assert ops._gemm_tile_request(0) == inside, env
            assert ops._gemm_tile_request(1) == 1 and ops._gemm_tile_request(2) == 2
            with ops.gemm_4wave_scope(False):
                assert ops._gemm_tile_request(0) == (3 if env == "1" else 0), env
        assert ops._gemm_tile_request(0) == outside
    assert ops.ATTN_Q_PRESCALED == 1 << 15 and ops.ATTN_STREAM == 1 << 12 and ops.ATTN_RES12 == 1 << 13
    for env, want in ((dict(DWM_ATTN_VARIANT="0x2000"), 1 << 13), (dict(DWM_ATTN_VARIANT="4096", DWM_ATTN_RES4=""), 1 << 12),
                      (dict(DWM_ATTN_RES4="2"), (1 << 12) | (1 << 13)), ({}, 0)):
        for k in ("DWM_ATTN_VARIANT", "DWM_ATTN_RES4"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        assert importlib.reload(ops)._ATTN_ENV_VARIANT == want, env
    for k in ("DWM_ATTN_VARIANT", "DWM_ATTN_RES4"):
        monkeypatch.delenv(k, raising=False)
    importlib.reload(ops)
