"""Build libdwm_hip.so (gfx950) in-tree with hipcc.  No torch / pybind dependency:
the library is a plain C ABI (include/dwm_hip.h) loaded through ctypes.

    python -m opendwm_amd.build [--force]
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libdwm_hip.so")
SOURCES = ["gemm_bf16.hip", "gemm_bf16_4w.hip", "gemm_tn.hip", "attention.hip", "attention_stream.hip", "attention_bwd.hip", "norm.hip",
           "elementwise.hip", "vae.hip", "train.hip", "fp32path.hip"]
# translation units built without -amdgpu-mfma-vgpr-form (accumulators allowed into AGPRs): the 4-wave GEMM keeps the 256 accumulator
# registers of a wave there, the one-wave-per-SIMD attention the output accumulators of up to five query tiles
AGPR_SOURCES: set = {"gemm_bf16_4w.hip", "attention_stream.hip"}
# per-file flags.  attention_stream.hip: the row-sum adds of its tile loop must stay scalar (left alone the SLP vectoriser packs the adds
# of two slices into v_pk_add_f32 bunched behind the later one; packed fp32 VALU beside MFMAs costs more than the adds it replaces)
FILE_FLAGS: dict = {"attention_stream.hip": ["-fno-slp-vectorize"]}
# -amdgpu-mfma-vgpr-form: keep MFMA accumulators in arch VGPRs (gfx950's unified file) so the VALU epilogues / softmax read them
# without v_accvgpr_read/write copies
VGPR_FORM = ["-mllvm", "-amdgpu-mfma-vgpr-form"]
ARCH = "gfx950"


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm to build libdwm_hip.so)")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def source_hash() -> str:
    """SHA-256 over the HIP sources and headers (sorted by name); baked into the library as dwm_source_hash()"""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))) + [os.path.join(INCLUDE, "dwm_hip.h")]
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = False) -> str:
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "attention_common.h"), os.path.join(INCLUDE, "dwm_hip.h")]
    shash = source_hash()
    stamp = os.path.join(CSRC, "build", ".source_hash")
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    base = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", f"-I{INCLUDE}", f"-I{CSRC}"]
    user = os.environ.get("DWM_EXTRA_FLAGS", "").split()
    jobs = []
    # the stamp holds the source hash and the extra flags of the objects on disk: other flags (e.g. -DDWM_DEV_HOOKS) change
    # neither the hash nor a modification time, so they force a full rebuild here
    extra = " ".join(os.environ.get("DWM_EXTRA_FLAGS", "").split())
    old_hash, old_extra = "", ""
    if os.path.exists(stamp):
        lines = open(stamp).read().split("\n")          # two lines: the hash, the flags (any character but a newline)
        old_hash, old_extra = lines[0].strip(), (lines[1].strip() if len(lines) > 1 else "")
    force = force or old_extra != extra
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        # elementwise.hip carries the source hash: it is rebuilt whenever any source changed
        if force or _stale(o, [s] + headers) or (src == "elementwise.hip" and old_hash != shash):
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        name = os.path.basename(s)
        fl = base + ([] if name in AGPR_SOURCES else VGPR_FORM) + FILE_FLAGS.get(name, []) + user
        if name == "elementwise.hip":
            fl = fl + [f'-DDWM_SOURCE_HASH="{shash}"']
        cmd = [hipcc] + fl + ["-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s}:\n{r.stderr}")
        if verbose and r.stderr:
            print(r.stderr, file=sys.stderr)
        return o

    if jobs:
        with ThreadPoolExecutor(max_workers=len(jobs)) as ex:
            list(ex.map(compile_one, jobs))
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
    with open(stamp, "w") as f:
        f.write(shash + "\n" + extra)
    return LIB


def ensure_built() -> str:
    """Build the library if it is missing (sources newer than the .so do NOT trigger a rebuild here: a snapshot copy may
    reorder mtimes; a stale binary is caught by _lib.load(), which compares dwm_source_hash() with the sources).  Serialised
    with a file lock so the ranks of one node do not compile concurrently."""
    if os.path.exists(LIB):
        return LIB
    import fcntl
    with open(os.path.join(HERE, ".build.lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        try:
            if not os.path.exists(LIB):
                build()
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
