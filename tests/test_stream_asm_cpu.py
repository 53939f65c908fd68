"""attention_stream.hip issues its S MFMAs from inline asm, which the compiler's hazard recogniser does not see; what it would have
enforced is checked on the GENERATED code by scripts/dev/check_stream_asm.py (no instruction touches an asm MFMA's result within the 11
wait states an 8-pass MFMA needs, destinations never overlap the operands).  This test compiles the file for gfx950 exactly as
opendwm_amd/build.py does (hipcc cross-compiles without a GPU) and runs the checker, so a change of the source or of the compiler
that moves an instruction into such a window fails here, not as a wrong number on the GPU."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_stream_inline_asm_mfma_hazards_on_generated_code(tmp_path):
    from opendwm_amd import build as B
    try:
        hipcc = B._hipcc()
    except RuntimeError:
        pytest.skip("hipcc not available")
    name = "attention_stream.hip"
    flags = [f"--offload-arch={B.ARCH}", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", f"-I{B.INCLUDE}", f"-I{B.CSRC}"] + \
        ([] if name in B.AGPR_SOURCES else B.VGPR_FORM) + B.FILE_FLAGS.get(name, [])
    r = subprocess.run([hipcc] + flags + ["-c", os.path.join(B.CSRC, name), "-o", str(tmp_path / "stream.o"), "-save-temps=obj", "-Wno-inline-asm"],
                       capture_output=True, text=True, cwd=str(tmp_path), timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    asm = [f for f in os.listdir(tmp_path) if f.endswith(".s") and "gfx950" in f]
    assert len(asm) == 1, os.listdir(tmp_path)
    c = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "dev", "check_stream_asm.py"), str(tmp_path / asm[0])],
                       capture_output=True, text=True, timeout=300)
    assert c.returncode == 0, c.stdout[-3000:]
    # the kernel was found, with the S chains of every tile count (2..5 tiles per wave)
    first = c.stdout.splitlines()[0]
    assert first.startswith("1 kernels") and " 0 problems" in first, c.stdout[:500]
    assert int(first.split(",")[1].split()[0]) >= 300, first
    shutil.rmtree(tmp_path, ignore_errors=True)
