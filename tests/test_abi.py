"""CPU: the C-ABI libraries load and export every symbol their headers declare (no compute)."""
import ctypes
import os
import re
import subprocess

from kleenexlang_amd import build

INC = os.path.join(build.ROOT, "include")


def _declared(header):
    txt = open(os.path.join(INC, header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(k(?:x|exc)_[a-z_0-9]+)\s*\(", txt)))


def test_kxhip_exports_every_declared_symbol():
    lib = ctypes.CDLL(os.path.join(build.OUT, "libkxhip.so"))
    names = _declared("kxhip.h")
    assert len(names) >= 17, names
    for n in names:
        assert hasattr(lib, n), "libkxhip.so lacks %s" % n


def test_kexc_exports_every_declared_symbol():
    lib = ctypes.CDLL(os.path.join(build.OUT, "libkexc.so"))
    names = _declared("kexc_api.h")
    assert set(names) == {"kexc_compile", "kexc_emit_c", "kexc_dump_fst", "kexc_last_error", "kexc_free"}
    for n in names:
        assert hasattr(lib, n)


def test_engine_is_gfx950_code_object():
    so = os.path.join(build.OUT, "libkxhip.so")
    out = subprocess.run(["strings", "-a", so], stdout=subprocess.PIPE, check=True).stdout
    assert b"gfx950" in out


def test_engine_refuses_to_run_without_a_gpu():
    """No CPU fallback: without a HIP device kx_load fails loudly (skipped on the GPU box)."""
    import torch
    import pytest
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from kleenexlang_amd import EngineError, Program, compile_file
    with pytest.raises(EngineError, match="no HIP device|HIP"):
        Program(compile_file("flip_ab"))
