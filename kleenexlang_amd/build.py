"""Build recipes for the native pieces (explicit g++ / hipcc command lines, in-tree outputs).

Outputs go to ``kleenexlang_amd/_build`` (git-ignored, but shipped to the GPU box):
  kexc          the compiler CLI                       (g++)
  libkexc.so    the compiler behind include/kexc_api.h (g++)
  libkxhip.so   the HIP engine behind include/kxhip.h  (hipcc --offload-arch=gfx950)
  kxrun         generic host driver that `kexc --out BIN` specialises
Oracle artefacts (test infrastructure) are built by ``oracle/Makefile``.
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "kleenexlang_amd")
CSRC = os.path.join(PKG, "csrc")
OUT = os.path.join(PKG, "_build")

KEXC_SRCS = ["frontend.cpp", "automata.cpp", "lower.cpp", "emit_c.cpp", "compile.cpp", "simulate.cpp"]
CXXFLAGS = ["-O2", "-std=c++17", "-fPIC", "-Wall", "-Wno-sign-compare"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def _deps(dirs):
    out = []
    for d in dirs:
        for r, _, fs in os.walk(d):
            out += [os.path.join(r, f) for f in fs]
    return out


def build_kexc(force=False):
    os.makedirs(OUT, exist_ok=True)
    srcs = [os.path.join(CSRC, "kexc", s) for s in KEXC_SRCS]
    deps = _deps([os.path.join(CSRC, "kexc"), os.path.join(ROOT, "include")])
    lib = os.path.join(OUT, "libkexc.so")
    exe = os.path.join(OUT, "kexc")
    if force or _newer(lib, deps):
        _run(["g++", *CXXFLAGS, "-shared", "-o", lib, *srcs, "-ldl"])
    if force or _newer(exe, deps):
        _run(["g++", *CXXFLAGS, "-o", exe, os.path.join(CSRC, "kexc", "main.cpp"), *srcs, "-ldl"])
    return lib, exe


def build_engine(force=False):
    os.makedirs(OUT, exist_ok=True)
    deps = _deps([os.path.join(CSRC, "engine"), os.path.join(ROOT, "include")])
    lib = os.path.join(OUT, "libkxhip.so")
    gen, inc = os.path.join(CSRC, "engine", "gen_sweeps.py"), os.path.join(CSRC, "engine", "kx_sweeps.inc")
    if not os.path.exists(inc) or os.path.getmtime(gen) > os.path.getmtime(inc):   # hand-scheduled sweeps (generated text)
        with open(inc, "w") as f:
            subprocess.check_call([sys.executable, gen], stdout=f)
    if force or _newer(lib, deps):
        if not os.path.exists(HIPCC):
            raise RuntimeError("hipcc not found at %s" % HIPCC)
        _run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
              "-Wall", "-Wno-unused-result", "-o", lib, os.path.join(CSRC, "engine", "kx_engine.hip"),
              os.path.join(CSRC, "engine", "kx_sharded.cpp"), "-ldl", "-lpthread"])
    drv = os.path.join(OUT, "kxrun")
    if force or _newer(drv, deps):
        _run(["g++", "-O2", "-std=c++17", "-o", drv, os.path.join(CSRC, "engine", "kxrun.cpp"), "-ldl"])
    return lib, drv


def engine_sha():
    """sha256 (16 hex digits) over the engine's sources: stamps profiles/*traffic.json so that bench.py only quotes
    PMC traffic that was measured with the engine build it is running."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(CSRC, "engine")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".inc", ".py", ".h")):     # (the kernels and what generates them; not kxrun.cpp / kx_sharded.cpp, host code)
            with open(os.path.join(d, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    with open(os.path.join(ROOT, "include", "kxp_format.h"), "rb") as f:
        h.update(f.read())
    return h.hexdigest()[:16]


def build_all(force=False):
    build_kexc(force)
    build_engine(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
