"""ctypes wrapper of the CPU oracle (oracle/kx_oracle.c).  TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — never from
the product package (kleenexlang_amd/), which has no CPU execution path at all.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class OracleMatchError(Exception):
    def __init__(self, pos, stage, partial):
        super().__init__("Match error at input symbol %d!" % pos)
        self.pos, self.stage, self.partial = pos, stage, partial


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    if os.path.isdir("/root/reference/crt"):  # reference runtime only travels as a built binary
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_build", "libkxoracle.so")
        if not os.path.exists(path):
            build()
        lib = ctypes.CDLL(path)
        args = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t,
                ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t),
                ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint32)]
        lib.kxo_run.argtypes = args
        lib.kxo_run_path.argtypes = args
        lib.kxo_free.argtypes = [ctypes.c_void_p]
        lib.kxo_info.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
        _LIB = lib
    return _LIB


def run(blob, data, path_form=False):
    """Execute a KXP blob on bytes with the sequential CPU restatement.

    path_form=False: register form, the way generated C + crt.c run it (the parity oracle);
    path_form=True : sequential evaluation of the path form (cross-check of the compiler)."""
    lib = _lib()
    out = ctypes.c_void_p()
    n = ctypes.c_size_t()
    fp = ctypes.c_uint64()
    fs = ctypes.c_uint32()
    data = bytes(data)
    fn = lib.kxo_run_path if path_form else lib.kxo_run
    rc = fn(blob, len(blob), data, len(data), ctypes.byref(out), ctypes.byref(n), ctypes.byref(fp), ctypes.byref(fs))
    try:
        res = ctypes.string_at(out, n.value) if out.value else b""
    finally:
        if out.value:
            lib.kxo_free(out)
    if rc == 1:
        raise OracleMatchError(fp.value, fs.value, res)
    if rc:
        raise RuntimeError("oracle: malformed program blob")
    return res


def info(blob, stage=0):
    v = (ctypes.c_uint32 * 8)()
    if _lib().kxo_info(blob, len(blob), stage, v):
        raise RuntimeError("oracle: malformed program blob")
    return dict(zip("nstates nclasses nregs nactions maxleaves nback nsync sync_complete".split(), list(v)))


def ref_binary(program, opt=3):
    """Path of the prebuilt generated-C + reference-crt.c binary (oracle/_ref), or None."""
    p = os.path.join(_HERE, "_ref", "%s_opt%d" % (program, opt))
    return p if os.path.exists(p) else None
