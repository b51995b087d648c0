import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Host-side libraries must exist (cheap no-op when up to date; hipcc cross-compiles without a GPU)."""
    from kleenexlang_amd import build
    build.build_kexc()
    if not os.path.exists(os.path.join(build.OUT, "libkxhip.so")):
        build.build_engine()
    from oracle import oracle
    oracle._lib()


@pytest.fixture(scope="session")
def vectors():
    with open(os.path.join(GOLDEN, "reference_vectors.json"), encoding="utf-8") as f:
        return json.load(f)


@pytest.fixture(scope="session")
def expected():
    with open(os.path.join(GOLDEN, "expected.json")) as f:
        return json.load(f)


_BLOBS = {}


def blob_of(name_or_source, opt=3):
    """Compile (cached) a workload program by name or an inline Kleenex source."""
    from kleenexlang_amd import compile_file, compile_source
    key = (name_or_source, opt)
    if key not in _BLOBS:
        if ":=" in name_or_source:
            _BLOBS[key] = compile_source(name_or_source, opt=opt)
        else:
            _BLOBS[key] = compile_file(name_or_source, opt=opt)
    return _BLOBS[key]


def line_input(lines):
    """What test_compiled/runtest.sh pipes in: `echo "$input"` of the newline-joined IN lines."""
    return ("\n".join(lines) + "\n").encode("utf-8")


def line_expected(lines):
    return "\n".join(lines).encode("utf-8")


def same_modulo_trailing_newlines(got, want):
    """runtest.sh compares `$(...)` captures, which drop trailing newlines."""
    return got.rstrip(b"\n") == want.rstrip(b"\n")


def dictionary_program(nwords=100, seed=5, lo=8, hi=15):
    """A word-for-word rewriter in the shape of the reference's bench/kleenex/src/make_danish.kex (dictionary words
    replaced, every other word copied), generated here: ~1000 states x 28 classes, i.e. a state table of > 100 KiB —
    beyond 16-bit LDS addressing, so the engine runs it from global memory (DevTables::big).  Returns (source, words)."""
    import random
    rnd = random.Random(seed)
    words = set()
    while len(words) < nwords:
        words.add("".join(rnd.choice("abcdefghijklmnopqrstuvwxyz") for _ in range(rnd.randint(lo, hi))))
    words = sorted(words)
    alts = "\n      | ".join('~/%s/ "%s"' % (w, w[::-1].upper() + str(i)) for i, w in enumerate(words))
    return 'main := word sep main | word | ""\nsep := /[^a-z0-9]+/\nword := %s\n      | /[a-z0-9]+/\n' % alts, words
