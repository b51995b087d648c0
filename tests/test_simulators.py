"""CPU: `kexc simulate --sim lockstep|backtrack` — the reference's two FST simulators restated in the compiler
(csrc/kexc/simulate.cpp; SymbolicFST.hs:243-262,361-380,407-429; Commands.hs:277-302), the user-visible oracle of the language.
They run the nondeterministic transducer and share nothing with determinization, lowering, the path form or the engine, so
the reference's own `test_simulated` check — every program × every simulator gives the `// OUT:` lines — is an independent
pin; `--sim sst` (the compiled program on the HIP engine) joins them in tests/test_engine_gpu.py."""
import json
import os
import random
import subprocess

import pytest
import randprog
from conftest import GOLDEN, blob_of, line_expected, line_input, same_modulo_trailing_newlines

from kleenexlang_amd import build, program_path, workloads
from oracle import oracle

SIMS = ["lockstep", "backtrack"]


def simulate(sim, program=None, source=None, data=b"", tmp=None, extra=()):
    kexc = os.path.join(build.OUT, "kexc")
    if source is not None:
        path = os.path.join(str(tmp), "p%08x.kex" % (hash(source) & 0xFFFFFFFF))
        with open(path, "w", encoding="utf-8") as f:
            f.write(source)
    else:
        path = program
    return subprocess.run([kexc, "simulate", "--sim", sim, *extra, path], input=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE)


@pytest.mark.parametrize("sim", SIMS)
def test_reference_vectors_under_the_fst_simulators(vectors, sim, tmp_path):
    """test/test_simulated/runtest.sh: all `// IN:` / `// OUT:` programs, the Regression.hs pairs, and the two action
    programs (actionbug, makeDanish) — the simulators replay the actions themselves."""
    for t in vectors["line_tests"]:
        r = simulate(sim, source=t["program"], data=line_input(t["in"]), tmp=tmp_path)
        assert r.returncode == 0 and same_modulo_trailing_newlines(r.stdout, line_expected(t["out"])), (t["name"], r.stderr[-200:])
    for t in vectors["exact_tests"]:
        for inp, out in t["cases"]:
            r = simulate(sim, source=t["program"], data=inp.encode("utf-8"), tmp=tmp_path)
            assert r.returncode == 0 and r.stdout == out.encode("utf-8"), (t["name"], inp)
    with open(os.path.join(GOLDEN, "action_vectors.json"), encoding="utf-8") as f:
        for t in json.load(f)["line_tests"]:
            r = simulate(sim, source=t["program"], data=line_input(t["in"]), tmp=tmp_path)
            assert r.returncode == 0 and same_modulo_trailing_newlines(r.stdout, line_expected(t["out"])), (t["name"], r.stderr[-200:])


@pytest.mark.parametrize("sim", SIMS)
def test_simulators_agree_with_the_compiled_program(sim, tmp_path):
    """Workloads (tens of KiB) and generated programs: simulator = register-form oracle of the compiled blob, rejections
    included ("Reject", exit code 1, nothing on stdout)."""
    for prog in ("apache_log", "csv2json", "iso_datetime_to_json", "thousand_sep"):
        data = workloads.generate(workloads.PROGRAM_INPUT[prog], 30000, 9)
        r = simulate(sim, program=program_path(prog), data=data)
        assert r.returncode == 0 and r.stdout == oracle.run(blob_of(prog), data), (prog, r.stderr[-200:])
        bad = data[:5000] + b"\x01\n\x02" + data[5000:6000]
        with pytest.raises(oracle.OracleMatchError):
            oracle.run(blob_of(prog), bad)
        r = simulate(sim, program=program_path(prog), data=bad)
        assert r.returncode == 1 and r.stdout == b"" and r.stderr.endswith(b"Reject\n")
    checked = rejected = 0
    for seed in range(60):
        src = randprog.program(seed)
        try:
            blob = blob_of(src, 0)
        except Exception:
            continue
        for data in randprog.inputs(seed, 4, 40):
            try:
                want = oracle.run(blob, data)
            except oracle.OracleMatchError:
                want = None
            r = simulate(sim, source=src, data=data, tmp=tmp_path)
            if want is None:
                assert r.returncode == 1 and r.stdout == b"", (seed, src, data)
                rejected += 1
            else:
                assert r.returncode == 0 and r.stdout == want, (seed, src, data, r.stdout, want)
            checked += 1
    assert checked > 100 and rejected > 5, (checked, rejected)


def test_backtracking_worst_case_stays_linear(tmp_path):
    """`(a|a)*b`-like blow-up (ref: test/test_simulated/src/backtracking-worst-case.kex): the (state, index) barrier keeps the
    depth-first simulator from exploring a failed suffix twice."""
    src = 'main := (/a/ | /a/ /a/)* /b/\n'
    r = simulate("backtrack", source=src, data=b"a" * 20000 + b"c", tmp=tmp_path)
    assert r.returncode == 1 and r.stderr.endswith(b"Reject\n")
    r = simulate("backtrack", source=src, data=b"a" * 20000 + b"b", tmp=tmp_path)
    assert r.returncode == 0 and r.stdout == b"a" * 20000 + b"b"


def test_regex_flavour_under_the_fst_simulators(tmp_path):
    """For a regular expression the FST simulators run its own transducer (tuTransducers): the matched string is copied;
    the coder is `--sim sst` (simulateCoder, Commands.hs:317-322)."""
    kexc = os.path.join(build.OUT, "kexc")
    for sim in SIMS:
        r = subprocess.run([kexc, "simulate", "--sim", sim, "--re", "(a|b)*c"], input=b"abbac", stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0 and r.stdout == b"abbac"
        r = subprocess.run([kexc, "simulate", "--sim", sim, "--re", "(a|b)*c"], input=b"abd", stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 1 and r.stderr.endswith(b"Reject\n")
