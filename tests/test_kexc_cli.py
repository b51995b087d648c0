"""CPU: the `kexc compile … --out BIN` command line (src/kexc.hs:12-50 of the reference)."""
import os
import subprocess

import pytest

from kleenexlang_amd import build, program_path
from oracle import oracle

KEXC = os.path.join(build.OUT, "kexc")


def run(*args, **kw):
    return subprocess.run([KEXC, *args], stdout=subprocess.PIPE, stderr=subprocess.PIPE, **kw)


def test_usage_on_wrong_arity():
    assert run("compile").returncode == 1
    assert run("compile", "a.kex", "b.kex").returncode == 1
    assert run().returncode == 1


def test_flags_before_and_after_and_bool_forms(tmp_path):
    blob = tmp_path / "p.kxp"
    r = run("--quiet", "compile", "--opt", "0", "--la=false", program_path("flip_ab"), "--blob", str(blob), "--act=false")
    assert r.returncode == 0 and r.stdout == b"", r.stderr
    assert oracle.run(blob.read_bytes(), b"ab\n") == b"ba\n"
    r = run("compile", program_path("flip_ab"), "--blob", str(blob), "--quiet=false")
    assert r.returncode == 0 and b"SST states: 2" in r.stdout


def test_parse_error_goes_to_stderr_exit_1(tmp_path):
    bad = tmp_path / "bad.kex"
    bad.write_text('main := "unterminated\n')
    r = run("compile", "--quiet", str(bad), "--blob", str(tmp_path / "x"))
    assert r.returncode == 1 and b"line" in r.stderr


def test_out_produces_self_contained_driver(tmp_path):
    """BIN = kxrun + blob; -i prints compile info and exits 2, -h usage exits 1 (crt/crt.c:380-399)."""
    build.build_engine()
    out = tmp_path / "flip"
    r = run("compile", "--quiet", program_path("flip_ab"), "--out", str(out))
    assert r.returncode == 0, r.stderr
    i = subprocess.run([str(out), "-i"], stdout=subprocess.PIPE)
    assert i.returncode == 2 and b"SST states:  2" in i.stdout
    h = subprocess.run([str(out), "-h"], stdout=subprocess.PIPE)
    assert h.returncode == 1 and b"Normal usage" in h.stdout


def test_backend_c_with_restated_runtime(tmp_path):
    """--backend=c prints reference-shaped C; with oracle/crt_port it builds and runs anywhere."""
    crt = os.path.join(build.ROOT, "oracle", "crt_port")
    out = tmp_path / "addc"
    src = tmp_path / "addc.c"
    r = run("compile", "--quiet", "--backend=c", "--crt-dir", crt, "--srcout", str(src), program_path("add_commas"), "--out", str(out))
    assert r.returncode == 0, r.stderr
    txt = src.read_text()
    assert "#define NUM_PHASES 1" in txt and "goto l1_" in txt and "consume(1);" in txt
    p = subprocess.run([str(out)], input=b"2016\n", stdout=subprocess.PIPE)
    assert p.returncode == 0 and p.stdout == b"2,016\n"
    p = subprocess.run([str(out), "-t"], input=b"1234567\n", stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.stdout == b"1,234,567\n" and p.stderr.startswith(b"time (ms): ")
    blob_prog = os.path.join(build.ROOT, "oracle", "_build", "x")
    p = subprocess.run([str(out)], input=b"12a", stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0 and p.stdout == b"12a"


def test_slash_inside_a_character_class_does_not_end_the_regex():
    """The reference's own bench programs write an unescaped '/' inside classes
    (ref: bench/kleenex/src/jix_responsetime.kex:23 `/\\[[A-Za-z0-9: /+-]+]/`, syntax_latex.kex:81)."""
    from oracle import oracle
    from conftest import blob_of
    blob = blob_of('main := (/[a-c: /+-]+/ "." | ~/;/)*\n')
    assert oracle.run(blob, b"a/b;c+-;: /") == b"a/b.c+-.: /."


def test_trailing_dollar_is_the_end_anchor_not_a_byte():
    """anchoredRegexP (Parser.hs:204-206) strips '^' and '$'; `main := /abc$/` must accept "abc", and an
    escaped dollar stays a literal byte."""
    from kleenexlang_amd import compile_source
    from oracle import oracle
    assert oracle.run(compile_source("main := /^abc$/\n"), b"abc") == b"abc"
    with pytest.raises(oracle.OracleMatchError):
        oracle.run(compile_source("main := /abc$/\n"), b"abc$")
    assert oracle.run(compile_source("main := /abc\\$/\n"), b"abc$") == b"abc$"
    assert oracle.run(compile_source("main := /a$b/\n"), b"a$b") == b"a$b"     # not at the end: a byte


def test_backend_c_paths_with_quotes_and_spaces_never_reach_a_shell(tmp_path):
    d = tmp_path / "dir with 'quote' and $(touch pwned)"
    d.mkdir()
    out = d / "bin'ary"
    crt = os.path.join(build.ROOT, "oracle", "crt_port")
    r = subprocess.run([KEXC, "compile", "--quiet", "--backend=c", "--crt-dir", crt, program_path("flip_ab"), "--out", str(out)],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr
    assert not (tmp_path / "pwned").exists()
    r = subprocess.run([str(out)], input=b"abba\n", stdout=subprocess.PIPE)
    assert r.stdout == b"baab\n"


def test_simulate_subcommand_command_line(tmp_path):
    """`kexc simulate|interpret [--sim lockstep|backtrack|sst] FILE` (src/kexc.hs:52-83, Options.hs:69-78): option checking
    and parse errors are the compiler's; the run itself needs the HIP engine and says so when there is none (no CPU route)."""
    r = run("simulate", "--sim", "bogus", program_path("flip_ab"))
    assert r.returncode == 1 and b'"bogus" is not a valid simulation type' in r.stderr
    bad = tmp_path / "bad.kex"; bad.write_text("main := (\n")
    r = run("interpret", str(bad))
    assert r.returncode == 1 and r.stdout == b"" and r.stderr
    assert run("simulate").returncode == 1
    env = dict(os.environ, TMPDIR=str(tmp_path))
    r = run("simulate", "--sim=sst", program_path("flip_ab"), input=b"ab\n", env=env)
    if r.returncode != 0:      # (CPU-only container: loud failure, nothing printed, the temporary binary is removed)
        assert r.stdout == b"" and b"no HIP device" in r.stderr or b"cannot load the HIP engine" in r.stderr
    else:
        assert r.stdout == b"ba\n"
    assert not [f for f in os.listdir(tmp_path) if f.startswith("kexc-sim-")]
