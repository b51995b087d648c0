"""CPU: `--la=true`, the reference's default.  Its determinizer then tests WORDS of predicates (prefixTests / ldp,
SymbolicFST.hs:264-312; kills + consumeTreeMany, Determinization.hs:213-257) and the blocks take the longest word that
matches (compileTransitions, SSTCompiler.hs:113-127).  The engine's tables read one symbol per step, so the word
machine's path form is unrolled over its leaves and determinized again (kexc.h: leafGraph).  The equality asserted is
the reference's own — Tests/Regression.hs:45-53: lookahead on/off give the same output — on every vector it holds,
through both doors: `kexc compile --la=true` and the seam (kexc_il_program's block form)."""
import os
import struct
import subprocess

import numpy as np
import pytest
import randprog
from conftest import blob_of, line_expected, line_input, same_modulo_trailing_newlines
from test_regex_coder import CASES as REGEX_CASES
from test_register_actions import PROGRAMS as ACTION_PROGRAMS

from kleenexlang_amd import CompileError, build, emit_pipeline, host, program_path, workloads
from oracle import oracle

import kxp  # noqa: E402  (tests/ is on sys.path through conftest)

WORKLOADS = ["apache_log", "csv2json", "iso_datetime_to_json", "thousand_sep", "add_commas", "flip_ab"]
_LA = {}


def la_blob(name_or_source, opt=3, regex=False):
    key = (name_or_source, opt, regex)
    if key not in _LA:
        if regex or ":=" in name_or_source:
            _LA[key] = host.compile_flags(name_or_source, opt=opt, la=True, regex=regex)
        else:
            with open(program_path(name_or_source), "rb") as f:
                _LA[key] = host.compile_flags(f.read(), os.path.basename(name_or_source), opt=opt, la=True)
    return _LA[key]


def outcome(blob, data, path_form):
    try:
        return oracle.run(blob, data, path_form=path_form)
    except oracle.OracleMatchError as e:
        return ("fail", e.pos)


def source_of(prog):
    if ":=" in prog:
        return prog
    with open(program_path(prog), "rb") as f:
        return f.read()


# ------------------------------------------------------------------ the word machine itself
def test_word_machine_has_the_longest_deterministic_prefixes():
    """ldp by hand.  `main := /abc/ "1" | /abd/ "2"`: the two leaves read a, b together (one block of the partition each
    time) and part on c / d, where the context {c-leaf, d-leaf} still has each predicate as a block of its own: the
    prefixes are [a,b,c] and [a,b,d].  Test [a] kills both leaves (neither prefix is entailed), so it is no transition."""
    w = host.dump_words('main := /abc/ "1" | /abd/ "2"\n')[0]
    init = w["states"][w["init"]]
    words = sorted(bytes(b for p in e["word"] for b in range(256) if p[b >> 3] >> (b & 7) & 1) for e in init["edges"])
    assert words == [b"abc", b"abd"]
    assert init["nleaves"] == 2 and [len(e["path"]) for e in init["edges"]] == [1, 1]
    # a star's leaves have no deterministic prefix beyond the loop's first symbol: single-symbol tests only
    w = host.dump_words("main := (/a/ | /b/)*\n")[0]
    assert {len(e["word"]) for st in w["states"] for e in st["edges"]} == {1}


def test_word_lengths_on_the_workloads():
    """The literal runs of the benchmark programs become words (`" "` `"["` in apache_log, `T` `:` in the dates)."""
    longest = {p: max(len(e["word"]) for st in host.dump_words(source_of(p), p)[0]["states"] for e in st["edges"]) for p in WORKLOADS}
    assert longest["apache_log"] >= 2 and longest["iso_datetime_to_json"] >= 3 and longest["flip_ab"] == 1, longest


# ------------------------------------------------------------------ Regression.hs:45-53 on every vector
@pytest.mark.parametrize("path_form", [False, True])
def test_reference_vectors_lookahead_on_equals_off(vectors, path_form):
    for t in vectors["line_tests"]:
        data = line_input(t["in"])
        got = oracle.run(la_blob(t["program"]), data, path_form=path_form)
        assert got == oracle.run(blob_of(t["program"]), data, path_form=path_form), t["name"]
        assert same_modulo_trailing_newlines(got, line_expected(t["out"])), (t["name"], got)
    for t in vectors["exact_tests"]:
        for inp, out in t["cases"]:
            assert oracle.run(la_blob(t["program"]), inp.encode("utf-8"), path_form=path_form) == out.encode("utf-8"), (t["name"], inp)


@pytest.mark.parametrize("prog", WORKLOADS)
def test_workloads_lookahead_on_equals_off(prog):
    data = workloads.generate(workloads.PROGRAM_INPUT[prog], 200000, 11) if prog in workloads.PROGRAM_INPUT else \
        (b"abba\nbb\n" * 500 if prog == "flip_ab" else workloads.digits(30000))
    for opt in (0, 3):
        on, off = la_blob(prog, opt), blob_of(prog, opt)
        host.validate_blob(on)
        assert oracle.info(on)["maxleaves"] <= 2 * oracle.info(off)["maxleaves"], prog    # no path is carried twice
        for pf in (False, True):
            assert oracle.run(on, data, path_form=pf) == oracle.run(off, data, path_form=pf), (prog, opt, pf)
    # a rejected input is rejected at the same symbol: the table machine built from the words reads single symbols again
    bad = data[:1000] + b"\x01\x02" + data[1000:2000]
    assert outcome(la_blob(prog), bad, True) == outcome(blob_of(prog), bad, True) == outcome(la_blob(prog), bad, False)


def test_action_programs_and_coders_lookahead_on_equals_off():
    for name, src in sorted(ACTION_PROGRAMS.items()):
        on, off = la_blob(src), blob_of(src)
        for data in (b"", b"ab,12\n", b"ab cd efg ", b"aab", b"abcd", b"x\xffy\nzz\n", b"abc,1\nx,\n,99\n"):
            for pf in (False, True):
                assert outcome(on, data, pf) == outcome(off, data, pf), (name, data, pf)
    for regex, inputs in REGEX_CASES:
        on, off = la_blob(regex, regex=True), host.compile_regex(regex)
        for data in inputs:
            for pf in (False, True):
                assert outcome(on, data, pf) == outcome(off, data, pf), (regex, data, pf)


def test_random_programs_lookahead_on_equals_off():
    """Outputs AND failure positions, register form and path form, on generated programs (words of up to 7 symbols)."""
    checked = accepted = 0
    longest = 0
    for seed in range(0, 120, 2):
        src = randprog.program(seed)
        try:
            off = blob_of(src, 0)
            if oracle.info(off)["nstates"] > 500:
                continue
            on = la_blob(src, 0)
        except CompileError:
            continue
        longest = max([longest] + [len(e["word"]) for st in host.dump_words(src)[0]["states"] for e in st["edges"]])
        for data in randprog.inputs(seed, 10, 24):
            res = [outcome(b, data, pf) for b in (off, on) for pf in (False, True)]
            assert all(r == res[0] for r in res), (seed, src, data, res)
            checked += 1
            accepted += not isinstance(res[0], tuple)
    assert checked > 300 and accepted > 50 and longest >= 3, (checked, accepted, longest)


# ------------------------------------------------------------------ the seam: block form of kexc_il_program
def marshal_words(w):
    """The word machine of one stage (host.dump_words) as the block form a Haskell caller would marshal (include/kexc_api.h)."""
    pool, off, ids = bytearray(), [0], {}

    def const(b):
        b = bytes(b)
        if b not in ids:
            ids[b] = len(off) - 1
            pool.extend(b)
            off.append(len(pool))
        return ids[b]

    const(b"")
    maxleaves = max(st["nleaves"] for st in w["states"])
    init_const = [const(p) for p in w["init_path"]] + [0] * (maxleaves - len(w["init_path"]))
    block, target, length, preds, back = [], [], [], [], []
    for q, st in enumerate(w["states"]):
        for e in st["edges"]:
            block.append(q), target.append(e["to"]), length.append(len(e["word"]))
            for i, p in enumerate(e["word"]):
                preds.extend(p)
                row = [0xFFFFFFFF] * maxleaves
                for leaf, path in enumerate(e["path"]):
                    copy, bytes_ = path["steps"][i][:2]
                    row[leaf] = (path["parent"] if i == 0 else 0) | copy << 8 | const(bytes_) << 9
                back.extend(row)
    return dict(ntests=len(block), nstates=len(w["states"]), init_state=w["init"], maxleaves=maxleaves,
                nleaves=[max(st["nleaves"], 1) for st in w["states"]], final_leaf=[0xFF if st["final_leaf"] < 0 else st["final_leaf"] for st in w["states"]],
                npconsts=len(off) - 1, pconst_off=off, pconst_pool=np.frombuffer(bytes(pool) or b"\0", dtype=np.uint8), init_const=init_const,
                has_actions=int(w["action_regs"] >= 0), action_regs=max(w["action_regs"], 0),
                test_block=block, test_target=target, test_len=length, test_preds=preds, test_back=back)


def stage_sections(blob):
    il = struct.unpack_from("<I", blob, 16)[0]
    return blob[20 + ((il + 3) & ~3):]


@pytest.mark.parametrize("prog", WORKLOADS + ["swap_fields", "two_stage"])
def test_block_form_through_the_seam_gives_the_compilers_own_tables(prog, tmp_path):
    """dump_words → kexc_il_program (ntests != 0) → kexc_emit_pipeline: byte for byte the stages `kexc compile --la=true`
    builds, and the output of the `--la=false` program."""
    src = ACTION_PROGRAMS.get(prog) or source_of(prog)
    out = tmp_path / "p.kxp"
    assert emit_pipeline([marshal_words(w) for w in host.dump_words(src, prog)], env_info="", srcout=out) == 0
    blob = out.read_bytes()
    host.validate_blob(blob)
    assert stage_sections(blob) == stage_sections(la_blob(src if prog in ACTION_PROGRAMS else prog))
    if prog in workloads.PROGRAM_INPUT:
        data = workloads.generate(workloads.PROGRAM_INPUT[prog], 50000, 4)
    else:
        data = {"flip_ab": b"abba\nbb\n", "swap_fields": b"abc,1\nx,\n,99\n", "two_stage": b"abcdxa"}.get(prog) or workloads.digits(5000)
    off = blob_of(src if prog in ACTION_PROGRAMS else prog)
    for pf in (False, True):
        assert oracle.run(blob, data, path_form=pf) == oracle.run(off, data, path_form=pf), (prog, pf)


def hand_made_block():
    """One block, one path, marshalled BY HAND — what compileState (SSTCompiler.hs:129-156) would print for it:
        NextI 1 2 [AcceptI]
        IfI (avail >= 2 && next[0] == 'a' && next[1] == 'b') [AppendI "X";          ConsumeI 2; GotoI 0]
        IfI (avail >= 1 && next[0] == 'a')                    [AppendI "Y";          ConsumeI 1; GotoI 0]
        IfI (avail >= 1 && next[0] == 'c')                    [AppendSymI next[0];  ConsumeI 1; GotoI 0]
        FailI
    The word [a,b] is tried before its prefix [a]."""
    def pred(*bs):
        p = [0] * 32
        for b in bs:
            p[b >> 3] |= 1 << (b & 7)
        return p
    consts = [b"", b"X", b"Y"]
    off = np.cumsum([0] + [len(c) for c in consts])
    return dict(ntests=3, nstates=1, init_state=0, maxleaves=1, nleaves=[1], final_leaf=[0],
                npconsts=3, pconst_off=off, pconst_pool=list(b"".join(consts)), init_const=[0],
                test_block=[0, 0, 0], test_target=[0, 0, 0], test_len=[1, 2, 1],          # (in any order: the longest match wins)
                test_preds=pred(ord("a")) + pred(ord("a")) + pred(ord("b")) + pred(ord("c")),
                test_back=[0 | 0 << 8 | 2 << 9,                       # [a]:    "Y"
                           0 | 0 << 8 | 0 << 9, 0 | 0 << 8 | 1 << 9,  # [a,b]:  nothing after a, "X" after b
                           0 | 1 << 8 | 0 << 9])                      # [c]:    the symbol itself


def test_hand_made_block_takes_the_longest_word_and_falls_back(tmp_path):
    out = tmp_path / "h.kxp"
    assert emit_pipeline([hand_made_block()], srcout=out) == 0
    blob = out.read_bytes()
    for data, want in ((b"", b""), (b"ab", b"X"), (b"a", b"Y"), (b"aa", b"YY"), (b"abacab", b"XYcX"), (b"aab", b"YX"), (b"cabac", b"cXYc")):
        for pf in (False, True):
            assert oracle.run(blob, data, path_form=pf) == want, (data, pf)
    with pytest.raises(oracle.OracleMatchError) as e:
        oracle.run(blob, b"abb")
    assert e.value.pos == 2
    # the same function written as Kleenex
    twin = blob_of('main := (~/ab/ "X" | ~/a/ "Y" | /c/)*\n')
    for data in (b"abacab", b"aab", b"cabac", b"abb", b"ba"):
        assert outcome(blob, data, True) == outcome(twin, data, True), data


def test_block_form_argument_errors(tmp_path):
    out = tmp_path / "x.kxp"
    for key, val, msg in (("test_target", [0, 3, 0], "test out of range"), ("test_len", [1, 0, 1], "test out of range"),
                          ("final_leaf", [4], "leaf counts"), ("init_const", [9], "initial constant"),
                          ("test_back", [0xFFFFFFFF, 0, 1 << 9, 1 << 8], "backward entry"), ("test_back", [5, 0, 1 << 9, 1 << 8], "backward entry")):
        bad = hand_made_block()
        bad[key] = val
        with pytest.raises(CompileError, match=msg):
            emit_pipeline([bad], srcout=out)
    bad = hand_made_block()
    bad["test_preds"] = [0] * 32 + bad["test_preds"][32:]
    with pytest.raises(CompileError, match="empty predicate"):
        emit_pipeline([bad], srcout=out)


# ------------------------------------------------------------------ the command line
def test_cli_la_flag(tmp_path):
    kexc = os.path.join(build.OUT, "kexc")
    on, off = tmp_path / "on.kxp", tmp_path / "off.kxp"
    r = subprocess.run([kexc, "compile", "--la=true", program_path("iso_datetime_to_json"), "--blob", str(on)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0 and b"--la=true" in r.stdout, r.stderr
    r = subprocess.run([kexc, "compile", program_path("iso_datetime_to_json"), "--blob", str(off)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0 and b"--la=false" in r.stdout, r.stderr
    assert on.read_bytes() != off.read_bytes() and b"--la=true" in on.read_bytes()[:200]
    data = workloads.generate(workloads.PROGRAM_INPUT["iso_datetime_to_json"], 20000, 2)
    assert oracle.run(on.read_bytes(), data) == oracle.run(off.read_bytes(), data)
    assert stage_sections(on.read_bytes()) == stage_sections(la_blob("iso_datetime_to_json"))
    r = subprocess.run([kexc, "compile", "--la", "--re", "(ab|a)(bc|c)?", "--blob", str(on)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr
    assert oracle.run(on.read_bytes(), b"abc") == oracle.run(host.compile_regex("(ab|a)(bc|c)?"), b"abc")


# ------------------------------------------------------------------ on the engine
@pytest.mark.gpu
@pytest.mark.parametrize("prog", ["apache_log", "csv2json", "iso_datetime_to_json", "add_commas"])
def test_lookahead_tables_on_the_engine(prog):
    from kleenexlang_amd import MatchError, Program
    blob = la_blob(prog)
    data = workloads.generate(workloads.PROGRAM_INPUT[prog], 8 << 20, 21) if prog in workloads.PROGRAM_INPUT else workloads.digits(1 << 20)
    p = Program(blob)
    try:
        assert p.run_host(data) == oracle.run(blob_of(prog), data, path_form=True)
        bad = data[:300000] + b"\x01\x02" + data[300000:400000]
        want = outcome(blob_of(prog), bad, True)
        if isinstance(want, tuple):
            with pytest.raises(MatchError) as e:
                p.run_host(bad)
            assert ("fail", e.value.pos) == want
        else:   # (add_commas copies what is not a digit: nothing to reject)
            assert p.run_host(bad) == want
    finally:
        p.close()
