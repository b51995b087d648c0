"""CPU: the compile-side seam that takes a `Pipeline` (include/kexc_api.h::kexc_emit_pipeline = compileProgram,
src/KMC/Program/Backends/C.hs:529-540): a front end with its own SSTs hands over tables, not source text."""
import os
import struct
import subprocess
import sys

import numpy as np
import pytest
from conftest import blob_of

from kleenexlang_amd import CompileError, emit_pipeline, workloads
from oracle import oracle

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kxp  # noqa: E402


def flip_ab_program():
    """`main := ("b" ~/a/ | "a" ~/b/ | /\\n/)*` marshalled BY HAND the way a Haskell caller would (SURVEY App. B shows the
    reference's one-state program): one block, output register only; classes a / b / newline / anything else."""
    cls = np.full(256, 3, dtype=np.uint8)
    cls[ord("a")], cls[ord("b")], cls[ord("\n")] = 0, 1, 2
    return dict(
        nstates=1, nclasses=4, init_state=0, nregs=1, class_of=cls,
        delta=[0, 0, 0, 0xFFFF], action=[0, 1, 2, 0], final_action=[3],
        # actions: 0 = outputarray("b"); 1 = outputarray("a"); 2 = outputconst(next[0]); 3 = nothing (accept)
        nactions=4, action_off=[0, 1, 2, 3, 3], ops=[(1 << 24) | 0, 0, (1 << 24) | 0, 1, (2 << 24) | 0, 0],
        nconsts=2, const_off=[0, 1, 2], const_pool=list(b"ba"),
        # path tree: a single path; every step extends leaf 0
        maxleaves=1, nback=3, back_row=[0, 1, 2, 0], nleaves=[1], final_leaf=[0],
        back=[0 | (0 << 8) | (1 << 9), 0 | (0 << 8) | (2 << 9), 0 | (1 << 8) | (0 << 9)],
        npconsts=3, pconst_off=[0, 0, 1, 2], pconst_pool=list(b"ba"), init_const=[0])


def test_hand_built_flip_ab_pipeline_runs_like_the_reference(tmp_path):
    src = tmp_path / "flip.kxp"
    lines = []
    assert emit_pipeline([flip_ab_program()], env_info="hand-made flip_ab", srcout=src, info=lines.append) == 0
    assert lines and "KXP table blob" in lines[0]
    blob = src.read_bytes()
    assert oracle.run(blob, b"abba\nbb\n") == b"baab\naa\n"                  # observed with the reference runtime
    assert oracle.run(blob, b"abba\nbb\n", path_form=True) == b"baab\naa\n"
    with pytest.raises(oracle.OracleMatchError) as e:
        oracle.run(blob, b"abxa\n")
    assert e.value.pos == 2                                                   # "Match error at input symbol 2!"
    assert oracle.run(blob, b"") == b""


def test_out_writes_a_runnable_binary_and_info_text(tmp_path):
    exe = tmp_path / "flip"
    assert emit_pipeline([flip_ab_program()], env_info="Options:\\nhand-made", out=exe) == 0
    r = subprocess.run([str(exe), "-i"], stdout=subprocess.PIPE)
    assert r.returncode == 2 and b"hand-made" in r.stdout                     # crt.c:384-386: -i prints the info, exit 2


def marshal(stage):
    """A parsed KXP stage (tests/kxp.py) as the dict emit_pipeline takes — what a Haskell front end would marshal."""
    return dict(nstates=stage.nstates, nclasses=stage.nclasses, init_state=stage.q0, nregs=stage.nregs, class_of=stage.cls,
                delta=stage.delta, action=stage.act, final_action=stage.final_act,
                nactions=len(stage.act_off) - 1, action_off=stage.act_off, ops=stage.ops,
                nconsts=len(stage.const_off) - 1, const_off=stage.const_off, const_pool=np.frombuffer(stage.cpool, dtype=np.uint8),
                maxleaves=stage.maxleaves, nback=stage.back.shape[0], back_row=stage.pback, nleaves=stage.nleaves, final_leaf=stage.fin_leaf,
                back=stage.back, npconsts=len(stage.pconst_off) - 1, pconst_off=stage.pconst_off,
                pconst_pool=np.frombuffer(stage.pool, dtype=np.uint8), init_const=stage.init_const)


@pytest.mark.parametrize("prog", ["apache_log", "csv2json", "iso_datetime_to_json", "thousand_sep", "add_commas"])
def test_tables_in_equal_source_in(prog, tmp_path):
    """Every workload program, taken apart into tables and handed to kexc_emit_pipeline, gives byte for byte the stages
    that kexc_compile builds from the source (the synchronising automaton is rebuilt from delta), and the same output."""
    blob = blob_of(prog)
    stages = kxp.parse(blob)
    out = tmp_path / "p.kxp"
    emit_pipeline([marshal(s) for s in stages], env_info="", srcout=out)
    again = out.read_bytes()
    il = struct.unpack_from("<I", blob, 16)[0]
    assert again[20:] == blob[20 + ((il + 3) & ~3):]                           # identical stage sections (info text aside)
    data = workloads.generate(workloads.PROGRAM_INPUT[prog], 20000, 3) if prog in workloads.PROGRAM_INPUT else workloads.digits(5000)
    assert oracle.run(again, data) == oracle.run(blob, data)


def test_multi_stage_and_argument_errors(tmp_path):
    src = 'start: a >> b\na := (/x/ "1" | /y/)*\nb := (~/1/ "one" | /./)*\n'
    stages = kxp.parse(blob_of(src))
    out = tmp_path / "two.kxp"
    emit_pipeline([marshal(s) for s in stages], srcout=out)
    assert oracle.run(out.read_bytes(), b"xyx") == oracle.run(blob_of(src), b"xyx") == b"xoneyxone"
    with pytest.raises(CompileError, match="8 bits"):
        emit_pipeline([flip_ab_program()], srcout=out, buffer_unit_bits=16)     # --wordsize 16 is not built
    with pytest.raises(CompileError, match="oracle/action"):
        emit_pipeline([flip_ab_program(), flip_ab_program()], srcout=out, oracle_action=True)
    with pytest.raises(CompileError, match="another version of include/kexc_api.h"):
        emit_pipeline([flip_ab_program()], srcout=out, program_size=200)          # a caller built against an older record
    bad = flip_ab_program(); bad["delta"] = [0, 7, 0, 0xFFFF]
    with pytest.raises(CompileError, match="out of range"):
        emit_pipeline([bad], srcout=out)


def coder_program_with_table():
    """The coder of `[a-d]*` the way the reference's compileCoder hands it over (Commands.hs:246-275): the class [a-d] has
    more than one member, so its symbol code is an `AppendTblI` (IL.hs:44; table built by SSTCompiler/Classes.hs:102-125),
    not a constant.  One block: on [a-d] append 00 (star: loop) and table[sym]; at end of input append 01 (star: leave).
    Path tree of the block: leaf 0 waits for a symbol (its pending choice 00), leaf 1 stands in the final state (01)."""
    cls = np.full(256, 1, dtype=np.uint8)
    cls[ord("a"):ord("d") + 1] = 0
    table = np.zeros(256, dtype=np.uint8)
    table[ord("a"):ord("d") + 1] = [0, 1, 2, 3]
    none = 0xFFFFFFFF
    return dict(
        nstates=1, nclasses=2, init_state=0, nregs=1, class_of=cls,
        delta=[0, 0xFFFF], action=[0, 0], final_action=[1],
        nactions=2, action_off=[0, 2, 3], ops=[(1 << 24) | 0, 0, (4 << 24) | 0, 0, (1 << 24) | 0, 1],
        nconsts=2, const_off=[0, 1, 2], const_pool=[0, 1],
        maxleaves=2, nback=1, back_row=[0, 0], nleaves=[2], final_leaf=[1],
        back=[0 | (0 << 8) | (0 << 9), 0 | (0 << 8) | (1 << 9)], back_table=[0, 0],
        npconsts=2, pconst_off=[0, 1, 2], pconst_pool=[0, 1], init_const=[0, 1],
        ntables=1, tbl_width=[1], tbl_data=table)


@pytest.mark.parametrize("route", ["table atom", "constants"])
def test_append_table_instruction_at_the_seam(tmp_path, monkeypatch, route):
    """AppendTblI at the seam.  A table of one-byte entries stays a TABLE ATOM (kxp_format.h: micro-op 4, the table field
    of a back entry): the program keeps its two classes.  Tables of wider entries — and any table under
    KEXC_LOWER_TABLES=1 — take the older route: classes refined until the table is constant on each (a, b, c, d, the
    rest), entries written out as constants.  Either way the function is the one `kexc compile --re '[a-d]*'` builds."""
    from kleenexlang_amd import host
    out = tmp_path / "coder.kxp"
    if route == "constants":
        monkeypatch.setenv("KEXC_LOWER_TABLES", "1")
    assert emit_pipeline([coder_program_with_table()], srcout=out) == 0
    blob = out.read_bytes()
    host.validate_blob(blob)
    st = kxp.parse(blob)[0]
    if route == "constants":
        assert st.nclasses == 5 and len(set(int(st.cls[c]) for c in b"abcd")) == 4 and st.tables is None
    else:
        assert st.nclasses == 2 and st.tables is not None and bytes(st.tables[0][ord("a"):ord("e")]) == bytes([0, 1, 2, 3])
        assert any(int(w) >> 24 == 4 for w in st.ops[0::2])          # KXP_OP_APPEND_TBL survives in the register form
    own = host.compile_regex("[a-d]*")
    assert kxp.parse(own)[0].nclasses == 2 and kxp.parse(own)[0].tables is not None
    for data in (b"", b"a", b"abcd", b"ddcbaabcd" * 50):
        want = bytes(x for ch in data for x in (0, ch - ord("a"))) + b"\x01"
        for pf in (False, True):
            assert oracle.run(blob, data, path_form=pf) == want == oracle.run(own, data, path_form=pf), (data, pf)
    with pytest.raises(oracle.OracleMatchError):
        oracle.run(blob, b"abe")
    bad = coder_program_with_table(); bad["ops"] = [(1 << 24) | 0, 0, (4 << 24) | 0, 3, (1 << 24) | 0, 1]
    with pytest.raises(CompileError, match="malformed micro-op"):
        emit_pipeline([bad], srcout=out)
    bad = coder_program_with_table(); bad["final_action"] = [0]
    with pytest.raises(CompileError, match="final action"):
        emit_pipeline([bad], srcout=out)


def mixed_copy_and_table_program():
    """One block, one path: [a-d] leaves through a table (AppendTblI: upper case), [x-z] is copied as it is (AppendSymI),
    ',' writes the constant ";" — plain copies beside table steps, the shape that needs a table id per path entry."""
    cls = np.full(256, 3, dtype=np.uint8)
    cls[ord("a"):ord("d") + 1] = 0
    cls[ord("x"):ord("z") + 1] = 1
    cls[ord(",")] = 2
    table = np.zeros(256, dtype=np.uint8)
    table[ord("a"):ord("d") + 1] = list(b"ABCD")
    return dict(
        nstates=1, nclasses=4, init_state=0, nregs=1, class_of=cls,
        delta=[0, 0, 0, 0xFFFF], action=[0, 1, 2, 0], final_action=[3],
        nactions=4, action_off=[0, 1, 2, 3, 3], ops=[(4 << 24) | 0, 0, (2 << 24) | 0, 0, (1 << 24) | 0, 0],
        nconsts=1, const_off=[0, 1], const_pool=list(b";"),
        maxleaves=1, nback=3, back_row=[0, 1, 2, 0], nleaves=[1], final_leaf=[0],
        back=[0 | (0 << 8) | (0 << 9), 0 | (1 << 8) | (0 << 9), 0 | (0 << 8) | (1 << 9)], back_table=[0, 0xFFFFFFFF, 0xFFFFFFFF],
        npconsts=2, pconst_off=[0, 0, 1], pconst_pool=list(b";"), init_const=[0],
        ntables=1, tbl_width=[1], tbl_data=table)


def mixed_expected(data):
    return bytes(c - 32 if c in b"abcd" else (ord(";") if c == ord(",") else c) for c in data)


def test_table_value_ff_in_a_stage_with_actions_is_escaped(tmp_path):
    """a stage with register actions takes no table atoms — its AppendTblI entries are written out as constants — and
    its output is a token stream whose escape byte is FF.  A table value FF must therefore leave as FF FF (the byte FF), not as
    the start of a Push / Pop / Write token: d -> FF here, and the replayed output holds the single byte."""
    prog = mixed_copy_and_table_program()
    table = np.array(prog["tbl_data"], dtype=np.uint8)
    table[ord("d")] = 0xFF
    prog["tbl_data"] = table
    prog["has_actions"], prog["action_regs"] = 1, 1
    out = tmp_path / "ff.kxp"
    assert emit_pipeline([prog], srcout=out) == 0
    blob = out.read_bytes()
    st = kxp.parse(blob)[0]
    assert st.tables is None and st.actions & 1                      # written out, and still an action stage
    assert b"\xff\xff" in bytes(st.pool) and b"\xff\xff" in bytes(st.cpool)
    for data in (b"d", b"abcd,xyz", b"dd,dxd", b"ddddddddcba" * 20):
        want = bytes(0xFF if c == ord("d") else c - 32 if c in b"abc" else (ord(";") if c == ord(",") else c) for c in data)
        for pf in (False, True):
            assert oracle.run(blob, data, path_form=pf) == want, (data, pf)


def test_plain_copies_beside_table_steps(tmp_path):
    out = tmp_path / "mixed.kxp"
    assert emit_pipeline([mixed_copy_and_table_program()], srcout=out) == 0
    blob = out.read_bytes()
    from kleenexlang_amd import host
    host.validate_blob(blob)
    for data in (b"", b"abxd,zzca", b"xyz" * 100, b",,,"):
        for pf in (False, True):
            assert oracle.run(blob, data, path_form=pf) == mixed_expected(data), (data, pf)
    with pytest.raises(oracle.OracleMatchError):
        oracle.run(blob, b"abq")


@pytest.mark.gpu
def test_table_atoms_on_the_engine(tmp_path, monkeypatch):
    """Symbol tables in the engine's output stage: per-entry table ids (plain copies beside table steps, GENERAL instances)
    and the one-table form (the piece is translated once, fast instances), also from global memory (BIG)."""
    import random
    from kleenexlang_amd import host
    out = tmp_path / "mixed.kxp"
    assert emit_pipeline([mixed_copy_and_table_program()], srcout=out) == 0
    mixed = out.read_bytes()
    assert emit_pipeline([coder_program_with_table()], srcout=out) == 0
    coder = out.read_bytes()
    rnd = random.Random(5)
    dm = bytes(rnd.choice(b"abcdxyz,") for _ in range(3 << 20))
    dc = bytes(rnd.choice(b"abcd") for _ in range(3 << 20))
    for env in ({}, {"KX_FORCE_BIG": "1"}, {"KX_FORCE_TBLMODE": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        for blob, data, want in ((mixed, dm, mixed_expected(dm)), (coder, dc, None)):
            for seg in (0, 4096):
                p = host.Program(blob, segment_bytes=seg)
                try:
                    for d in (data, data[:1000], b""):
                        assert p.run_host(d) == (mixed_expected(d) if want is not None else oracle.run(blob, d)), (env, seg, len(d))
                    with pytest.raises(host.MatchError) as e:
                        p.run_host(data[:70000] + b"Q" + data[:10])
                    assert e.value.pos == 70000
                finally:
                    p.close()
        for k in env:
            monkeypatch.delenv(k)


# ------------------------------------------------------------------ register actions at the seam
def _action_vectors():
    import json
    from conftest import GOLDEN
    with open(os.path.join(GOLDEN, "action_vectors.json"), encoding="utf-8") as f:
        return json.load(f)["line_tests"]


def _marshal_with_actions(stage):
    """A stage whose output is a token stream (kxp_format.h: FF 00 Push, FF 01 r Pop r, FF 02 r Write r, FF FF = byte FF) the way a
    front end hands it over: the ordinary tables + has_actions / action_regs (include/kexc_api.h)."""
    d = marshal(stage)
    d["has_actions"], d["action_regs"] = int(stage.actions & 1), int(stage.actions >> 8)
    return d


def test_action_program_marshalled_through_the_seam(tmp_path):
    """What a front end compiling with the reference's default flags (`--act=true`) must do for a program with register actions
    (`constructTransducerInBand`): hand over ONE transducer per stage whose output carries Push / Pop r /
    Write r in band, with has_actions / action_regs set.  Here the reference's two action vectors (actionbug, makeDanish) go that
    way — tables taken apart, marshalled by hand through ctypes, put together again by kexc_emit_pipeline — and give the
    reference's `// OUT:` lines (tests/golden/action_vectors.json), in the register form and in the path form."""
    from conftest import line_expected, line_input, same_modulo_trailing_newlines
    for t in _action_vectors():
        stages = kxp.parse(blob_of(t["program"], 0))
        assert any(s.actions & 1 for s in stages), t["name"]
        out = tmp_path / (t["name"] + ".kxp")
        assert emit_pipeline([_marshal_with_actions(s) for s in stages], srcout=out) == 0
        again = out.read_bytes()
        assert [s.actions for s in kxp.parse(again)] == [s.actions for s in stages]
        for pf in (False, True):
            got = oracle.run(again, line_input(t["in"]), path_form=pf)
            assert same_modulo_trailing_newlines(got, line_expected(t["out"])), (t["name"], pf, got)
    # without the flag the same tables are an ordinary rewriter: the tokens come out as bytes
    s0 = kxp.parse(blob_of(_action_vectors()[0]["program"], 0))
    plain = marshal(s0[0])
    assert emit_pipeline([plain], srcout=tmp_path / "plain.kxp") == 0
    assert b"\xff" in oracle.run((tmp_path / "plain.kxp").read_bytes(), b"c\n")


@pytest.mark.gpu
def test_action_program_from_the_seam_runs_on_the_engine(tmp_path):
    """…and the blob that kexc_emit_pipeline builds from the marshalled action program runs on the GPU (transducer kernels + the
    action post-pass) and writes the reference's expected lines; a long input of the same shape against the oracle."""
    from conftest import line_expected, line_input, same_modulo_trailing_newlines
    from kleenexlang_amd import host
    for t in _action_vectors():
        stages = kxp.parse(blob_of(t["program"], 0))
        out = tmp_path / (t["name"] + ".kxp")
        assert emit_pipeline([_marshal_with_actions(s) for s in stages], srcout=out) == 0
        blob = out.read_bytes()
        p = host.Program(blob)
        try:
            got = p.run_host(line_input(t["in"]))
            assert same_modulo_trailing_newlines(got, line_expected(t["out"])), (t["name"], got)
            big = line_input(t["in"]) * 3000
            assert p.run_host(big) == oracle.run(blob, big), t["name"]
        finally:
            p.close()
