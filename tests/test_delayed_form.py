"""The delayed form of a stage (round 5; kleenexlang_amd/csrc/engine/kx_delayed.h): a forward transducer with fixed delay K that the
engine runs instead of the path form's forward + backward passes wherever K further symbols decide every step's output.

CPU part (no GPU): the table the library builds from a blob (kx_df_describe / kx_df_pending — host code of libkxhip.so) is
  * compared with an independent restatement of the construction in this file (number of product states / transitions / undecided
    contexts reachable from the start state), and
  * SIMULATED here, in Python, on inputs — product transitions, the bytes each step writes K symbols late, the tail from the
    pending functions at the end leaf — and the result compared with the oracle (which runs the register form the way
    crt.c does).  A context the delay does not decide must show up as the escape state, never as wrong output.
GPU part: the engine on the delayed form, on the general engine, and falling back from one to the other in mid-run."""
import os
import random
import struct

import pytest
import kxp
import randprog
from conftest import blob_of

from kleenexlang_amd import CompileError, MatchError, Program, host, workloads
from oracle import oracle


class Escape(Exception):
    pass


def describe(blob, K, J=None):
    old = {k: os.environ.get(k) for k in ("KX_DF_K", "KX_DF_J")}
    os.environ["KX_DF_K"] = str(K)
    if J is not None:
        os.environ["KX_DF_J"] = str(J)
    try:
        return host.df_describe(blob)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def pending(blob, state, slot, K):
    old = os.environ.get("KX_DF_K")
    os.environ["KX_DF_K"] = str(K)
    try:
        return host.df_pending(blob, 0, state, slot)
    finally:
        if old is None:
            os.environ.pop("KX_DF_K")
        else:
            os.environ["KX_DF_K"] = old


def deferred(blob, state, K):
    old = os.environ.get("KX_DF_K")
    os.environ["KX_DF_K"] = str(K)
    try:
        return host.df_deferred(blob, 0, state)
    finally:
        if old is None:
            os.environ.pop("KX_DF_K")
        else:
            os.environ["KX_DF_K"] = old


def simulate(blob, info, img, data, K):
    """Run the table image on `data`: output bytes, or ('fail', pos); raises Escape where the delay does not decide."""
    st = kxp.parse(blob)[0]
    C, dead, esc = info.nclasses, info.dead_handle, info.escape_handle
    h = info.start_handle
    out = bytearray()
    csh = 3 if C > 31 else 0     # (more than 31 byte classes: the class table holds class indices)
    for s, b in enumerate(data):
        a = h + (img[b] << csh)
        lo, hi = struct.unpack_from("<II", img, a)
        h = lo & 0xFFFF
        if h == dead:
            return ("fail", s)
        if h == esc:
            raise Escape(s)
        copy = 0 if hi & 1 else 1
        if copy:
            assert s - K >= 0
            out.append(data[s - K])
        ln = (hi >> 24) - copy
        assert (ln > 0) == bool((hi >> 23) & 1)
        if ln:
            off = info.off_pool + ((hi >> 10) & 0x1FFF) * 16
            out += img[off:off + ln]
    idx = (h - 256) // (C * 8)
    assert idx < info.nstates
    n = len(data)
    q = None
    tail = bytearray(deferred(blob, idx, K))    # constants still due (merged constants): older than every pending step
    for j in range(K):
        q, kinds = pending(blob, idx, j, K)
        fl = int(st.fin_leaf[q])
        if fl == 0xFF:
            return ("fail", n)
        kd = kinds[0] if len(kinds) == 1 else kinds[fl]
        if kd & 1:
            tail.append(data[n - K + j])
        pc = kd >> 1
        if pc < len(st.pconst_off) - 1:
            tail += st.pool[int(st.pconst_off[pc]):int(st.pconst_off[pc + 1])]
    return bytes(out + tail)


def expect(blob, data):
    try:
        return oracle.run(blob, data)
    except oracle.OracleMatchError as e:
        return ("fail", e.pos)


def model_counts(blob, K, J=0):
    """Independent restatement of the construction, start-reachable part only: (states, transitions, undecided contexts).
    J > 0: merged constants — a state also holds the constants that are due, (text, age) each; a step whose successor cannot copy
    keeps them (but writes those that have waited J steps), any other step writes them all in front of its own."""
    st = kxp.parse(blob)[0]
    back = st.back
    canon = {}
    def kind(e):
        pc = (int(e) >> 9) & 0x7FFF
        text = bytes(st.pool[int(st.pconst_off[pc]):int(st.pconst_off[pc + 1])])
        return ((int(e) >> 8) & 1, canon.setdefault(text, pc))
    def norm(g):
        return ("v", g[0]) if len(set(g)) == 1 else ("f", tuple(g))
    nothing = ("v", (0, canon.setdefault(b"", -1)))
    nl0 = int(st.nleaves[st.q0])
    def init_kind(l):
        pc = int(st.init_const[l])
        return (0, canon.setdefault(bytes(st.pool[int(st.pconst_off[pc]):int(st.pconst_off[pc + 1])]), pc))
    text_of = {}
    def const_text(kd):
        if not text_of:
            text_of.update({v: k for k, v in canon.items()})
        return text_of.get(kd[1], b"")
    start = (st.q0, tuple([nothing] * (K - 1) + [norm([init_kind(l) for l in range(nl0)])]), ())
    ids = {start: 0}
    todo = [start]
    ntr = nesc = 0
    while todo:
        q, pend, due0 = todo.pop()
        for c in range(st.nclasses):
            t = int(st.delta[q, c])
            if t == 0xFFFF:
                continue
            r = int(st.pback[q, c])
            nl = int(st.nleaves[t])
            live = [l for l in range(nl) if int(back[r, l]) != 0xFFFFFFFF]
            par = {l: int(back[r, l]) & 0xFF for l in live}
            def through(it):
                if it[0] == "v":
                    return it
                vals = {l: it[1][par[l]] for l in live}
                return norm([vals.get(l, vals[live[0]]) for l in range(nl)])
            newp = [through(it) for it in pend]
            own = {l: kind(back[r, l]) for l in live}
            newp.append(norm([own.get(l, own[live[0]]) for l in range(nl)]))
            if newp[0][0] != "v":
                nesc += 1
                continue
            text_of.clear()
            due = list(due0)
            own_text = const_text(newp[0][1])
            if own_text:
                due.append((own_text, 0))
            nxt = newp[1]
            may_copy = J == 0 or (nxt[1][0] == 1 if nxt[0] == "v" else any(k[0] == 1 for k in nxt[1]))
            nout = len(due) if may_copy else sum(1 for _, age in due if age >= J)
            assert not (due0 and newp[0][1][0]), "a state with constants due copies"
            ns = (t, tuple(newp[1:]), tuple((tx, age + 1) for tx, age in due[nout:]))
            ntr += 1
            if ns not in ids:
                ids[ns] = len(ids)
                todo.append(ns)
    return len(ids), ntr, nesc


WORKLOADS = {"apache_log": "apache_log", "csv2json": "csv", "iso_datetime_to_json": "datetime", "thousand_sep": "numbers", "flip_ab": None,
             "add_commas": None}


@pytest.mark.parametrize("K", [1, 2])
def test_construction_matches_an_independent_restatement(K):
    for prog in WORKLOADS:
        blob = blob_of(prog)
        for J in (0, 2, 6):
            info, _ = describe(blob, K, J)
            states, ntr, nesc = model_counts(blob, K, J)
            assert info.delay == K and info.merge_window == J, (prog, K, J, info.merge_window)
            assert (info.transitions_start, info.escapes_start) == (ntr, nesc), (prog, K, J)
            assert info.nstates >= states and info.transitions >= ntr
    # merged constants on the BASELINE log: 15 constants per line become 8 (window 6) or 9 (window 2)
    info0, _ = describe(blob_of("apache_log"), 2, 0)
    info2, _ = describe(blob_of("apache_log"), 2, 2)
    assert info2.nstates > info0.nstates


def test_a_run_started_in_mid_input_joins_the_run_from_the_start():
    """What lets a segment, a window or a shard begin in (state, nothing pending, nothing due): K + J symbols later the product state
    is the one the run from the start of the input is in — the pending functions depend on the last K transitions, the constants
    due on the J steps behind them (each constant's step is a function of what FOLLOWS it, never of what was due before)."""
    r = random.Random(17)
    for prog, shape in (("apache_log", "apache_log"), ("csv2json", "csv"), ("iso_datetime_to_json", "datetime")):
        blob = blob_of(prog)
        st = kxp.parse(blob)[0]
        for K, J in ((1, 0), (2, 0), (2, 2), (2, 6), (1, 3), (2, None)):
            info, img = describe(blob, K, J)
            if not info.nstates:
                continue
            J = info.merge_window
            C = info.nclasses
            data = workloads.generate(shape, 5000, seed=11)
            def run(h, lo, hi):
                hs = []
                for b in data[lo:hi]:
                    h = struct.unpack_from("<I", img, h + img[b])[0] & 0xFFFF
                    hs.append(h)
                return hs
            true = run(info.start_handle, 0, len(data))
            if info.escape_handle in true:
                continue          # (apache_log at K = 1)
            # the SST state sequence (the oracle's register form has no such view: walk the blob's own transition table)
            q, qs = st.q0, []
            for b in data:
                q = int(st.delta[q, st.cls[b]]); qs.append(q)
            old = {k: os.environ.get(k) for k in ("KX_DF_K", "KX_DF_J")}
            os.environ["KX_DF_K"] = str(K); os.environ["KX_DF_J"] = str(J)
            try:
                for _ in range(60):
                    p = r.randrange(1, len(data) - 200)
                    h0 = host.df_start_of_state(blob, 0, qs[p - 1])
                    assert h0 != 0xFFFF
                    mine = run(h0, p, p + K + J + 40)
                    assert mine[K + J - 1:] == true[p + K + J - 1:p + K + J + 40], (prog, K, J, p)
            finally:
                for k, v in old.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v


def test_which_workloads_have_a_delayed_form():
    """apache_log, csv2json and iso_datetime_to_json are decided by two symbols everywhere the start state reaches (apache_log up to the
    escaped-quote contexts of its quoted fields).  thousand_sep and add_commas group digits from the END of a number: no delay decides
    — the table exists (at most a quarter of the start-reachable transitions escape) but every input leaves it in its first pieces
    (the GPU tests see the fall-back)."""
    for prog, want in (("apache_log", 1), ("csv2json", 1), ("iso_datetime_to_json", 1), ("flip_ab", 1), ("thousand_sep", 1), ("add_commas", 1)):
        info, _ = host.df_describe(blob_of(prog), with_image=False)
        assert info.available == want, (prog, info.reason, info.escapes_start, info.transitions_start)
    info, _ = host.df_describe(blob_of("csv2json"), with_image=False)
    assert info.escapes == 0 and info.escapes_start == 0           # fully static: no input can escape
    info, _ = host.df_describe(blob_of("apache_log"), with_image=False)
    assert 0 < info.escapes_start * 4 <= info.transitions_start     # a few contexts (backslash before a quote) stay undecided


@pytest.mark.parametrize("K", [1, 2])
def test_simulated_table_against_the_oracle_on_the_workloads(K):
    r = random.Random(5)
    for prog, shape in WORKLOADS.items():
        blob = blob_of(prog)
        info, img = describe(blob, K)
        if not info.nstates:
            continue
        inputs = []
        if shape:
            whole = workloads.generate(shape, 6000, seed=9)
            inputs += [whole, whole[:1], whole[:2], whole[:3], whole[:777], b""]
            for _ in range(6):   # damaged inputs: rejected at the oracle's position
                i = r.randrange(len(whole))
                inputs.append(whole[:i] + bytes([r.choice(b"\x00\"\\ ,\n9a")]) + whole[i + 1:])
        else:
            inputs += [b"", b"a", b"ab", b"abba", b"12345678", b"1", b"1234\n"]
        nesc = 0
        for data in inputs:
            try:
                got = simulate(blob, info, img, data, K)
            except Escape:
                nesc += 1
                continue
            assert got == expect(blob, data), (prog, K, data[:60])
        if prog in ("csv2json", "iso_datetime_to_json", "flip_ab"):
            assert nesc == 0, prog
        if prog == "apache_log" and K == 2:
            assert nesc <= 2, nesc    # (only a damaged byte that happens to be a backslash may escape)


def test_simulated_table_against_the_oracle_on_reference_vectors(vectors):
    checked = 0
    for t in vectors["exact_tests"]:
        blob = blob_of(t["program"], 3)
        if len(kxp.parse(blob)) != 1 or kxp.parse(blob)[0].actions & 1:
            continue
        for K in (1, 2):
            info, img = describe(blob, K)
            if not info.nstates:
                continue
            for inp, out in t["cases"]:
                try:
                    got = simulate(blob, info, img, inp.encode("utf-8"), K)
                except Escape:
                    continue
                assert got == out.encode("utf-8"), (t["name"], K, inp)
                checked += 1
    assert checked >= 10, checked


def test_simulated_table_against_the_oracle_on_random_programs():
    """Generated programs (regex and term operators, lazy forms, ranges, suppression, constants) × inputs: wherever the table does not
    escape, its output — and its failure position — are the oracle's."""
    checked = escaped = 0
    for seed in range(0, 120):
        src = randprog.program(seed)
        try:
            blob = blob_of(src, 3)
        except CompileError:
            continue
        if oracle.info(blob)["nstates"] > 400:
            continue
        for K in (1, 2):
            info, img = describe(blob, K)
            if not info.nstates:
                continue
            for data in randprog.inputs(seed, 8, 20):
                try:
                    got = simulate(blob, info, img, data, K)
                except Escape:
                    escaped += 1
                    continue
                assert got == expect(blob, data), (seed, src, K, data)
                checked += 1
    assert checked > 800 and escaped > 0, (checked, escaped)


def many_classes_program(width):
    """40 letters, each with its own replacement (a rot13-like table written out as alternatives: every letter is a byte class of its
    own — more than the 31 whose class * 8 fits the class table's byte), digits copied; `width` = bytes a letter becomes."""
    letters = "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMN"
    alts = " | ".join('~/%s/ "%s"' % (c, (c.swapcase() + c) * (width // 2)) for c in letters)
    return "main := (%s | /[0-9 ]/)*\n" % alts, letters


def test_more_than_31_byte_classes():
    r = random.Random(11)
    for width in (2, 8):
        src, letters = many_classes_program(width)
        blob = blob_of(src)
        assert kxp.parse(blob)[0].nclasses > 31
        # (the engine's own choice of delay: one symbol decides everything here, and a second would square the 42 kinds of output
        #  that wait in the state — beyond the image budget)
        info, img = host.df_describe(blob)
        assert info.available == 1 and info.escapes == 0 and info.delay == 1, (info.reason, info.delay)
        assert describe(blob, 2)[0].available == 0
        for n in (0, 1, 5, 300):
            data = bytes(r.choice((letters + "0123456789 ").encode()) for _ in range(n))
            assert simulate(blob, info, img, data, 1) == expect(blob, data), (width, data)
        bad = b"abc" + b"?" + b"def"
        assert simulate(blob, info, img, bad, 1) == expect(blob, bad) == ("fail", 3)


# ------------------------------------------------------------------------------------------------------------------ GPU
def _run(blob, data, **env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        p = Program(blob)
        try:
            return p.run_host(data), p.stage_delayed_form(0)
        except MatchError as e:
            return ("fail", e.pos), p.stage_delayed_form(0)
        finally:
            p.close()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.gpu
@pytest.mark.parametrize("prog,shape", [("apache_log", "apache_log"), ("csv2json", "csv"), ("iso_datetime_to_json", "datetime")])
def test_engine_runs_the_delayed_form_and_agrees_with_the_general_engine(prog, shape):
    blob = blob_of(prog)
    for n, seed in ((100, 1), (5000, 2), (3 << 20, 3)):
        data = workloads.generate(shape, n, seed)
        want = expect(blob, data)
        for K in (1, 2):
            got, state = _run(blob, data, KX_DF_K=K)
            # (apache_log's synthetic lines need two symbols after a field's closing quote: at K = 1 every line holds undecided contexts —
            #  in every segment: the slow path is not even armed, the shard falls back and the stage gives the form up, 3; the two
            #  segments of the small input only make it back off, 2)
            assert got == want and state == ((3 if n > 5000 else 2) if prog == "apache_log" and K == 1 and n > 100 else 1), (prog, n, K, state)
        got, state = _run(blob, data, KX_DF=0)
        assert got == want and state == 0


@pytest.mark.gpu
def test_more_than_31_byte_classes_on_the_engine():
    """The class table of such a stage holds class indices and the sequences shift (`piece_dfrun1w`, `piece_dfwalk2cw_*`, `piece_dfwalk1cw_*`);
    width 8 makes the output four times the input: a lane takes half a piece."""
    r = random.Random(12)
    for width in (2, 8):
        src, letters = many_classes_program(width)
        blob = blob_of(src)
        for n in (100, 70000, 3 << 20):
            data = bytes(r.choice((letters + "0123456789 ").encode()) for _ in range(min(n, 70000))) * max(1, n // 70000)
            want = expect(blob, data)
            got, state = _run(blob, data)
            assert got == want and state == 1, (width, n)
            got, state = _run(blob, data, KX_DF=0)
            assert got == want and state == 0
        bad = data[:100000] + b"?" + data[100001:]
        got, _ = _run(blob, bad)
        assert got == expect(blob, bad)


@pytest.mark.gpu
def test_program_that_always_escapes_gives_the_delayed_form_up():
    """thousand_sep: where the commas go is decided by the END of the number.  A run that tries the delayed form leaves it in the first
    pieces of (nearly) every segment and is redone by the general engine — and the stage gives the form up for good
    (kx_stage_delayed_form: 3), instead of paying a forward pass for nothing ever more rarely (round 5: runs 1, 3, 7, …).
    kx_stage_reset_delayed_form makes it try again.  (With the slow path of round 6 the lanes resolve their own stretches — exact, but
    slow: the same verdict follows from the share of the input that went that way.)"""
    blob = blob_of("thousand_sep")
    data = workloads.generate("numbers", 1 << 20, 6)
    want = oracle.run(blob, data)
    p = Program(blob)
    try:
        assert p.stage_delayed_form(0) == 1
        states = []
        for _ in range(4):
            assert p.run_host(data) == want
            states.append(p.stage_delayed_form(0))
        assert states == [3, 3, 3, 3], states
        p.reset_delayed_form(0)
        assert p.stage_delayed_form(0) == 1
        assert p.run_host(data) == want and p.stage_delayed_form(0) == 3
    finally:
        p.close()
    # without the slow path (round 5's engine) a SMALL input — fewer than 8 escaping lanes — only makes the stage back off: 2
    p = Program(blob, config=host.config_from_env(disable=host.KX_OFF_SLOW))
    try:
        small = data[:9000]
        small = small[:small.rfind(b"\n") + 1]
        states = []
        for _ in range(4):
            assert p.run_host(small) == oracle.run(blob, small)
            states.append(p.stage_delayed_form(0))
        # run 1 tries and escapes in its 3 segments (1 run to skip), run 2 skips, run 3 tries and escapes (3 to skip), run 4 skips
        assert states == [2, 1, 2, 2], states
    finally:
        p.close()


@pytest.mark.gpu
def test_escape_in_mid_run_falls_back_to_the_general_engine():
    """apache_log's quoted fields allow \\" inside: after a backslash the table cannot tell within two symbols whether a quote closes the
    field.  A log that holds such a line far inside must come out as the oracle has it — through the fall-back — and the stage
    remembers: the next run goes straight to the general engine.  A damaged log is rejected at the oracle's position either way."""
    blob = blob_of("apache_log")
    base = workloads.generate("apache_log", 1 << 20, 4)
    lines = base.split(b"\n")
    k = len(lines) // 2
    # (after \" the field may have ended: what follows — blanks, a digit — would also start the status field, and only the x decides)
    lines[k] = lines[k].replace(b' HTTP/', b'\\"   5x HTTP/', 1)
    data = b"\n".join(lines)
    want = expect(blob, data)
    assert not isinstance(want, tuple)
    p = Program(blob, config=host.config_from_env(disable=host.KX_OFF_SLOW))    # round 5's engine: no slow path in the forward pass
    try:
        assert p.stage_delayed_form(0) == 1
        assert p.run_host(base) == oracle.run(blob, base) and p.stage_delayed_form(0) == 1
        assert p.run_host(data) == want
        assert p.stage_delayed_form(0) == 2                                            # backing off: the next run goes to the general engine
        assert p.run_host(base) == oracle.run(blob, base) and p.stage_delayed_form(0) == 1
        assert p.run_host(base) == oracle.run(blob, base) and p.stage_delayed_form(0) == 1   # on the form again
    finally:
        p.close()
    p = Program(blob)                                                                  # round 6: the lane resolves the stretch, the shard stays on the form
    try:
        assert p.run_host(data) == want and p.stage_delayed_form(0) == 1
    finally:
        p.close()
    bad = base[:700000] + b"\x00" + base[700001:]
    for env in ({}, {"KX_DF": 0}):
        got, _ = _run(blob, bad, **env)
        assert got == expect(blob, bad)


@pytest.mark.gpu
def test_delayed_form_in_windows_and_shards(tmp_path):
    """Later shards start from (state, nothing pending), take their head from the previous shard's end state (k_dhead) and owe their
    predecessor a start-leaf map (k_dmap); the last K steps of every shard are written by the host from the pending functions."""
    import subprocess
    from kleenexlang_amd import build, program_path
    for prog, shape in (("csv2json", "csv"), ("apache_log", "apache_log"), ("iso_datetime_to_json", "datetime")):
        exe = tmp_path / prog
        assert subprocess.run([os.path.join(build.OUT, "kexc"), "compile", "--quiet", program_path(prog), "--out", str(exe)]).returncode == 0
        data = workloads.generate(shape, 1 << 20, 23)
        want = oracle.run(blob_of(prog), data)
        for win in (4096, 50000, 262144):
            r = subprocess.run([str(exe)], input=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, KX_WINDOW_BYTES=str(win), KX_DEBUG="1"))
            assert r.returncode == 0 and r.stdout == want, (prog, win)
            assert b"delayed form" in r.stderr and b"falls back" not in r.stderr, (prog, win)
        src = tmp_path / (prog + ".in")
        src.write_bytes(data)
        for g in (2, 3):
            with open(src, "rb") as fin:
                r = subprocess.run([str(exe), "--gpus", str(g)], stdin=fin, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                   env=dict(os.environ, KX_DEBUG="1", KX_SHARD_SAME_DEVICE="1"))   # (the box has one GPU: every rank on device 0)
            assert r.returncode == 0 and r.stdout == want, (prog, g, r.stderr[-300:])


@pytest.mark.gpu
def test_short_last_window_whose_alternatives_stay_open_to_the_end_of_input(tmp_path):
    """A last window (or shard) that lies inside the input's last line: the walk that composes its start-leaf map (k_dmap) reaches the
    end of the input with the alternatives "another record follows / this was the last one" still open.  The map handed to the
    window before it must then be taken at the FINAL state's leaf (as the general engine's k_backlen pins it), not at leaf 0
    (wrong bytes in the previous window's tail, or in all of it where that window ran the general engine)."""
    import subprocess
    from kleenexlang_amd import build, program_path
    seen = False
    for prog, shape in (("apache_log", "apache_log"), ("csv2json", "csv"), ("iso_datetime_to_json", "datetime")):
        exe = tmp_path / prog
        assert subprocess.run([os.path.join(build.OUT, "kexc"), "compile", "--quiet", program_path(prog), "--out", str(exe)]).returncode == 0
        base = workloads.generate(shape, 40000, 29)
        lines = base.split(b"\n")[:-1]
        for tail in (1, 2, 3, 9, 40):
            # the input ends `tail` bytes behind a window boundary of 4096: the last window holds the end of the last line only
            data = b""
            for ln in lines:
                if len(data) + len(ln) + 1 > 3 * 4096 + tail:
                    break
                data += ln + b"\n"
            pad = 3 * 4096 + tail - len(data)
            last = lines[0]
            if pad < len(last) + 1:      # (lengthen the line before the last so that a whole last line fits)
                continue
            # a last line of exactly `pad` bytes: stretch a free-text part of a generated line
            if shape == "apache_log":
                ln = last[:-1] + b"x" * (pad - 1 - len(last)) + last[-1:]
            elif shape == "csv":
                f = last.split(b",")
                f[1] = f[1] + b"x" * (pad - 1 - len(last))
                ln = b",".join(f)
            else:
                continue
            data += ln + b"\n"
            assert len(data) == 3 * 4096 + tail
            want = expect(blob_of(prog), data)
            if isinstance(want, tuple):
                continue
            for env in ({}, {"KX_DF": "0"}):
                r = subprocess.run([str(exe)], input=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                   env=dict(os.environ, KX_WINDOW_BYTES="4096", KX_DEBUG="1", **env))
                assert r.returncode == 0 and r.stdout == want, (prog, tail, env)
                seen = seen or b"stays open to the end of the input" in r.stderr
            src = tmp_path / (prog + ".in")
            src.write_bytes(data)
            with open(src, "rb") as fin:
                r = subprocess.run([str(exe), "--gpus", "4"], stdin=fin, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                   env=dict(os.environ, KX_DEBUG="1", KX_SHARD_SAME_DEVICE="1"))
            assert r.returncode == 0 and r.stdout == want, (prog, tail, "shards", r.stderr[-300:])
            seen = seen or b"stays open to the end of the input" in r.stderr
    assert seen, "no case reached the end of the input with an open map: the test does not exercise the path it is for"


def _with_escaped_quotes(base, every, r):
    """apache_log lines of which about one in `every` holds \\" inside its request field, in the two shapes that K = 2 symbols do not
    decide (after \\" the field may have ended: blanks and a digit would also start the status field; only the letter behind decides)."""
    lines = base.split(b"\n")
    hit = 0
    for k in range(len(lines) - 1):
        if r.randrange(every) == 0 and b' HTTP/' in lines[k]:
            lines[k] = lines[k].replace(b' HTTP/', r.choice([b'\\"   5x HTTP/', b'\\" 7a\\" 33b HTTP/', b'\\" HTTP/']), 1)
            hit += 1
    return b"\n".join(lines), hit


class _Lanes:
    """The forward pass's lanes in Python: parts [B_i, B_i+1) that begin K + J symbols (rounded up to a piece) behind a position whose
    SST state is known, each run on the table from (state, nothing pending, nothing due) — with the EXACT SLOW PATH of k_dforward
    (kx_dfkernels.inc: df_slow_resolve) restated here for a piece that holds an undecided context: anchors forward, rows backward,
    the three ways out (the table takes over at a piece boundary / the part is handed over at its end minus the constants the next
    lane still has due / the input ends in the final state's leaf)."""

    def __init__(self, blob, K, J):
        self.blob, self.K = blob, K
        self.st = kxp.parse(blob)[0]
        self.info, self.img = describe(blob, K, J)
        self.J = self.info.merge_window
        self.cfg = host.config_from_env({"KX_DF_K": str(K), "KX_DF_J": str(self.J)})
        self.C = self.info.nclasses
        self._sos, self._pend, self._def = {}, {}, {}
        self.is_last = True      # the shard is the input's last (the final state's leaf ends it)

    def idx(self, h):
        return (h - 256) // (self.C * 8)

    def sos(self, q):
        if q not in self._sos:
            self._sos[q] = host.df_start_of_state(self.blob, 0, q, cfg=self.cfg)
        return self._sos[q]

    def pend(self, i, j):
        if (i, j) not in self._pend:
            self._pend[(i, j)] = host.df_pending(self.blob, 0, i, j, cfg=self.cfg)
        return self._pend[(i, j)]

    def due(self, i):       # constants due in product state i, as one byte string per constant
        if i not in self._def:
            self._def[i] = host.df_deferred(self.blob, 0, i, cfg=self.cfg)
        return self._def[i]

    def step(self, h, b):
        lo, hi = struct.unpack_from("<II", self.img, h + self.img[b])
        return lo & 0xFFFF, hi

    def text(self, pc):
        st = self.st
        return bytes(st.pool[int(st.pconst_off[pc]):int(st.pconst_off[pc + 1])]) if pc < len(st.pconst_off) - 1 else b""

    def kind_bytes(self, kd, data, u):
        return (data[u:u + 1] if kd & 1 else b"") + self.text(kd >> 1)

    def slow(self, data, a, h0, end, last, nxt):
        """-> (bytes the stretch writes, position where the table takes over, handle there) or None (not resolvable here)"""
        st, K, n = self.st, self.K, len(data)
        i0 = self.idx(h0)
        q = self.pend(i0, 0)[0]
        out_def = self.due(i0)
        pend_kinds = [None] * K
        rows, qs = {}, {a: q}
        M = list(range(int(st.nleaves[q])))
        t, anchor, prev = a, a, a
        kinds = {}

        def back_to(frm, to, leaf):
            for u in range(to - 1, frm - 1, -1):
                e = int(st.back[rows[u], leaf])
                kinds[u] = ((e >> 8) & 1) | ((e >> 9) << 1)
                leaf = e & 0xFF
            return leaf

        def anchor_known(leaf):
            if anchor == a:
                for j in range(K):
                    ks = self.pend(i0, j)[1]
                    pend_kinds[j] = ks[0] if len(ks) == 1 else ks[leaf]
            else:
                back_to(prev, anchor, leaf)

        mode = 0
        while True:
            if t >= n:
                if not self.is_last:
                    return None
                if int(st.fin_leaf[q]) == 0xFF:
                    return ("fail", n)                            # (end of input in a state that is not final)
                la = back_to(anchor, n, int(st.fin_leaf[q]))
                anchor_known(la)
                mode, e_out = (3, n) if last else (2, end)      # (a lane that is not the input's last hands its part over as ever)
                break
            c = int(st.cls[data[t]])
            nq = int(st.delta[q, c])
            if nq == 0xFFFF:
                return ("fail", t)
            r = int(st.pback[q, c])
            rows[t] = r
            M2 = []
            for l in range(int(st.nleaves[nq])):
                e = int(st.back[r, l])
                M2.append(None if e == 0xFFFFFFFF else M[e & 0xFF])
            live = [x for x in M2 if x is not None]
            M, q, t = M2, nq, t + 1
            qs[t] = q
            if len(set(live)) != 1 and (t - a) % 64 == 0 and t + 64 <= end and self.sos(q) != 0xFFFF:
                # output equivalence at a piece boundary: the (at most 4) live leaves write the same on every open step
                lv = [l for l, x in enumerate(M) if x is not None]
                if 0 < len(lv) <= 4:
                    cur, same = list(lv), True
                    for u in range(t - 1, prev - 1, -1):
                        es = [int(st.back[rows[u], c_]) for c_ in cur]
                        if len({e >> 8 for e in es}) != 1:
                            same = False
                            break
                        cur = [e & 0xFF for e in es]
                    if same and pend_kinds[0] is None:
                        for j in range(K):
                            ks = self.pend(i0, j)[1]
                            if len({(ks[0] if len(ks) == 1 else ks[c_]) for c_ in cur}) != 1:
                                same = False
                    if same:
                        la = back_to(prev, t, lv[0])
                        if pend_kinds[0] is None:
                            for j in range(K):
                                ks = self.pend(i0, j)[1]
                                pend_kinds[j] = ks[0] if len(ks) == 1 else ks[la]
                        mode, e_out = 1, t
                        break
            if len(set(live)) != 1:
                continue
            anchor_known(live[0])
            decided = anchor
            prev, anchor = anchor, t
            M = list(range(int(st.nleaves[q])))
            e = decided & ~63
            if e > a and e + 64 <= end and self.sos(qs[e]) != 0xFFFF:
                mode, e_out = 1, e
                break
            if not last and decided + K >= end:
                mode, e_out = 2, end
                break
        items = [(False, out_def)]                                        # (has an input byte?, bytes)
        for j in range(K):
            items.append((True, self.kind_bytes(pend_kinds[j], data, a - K + j)))
        stop = e_out - K if mode == 2 else e_out
        for u in range(a, stop):
            items.append((True, self.kind_bytes(kinds[u], data, u)))
        if mode == 2:
            if a - K + K > stop:      # (the pending steps themselves lie behind end - K: cannot happen, a + 64 <= end)
                return None
            # the constants the next lane's state still has due at `end` are the LAST constants of this stretch: theirs to write
            due_next = self.due(self.idx(nxt))
            outs = [bytearray(b) for _, b in items]
            hasb = [hb for hb, _ in items]
            rest = len(due_next)
            k = len(outs) - 1
            while rest and k >= 0:
                copied = 1 if hasb[k] and len(outs[k]) and outs[k][:1] == data[a - K + (k - 1):a - K + k] and False else 0
                k -= 1
            # (restated on kinds instead: strip from the back)
            seq = [("def", None)] + [("p", j) for j in range(K)] + [("s", u) for u in range(a, stop)]
            res = []
            rest_bytes = bytes(due_next)
            for tag, v in reversed(seq):
                if tag == "def":
                    b = out_def
                    if rest_bytes and b and rest_bytes.endswith(b):
                        rest_bytes = rest_bytes[:len(rest_bytes) - len(b)]; b = b""
                    res.append(b)
                    continue
                kd = pend_kinds[v] if tag == "p" else kinds[v]
                u = a - K + v if tag == "p" else v
                cst = self.text(kd >> 1)
                if rest_bytes and cst:
                    assert rest_bytes.endswith(cst), "the next lane's constants due are not the last constants of the stretch"
                    rest_bytes = rest_bytes[:len(rest_bytes) - len(cst)]
                    cst = b""
                res.append((data[u:u + 1] if kd & 1 else b"") + cst)
            if rest_bytes:
                return None
            return b"".join(reversed(res)), end, None
        body = b"".join(b for _, b in items)
        return body, e_out, self.sos(qs[e_out])

    def run(self, data, cuts):
        """cuts: sorted positions (> 0) whose SST state is known (k_sync's points).  -> output bytes, or ('fail', pos), or None when a
        stretch cannot be resolved in place (the shard would fall back)."""
        st, K, J, n = self.st, self.K, self.J, len(data)
        q, qs = st.q0, []
        for b in data:
            qs.append(q)
            q = int(st.delta[q, st.cls[b]])
            if q == 0xFFFF:
                break
        starts = [(0, self.info.start_handle)]                             # (own start B, handle there: behind the warm-up)
        for p in cuts:
            if p >= len(qs):
                continue
            B = (p + K + J + 63) & ~63
            if not (B < n and B > starts[-1][0] and self.sos(qs[p]) != 0xFFFF):
                continue
            h = self.sos(qs[p])
            for t in range(p, B):                                           # warm-up: what these steps write is the lane before's
                h, _ = self.step(h, data[t])
                if h in (self.info.dead_handle, self.info.escape_handle):
                    break
            else:                                                           # (an unclean warm-up makes no part: the lane before runs through)
                starts.append((B, h))
        out = bytearray()
        for li, (B, h) in enumerate(starts):
            last = li + 1 == len(starts)
            end = n if last else starts[li + 1][0]
            nxt = None if last else starts[li + 1][1]
            pos = B
            while pos < end:
                pl = min(64, end - pos)
                h0, piece = h, bytearray()
                bad = None
                for t in range(pos, pos + pl):
                    h, hi = self.step(h, data[t])
                    if h == self.info.dead_handle:
                        bad = ("fail", t); break
                    if h == self.info.escape_handle:
                        bad = "esc"; break
                    if not hi & 1:
                        piece.append(data[t - K])
                    ln = (hi >> 24) - (0 if hi & 1 else 1)
                    if ln:
                        off = self.info.off_pool + ((hi >> 10) & 0x1FFF) * 16
                        piece += self.img[off:off + ln]
                if bad is None:
                    out += piece; pos += pl
                    continue
                if bad != "esc":
                    return bad
                res = self.slow(data, pos, h0, end, last, nxt)
                if res is None or (isinstance(res, tuple) and res[0] == "fail"):
                    return res
                body, pos, h = res
                out += body
            if last:
                if int(st.fin_leaf[qs[n - 1]] if False else 0) and False:
                    pass
        # the tail: what the end state still owes (the host's part)
        if h is None:
            return None
        i = self.idx(h)
        qn = self.pend(i, 0)[0]
        fl = int(st.fin_leaf[qn])
        if fl == 0xFF:
            return ("fail", n)
        tail = bytearray(self.due(i))
        for j in range(K):
            ks = self.pend(i, j)[1]
            kd = ks[0] if len(ks) == 1 else ks[fl]
            tail += self.kind_bytes(kd, data, n - K + j)
        return bytes(out + tail)


def test_slow_path_model_against_the_oracle():
    """The lanes of the forward pass restated in Python — parts that begin at arbitrary known positions, the table, and the exact slow
    path for pieces with an undecided context — give the oracle's bytes on logs with escaped quotes in 1 line of 30, of 3, and in
    every line, whatever the cuts (parts of 64 bytes up to one part for the whole input), for windows J = 0, 1, 2."""
    blob = blob_of("apache_log")
    r = random.Random(3)
    base = workloads.generate("apache_log", 40000, 13)
    for J in (0, 1, 2):
        lanes = _Lanes(blob, 2, J)
        for every in (30, 3, 1):
            data, hit = _with_escaped_quotes(base, every, r)
            assert hit > 0
            want = expect(blob, data)
            assert not isinstance(want, tuple)
            for ncuts in (0, 5, 60, 400):
                cuts = sorted(r.sample(range(1, len(data) - 1), ncuts))
                got = lanes.run(data, cuts)
                assert got is not None, (J, every, ncuts, "a stretch was not resolvable in place")
                assert got == want, (J, every, ncuts)
        # the last line holds the context; truncated inputs fail where the oracle fails
        lines = base[:6000].split(b"\n")[:-1]
        lines[-1] = lines[-1].replace(b' HTTP/', b'\\"   5x HTTP/', 1)
        data = b"\n".join(lines) + b"\n"
        for cut in (0, 1, 7, 40):
            d = data[:len(data) - cut]
            got = lanes.run(d, sorted(r.sample(range(1, len(d) - 1), 20)))
            assert got == expect(blob, d), (J, cut)


@pytest.mark.gpu
def test_undecided_contexts_are_resolved_by_the_lane_that_meets_them():
    """Round 6: a context that the delay does not decide no longer sends the shard to the general engine.  The lane of the forward pass
    that meets it resolves the stretch itself on the path form (anchors forward, rows backward), hands the pieces' output to the placing
    kernel as side entries and takes the table up again behind it.  1, 10, about 1 000 such lines and (nearly) every line; segments of
    64 bytes (a stretch crosses many lanes' parts: the hand-over at a part's end), 4 KiB and the default; the last line (the stretch
    runs to the end of the input); damaged inputs rejected at the oracle's position; windows (the end of a window that is not the
    last: that window falls back, nothing else)."""
    blob = blob_of("apache_log")
    base = workloads.generate("apache_log", 4 << 20, 41)
    r = random.Random(7)
    nlines = base.count(b"\n")
    for every in (nlines, nlines // 10, max(1, nlines // 1000), 2):
        data, hit = _with_escaped_quotes(base, every, r)
        if every == nlines and hit == 0:
            lines = base.split(b"\n"); lines[nlines // 2] = lines[nlines // 2].replace(b' HTTP/', b'\\"   5x HTTP/', 1); data = b"\n".join(lines); hit = 1
        want = expect(blob, data)
        assert not isinstance(want, tuple), (every, want)
        for seg in (0, 4096, 64):
            if seg == 64 and len(data) > (1 << 20):
                data_s, want_s = data[:data.rfind(b"\n", 0, 1 << 20) + 1], None
                want_s = expect(blob, data_s)
            else:
                data_s, want_s = data, want
            p = Program(blob, segment_bytes=seg)
            try:
                assert p.run_host(data_s) == want_s, (every, seg)
                # (dense contexts — 1 line in 17, every second line: the run stays exact; where more than an eighth of the input went
                #  through the slow path the stage takes the general engine for what follows, 3)
                st_ = p.stage_delayed_form(0)
                assert st_ == 1 if every > 100 else st_ in (1, 2, 3), (every, seg, st_, "the shard fell back")
            finally:
                p.close()
    # the last line holds the undecided context: the stretch runs to the end of the input, where the final state's leaf decides
    lines = base[:200000].split(b"\n")[:-1]
    lines[-1] = lines[-1].replace(b' HTTP/', b'\\"   5x HTTP/', 1)
    data = b"\n".join(lines) + b"\n"
    for seg in (0, 4096):
        p = Program(blob, segment_bytes=seg)
        try:
            assert p.run_host(data) == expect(blob, data) and p.stage_delayed_form(0) == 1
            for cut in (1, 5, 30):      # a truncated last line: rejected where the oracle rejects it
                bad = data[:-cut]
                try:
                    got = p.run_host(bad)
                except MatchError as e:
                    got = ("fail", e.pos)
                assert got == expect(blob, bad), cut
                p.reset_delayed_form(0)
        finally:
            p.close()
    # a damaged byte inside a stretch
    data, _ = _with_escaped_quotes(base[:1 << 20], 3, r)
    i = data.find(b'\\"')
    bad = data[:i + 4] + b"\n" + data[i + 5:]
    for env in ({}, {"KX_NO_SLOW": 1}, {"KX_DF": 0}):
        got, _ = _run(blob, bad, **env)
        assert got == expect(blob, bad), env
    got, state = _run(blob, data, KX_NO_SLOW=1)     # (round 5's way: the shard falls back; same bytes)
    assert got == expect(blob, data) and state in (2, 3)
