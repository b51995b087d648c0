"""Regex flavour (`kexc compile FILE.re|FILE.rx`, `--re EXPR`): the bit-coder.

The reference turns the regex's transducer into its ORACLE (`SymbolicFST/OracleMachine.hs:47-61`): symbol outputs are
dropped, the k-th of n > 1 ε-alternatives writes the fixed-width code of k, a copied symbol from a predicate with more
than one member writes the code of its index in the predicate.  The front end instantiates the digit type with Word8
(`Frontend.hs:117`), so a code is one byte per choice.  The reference holds no expected outputs for this flavour (only
`bench/regex_src/{as,csv_project3}.rx`, sources without vectors), so the pin here is a restatement written down a second
time and independently: `greedy_code` below is a backtracking matcher over its own little regex AST that records the
choices of the FIRST successful parse in priority order — which is what the lock-step simulation with "earlier path
wins" computes (`SymbolicFST.hs:361-380`).  Routes compared: that model, the lock-step simulation of the oracle machine
(`oracle/fst_sim.py`), the register-form and path-form CPU oracle on the compiled blob — and, on the GPU, the engine.
The association of `a|b|c` (right-nested) and the syntax itself come from the un-vendored `regexps-syntax` package:
parity unpinned beyond the shapes the reference's `.kex`/`.rx` files use."""
import os
import random
import subprocess

import pytest

from kleenexlang_amd import host
from oracle import fst_sim, oracle


# ------------------------------------------------------------------ the independent model
class _P:
    """Regex subset → AST tuples.  ('chr', b) ('set', frozenset) ('cat', a, b) ('alt', a, b) ('star', e, lazy)
    ('plus', e, lazy) ('opt', e, lazy) ('rep', e, lo, hi|None) ('one',)"""

    def __init__(self, s):
        self.s, self.i = s, 0

    def peek(self):
        return self.s[self.i] if self.i < len(self.s) else None

    def esc(self):
        c = self.s[self.i]
        self.i += 1
        return {"n": 10, "t": 9, "r": 13}.get(c, ord(c))

    def alt(self):
        left = self.cat()
        if self.peek() == "|":
            self.i += 1
            return ("alt", left, self.alt())
        return left

    def cat(self):
        acc = None
        while self.peek() is not None and self.peek() not in "|)":
            r = self.rep()
            acc = r if acc is None else ("cat", acc, r)
        return acc if acc is not None else ("one",)

    def rep(self):
        r = self.atom()
        while True:
            c = self.peek()
            if c in ("*", "+", "?"):
                self.i += 1
                lazy = self.peek() == "?"
                if lazy:
                    self.i += 1
                r = ({"*": "star", "+": "plus", "?": "opt"}[c], r, lazy)
            elif c == "{":
                j = self.s.index("}", self.i)
                body = self.s[self.i + 1:j]
                self.i = j + 1
                if "," in body:
                    lo, hi = body.split(",")
                    r = ("rep", r, int(lo), int(hi) if hi else None)
                else:
                    r = ("rep", r, int(body), int(body))
            else:
                return r

    def atom(self):
        c = self.peek()
        self.i += 1
        if c == ".":
            return ("set", frozenset(range(256)))
        if c == "(":
            r = self.alt()
            assert self.peek() == ")"
            self.i += 1
            return r
        if c == "[":
            neg = self.peek() == "^"
            if neg:
                self.i += 1
            members = set()
            while self.peek() != "]":
                lo = hi = self._member()
                if self.peek() == "-" and self.s[self.i + 1] != "]":
                    self.i += 1
                    hi = self._member()
                members.update(range(lo, hi + 1))
            self.i += 1
            return ("set", frozenset(set(range(256)) - members if neg else members))
        if c == "\\":
            return ("chr", self.esc())
        return ("chr", ord(c))

    def _member(self):
        if self.s[self.i] == "\\":
            self.i += 1
            return self.esc()
        self.i += 1
        return ord(self.s[self.i - 1])


def greedy_code(regex, data):
    """The code of the first parse of `data` (bytes) in priority order, or None.  Desugaring.hs:70-118 + OracleMachine.hs:54-61."""
    ast = _P(regex).alt()

    def m(node, i, code, k):
        kind = node[0]
        if kind == "one":
            return k(i, code)
        if kind == "chr":
            return k(i + 1, code) if i < len(data) and data[i] == node[1] else None
        if kind == "set":
            if i >= len(data) or data[i] not in node[1]:
                return None
            members = sorted(node[1])
            return k(i + 1, code + (bytes([members.index(data[i])]) if len(members) > 1 else b""))
        if kind == "cat":
            return m(node[1], i, code, lambda j, c: m(node[2], j, c, k))
        if kind == "alt":
            r = m(node[1], i, code + b"\0", k)
            return r if r is not None else m(node[2], i, code + b"\1", k)
        if kind == "opt":
            take, skip = (b"\1", b"\0") if node[2] else (b"\0", b"\1")
            order = [(skip, None), (take, node[1])] if node[2] else [(take, node[1]), (skip, None)]
            for tag, body in order:
                r = k(i, code + tag) if body is None else m(body, i, code + tag, k)
                if r is not None:
                    return r
            return None
        if kind == "star":
            loop, leave = (b"\1", b"\0") if node[2] else (b"\0", b"\1")

            def again(j, c):
                return m(node, j, c, k) if j > i else None   # (the tests use no nullable loop bodies)
            first = (lambda: k(i, code + leave)) if node[2] else (lambda: m(node[1], i, code + loop, again))
            second = (lambda: m(node[1], i, code + loop, again)) if node[2] else (lambda: k(i, code + leave))
            r = first()
            return r if r is not None else second()
        if kind == "plus":
            return m(node[1], i, code, lambda j, c: m(("star", node[1], node[2]), j, c, k))
        if kind == "rep":
            _, e, lo, hi = node
            # Desugaring.hs:106-115: n mandatory copies, then — when n != m — m OPTIONAL ones (`replicate m' iquest`, not m - n)
            seq = [e] * lo + ([("star", e, False)] if hi is None else [] if hi == lo else [("opt", e, False)] * hi)
            if not seq:
                return k(i, code)
            tree = seq[0]
            for x in seq[1:]:
                tree = ("cat", tree, x)
            return m(tree, i, code, k)
        raise AssertionError(kind)

    return m(ast, 0, b"", lambda j, c: c if j == len(data) else None)


# ------------------------------------------------------------------ cases
CASES = [
    ("a*", [b"", b"a", b"aaaa", b"b"]),                                       # ref: bench/regex_src/as.rx
    ("(a|b)*c", [b"c", b"abbac", b"ab", b""]),
    ("a|b|c", [b"a", b"b", b"c", b"d"]),
    ("[a-c]*x?", [b"", b"abcx", b"cab", b"x", b"xx"]),
    ("(a*)(a|b)(b*)", [b"a", b"aab", b"b", b"abbb", b""]),
    ("(ab|a)(bc|c)?", [b"abc", b"ab", b"a", b"abbc", b"ac"]),
    ("a*?b+?b*", [b"ab", b"aabbb", b"b", b"a"]),
    ("x{2}y{1,3}z{2,}", [b"xxyzz", b"xxyyyzzzz", b"xyzz", b"xxyyyyzz"]),
    (".[^a]", [b"zb", b"\x00\xff", b"za", b"z"]),
    ("a*([^,\\n]*),([^,\\n]*)\\n", [b"aab,cd\n", b",\n", b"x,y,z\n", b"q\n"]),   # shape of bench/regex_src/csv_project3.rx
    ("", [b"", b"a"]),
    # more distinct multi-member predicates than a path entry can name tables (KXP_ENGINE_TABLES = 7): the largest keep their
    # table atoms, the others are written out symbol by symbol (this regex compiled and then did not load)
    ("[a-c][d-f][g-i][j-l][m-o][p-r][s-u][v-x][0-4][5-9]x*", [b"adgjmpsv05", b"cfilorux49xx", b"beh", b"adgjmpsv0"]),
]


@pytest.mark.parametrize("regex,inputs", CASES)
def test_coder_routes_agree(regex, inputs):
    fst = host.dump_regex_fst(regex, oracle=True)
    for opt in (0, 3):
        blob = host.compile_regex(regex, opt=opt)
        host.validate_blob(blob)
        for data in inputs:
            want = greedy_code(regex, data)
            assert fst_sim.run(fst, data) == want, (regex, data)
            for pf in (False, True):
                if want is None:
                    with pytest.raises(oracle.OracleMatchError):
                        oracle.run(blob, data, path_form=pf)
                else:
                    assert oracle.run(blob, data, path_form=pf) == want, (regex, opt, pf, data)


def test_codes_are_what_the_oracle_machine_says():
    """Spot checks of the meaning itself (OracleMachine.hs:54-61; Util/Coding.hs: one base-256 digit up to 256 choices)."""
    run = lambda rx, s: oracle.run(host.compile_regex(rx), s)
    assert run("a*", b"aaa") == b"\0\0\0\1"            # star = RSum [loop, exit] (Desugaring.hs:84-88)
    assert run("a*?", b"aa") == b"\1\1\0"              # lazy star = RSum [exit, loop]
    assert run("a|b", b"b") == b"\1"
    assert run("abc", b"abc") == b""                   # a deterministic parse has the empty code
    assert run("[a-d]", b"c") == b"\2"                 # CodeArg: index within the predicate
    assert run("[k]", b"k") == b""                     # singleton predicate: CodeConst []
    assert run(".", b"\xfe") == b"\xfe"                # 256 members: the code of a symbol is the symbol
    assert run("(a|ab)(c|bcd)(d*)", b"abcd") == b"\0\1\1"   # greedy: a, then bcd, then d* leaves at once


def _random_regex(rnd, depth=0):
    r = rnd.random()
    if depth > 3 or r < 0.25:
        return rnd.choice(["a", "b", "c", "[ab]", "[a-c]", "[^a]", "."])
    if r < 0.5:
        return _random_regex(rnd, depth + 1) + _random_regex(rnd, depth + 1)
    if r < 0.7:
        return "(" + _random_regex(rnd, depth + 1) + "|" + _random_regex(rnd, depth + 1) + ")"
    atom = rnd.choice(["a", "b", "[ab]", "[a-c]", "(a|bc)", "(ab|b)", "."])   # never nullable
    return atom + rnd.choice(["*", "+", "?", "*?", "+?", "??", "{2}", "{1,2}", "{1,}"])


def test_random_regexes_all_routes():
    rnd = random.Random(20260929)
    checked = accepted = 0
    for _ in range(60):
        regex = _random_regex(rnd)
        fst = host.dump_regex_fst(regex, oracle=True)
        blob = host.compile_regex(regex, opt=rnd.choice([0, 3]))
        for _ in range(12):
            data = bytes(rnd.choice(b"abc") for _ in range(rnd.randrange(0, 7)))
            want = greedy_code(regex, data)
            assert fst_sim.run(fst, data) == want, (regex, data)
            if want is None:
                with pytest.raises(oracle.OracleMatchError):
                    oracle.run(blob, data)
            else:
                accepted += 1
                assert oracle.run(blob, data) == want, (regex, data)
                assert oracle.run(blob, data, path_form=True) == want, (regex, data)
            checked += 1
    assert accepted > checked // 10


def test_cli_flavour_by_extension_and_expression_argument(tmp_path):
    """getCompileFlavor (Frontend.hs:140-152): `--re` makes the argument the expression; `.re`/`.rx` files are regexes;
    anything else but `.kex` is refused with the reference's message.  `--wordsize`: Options.hs:130-144."""
    from kleenexlang_amd import build
    kexc = os.path.join(build.OUT, "kexc")
    rx = tmp_path / "p.rx"
    rx.write_text("(a|b)*c")
    r = subprocess.run([kexc, "compile", "--quiet", str(rx), "--blob", str(tmp_path / "p.kxp")], stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr
    assert oracle.run((tmp_path / "p.kxp").read_bytes(), b"abc") == b"\0\0\0\1\1"
    r = subprocess.run([kexc, "compile", "--quiet", "--re", "(a|b)*c", "--blob", str(tmp_path / "q.kxp")], stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "q.kxp").read_bytes()[-64:] != b"" and oracle.run((tmp_path / "q.kxp").read_bytes(), b"c") == b"\1"
    bad = tmp_path / "p.txt"
    bad.write_text("a")
    r = subprocess.run([kexc, "compile", str(bad)], stderr=subprocess.PIPE, stdout=subprocess.PIPE)
    assert r.returncode == 1 and b"Expects one of '.kex', '.re', or '.rx'." in r.stderr
    r = subprocess.run([kexc, "compile", "--wordsize", "12", str(rx)], stderr=subprocess.PIPE, stdout=subprocess.PIPE)
    assert r.returncode == 1 and b'"12" is not a valid word size' in r.stderr
    r = subprocess.run([kexc, "compile", "--quiet", "--wordsize", "8", str(rx), "--blob", str(tmp_path / "w.kxp")], stderr=subprocess.PIPE)
    assert r.returncode == 0
    r = subprocess.run([kexc, "compile", "--wordsize", "16", str(rx)], stderr=subprocess.PIPE, stdout=subprocess.PIPE)
    assert r.returncode == 1 and b"crt/crt.c:143-155" in r.stderr
    slash = tmp_path / "s.re"
    slash.write_text("a/b")
    r = subprocess.run([kexc, "compile", str(slash)], stderr=subprocess.PIPE, stdout=subprocess.PIPE)
    assert r.returncode == 1 and b"'/'" in r.stderr          # rep_illegal_chars = "/" (Parser.hs:204-206)


# ------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_coder_on_the_engine():
    """The coder's tables on the HIP engine: small cases against the model, and a 64 MiB CSV through the shape of
    bench/regex_src/csv_project3.rx against the CPU oracle (every byte), across segment sizes."""
    for regex, inputs in CASES:
        prog = host.Program(host.compile_regex(regex))
        for data in inputs:
            want = greedy_code(regex, data)
            if want is None:
                with pytest.raises(host.MatchError):
                    prog.run_host(data)
            else:
                assert prog.run_host(data) == want, (regex, data)
        prog.close()
    rnd = random.Random(7)
    fields = [bytes(rnd.choice(b"abcdefghij0123456789 ") for _ in range(rnd.randrange(0, 12))) for _ in range(4096)]
    rows = b"".join(b",".join(rnd.choice(fields) for _ in range(6)) + b"\n" for _ in range(20000))
    regex = "(([^,\\n]*),([^,\\n]*),([^,\\n]*),([^,\\n]*),([^,\\n]*),([^,\\n]*)\\n)*"
    blob = host.compile_regex(regex)
    want = oracle.run(blob, rows)
    assert len(want) > len(rows) // 2
    for seg in (0, 4096, 65536):
        prog = host.Program(blob, segment_bytes=seg)
        assert prog.run_host(rows) == want
        prog.close()
    big = rows * (64 * 1024 * 1024 // len(rows))
    prog = host.Program(blob)
    got = prog.run_host(big)
    assert got == want[:-1] * (len(big) // len(rows)) + want[-1:]   # (every row's code, then the final "leave the loop")
    prog.close()


@pytest.mark.gpu
def test_coder_binary_and_simulate(tmp_path):
    """The produced binary of a `.rx` file (`BIN < in > out`, windowed) and `kexc simulate --re EXPR` (simulateCoder,
    Commands.hs:317-322) write the same code as the CPU oracle; a string outside the language is a match error."""
    from kleenexlang_amd import build
    kexc = os.path.join(build.OUT, "kexc")
    regex = "(([a-z]*|[0-9]+)(,|\\n))*"
    rnd = random.Random(11)
    data = b"".join(rnd.choice([b"abc", b"", b"0042", b"z", b"7"]) + rnd.choice([b",", b"\n"]) for _ in range(300000))
    want = oracle.run(host.compile_regex(regex), data)
    assert greedy_code(regex, data[:40] + b"\n") == oracle.run(host.compile_regex(regex), data[:40] + b"\n")
    rx = tmp_path / "fields.rx"
    rx.write_text(regex)
    binp = tmp_path / "fields_bin"
    r = subprocess.run([kexc, "compile", "--quiet", str(rx), "--out", str(binp)], stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr
    for env in ({}, {"KX_WINDOW_BYTES": "65536"}):
        r = subprocess.run([str(binp)], input=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env={**os.environ, **env})
        assert r.returncode == 0 and r.stdout == want, (env, r.stderr[-300:])
    r = subprocess.run([kexc, "simulate", "--sim", "sst", "--re", regex], input=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0 and r.stdout == want, r.stderr[-300:]
    # (the FST simulators run the regex's own transducer, which copies what it matches: simulateLockstep on tuTransducers)
    part = data[:data.rindex(b"\n", 0, 20000) + 1]
    r = subprocess.run([kexc, "simulate", "--re", regex], input=part, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0 and r.stdout == part, r.stderr[-300:]
    r = subprocess.run([str(binp)], input=b"abc,DEF\n", stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1 and b"Match error at input symbol 4!" in r.stderr, r.stderr
    r = subprocess.run([str(binp), "-i"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert b"Oracle SST states" in r.stdout, r.stdout


@pytest.mark.gpu
def test_random_regex_coders_on_the_engine():
    """Soak: random regexes (the generator of the CPU test) × strings drawn from their own language by a random walk over
    the model, blown up to KiB sizes by repetition under an outer star — engine against the CPU oracle, every byte."""
    rnd = random.Random(4242)
    runs = 0
    for _ in range(40):
        inner = _random_regex(rnd)
        regex = "((" + inner + ")x)*"
        blob = host.compile_regex(regex, opt=rnd.choice([0, 3]))
        words = []
        for _ in range(200):
            w = bytes(rnd.choice(b"abc") for _ in range(rnd.randrange(0, 6)))
            if greedy_code(inner, w) is not None:
                words.append(w)
        if not words:
            continue
        prog = host.Program(blob, segment_bytes=rnd.choice([0, 4096, 16384]))
        for size in (0, 1, 50, 3000, 40000):
            data = b"".join(rnd.choice(words) + b"x" for _ in range(size))
            assert prog.run_host(data) == oracle.run(blob, data), (regex, size)
            runs += 1
        prog.close()
    assert runs >= 100


# ------------------------------------------------------------------ the reference's own regex sources, pinned by hand
def _coder_vectors():
    import json
    from conftest import GOLDEN
    with open(os.path.join(GOLDEN, "coder_vectors.json"), encoding="utf-8") as f:
        return json.load(f)


def test_reference_regex_sources_against_hand_derived_codes():
    """the codes of bench/regex_src/as.rx and csv_project3.rx on small inputs, derived by hand from
    OracleMachine.hs:47-61 + Util/Coding.hs:13-19 + Desugaring.hs:70-118 (tests/golden/coder_derivation.md — a worked derivation,
    not a second program) — against the compiler (`--opt 0/3`, `--la` off/on), in the register form and in the path form.  This
    pins the coder on what the reference's two sources use: star, negated class, singleton reads, groups, concatenation."""
    v = _coder_vectors()
    ref_src = "/root/reference/bench/regex_src"
    if os.path.isdir(ref_src):   # the committed regex text is the reference's file, byte for byte
        for name, rx in v["sources"].items():
            assert open(os.path.join(ref_src, name + ".rx"), "rb").read().rstrip(b"\n") == rx.encode(), name
    for opt in (0, 3):
        blobs = {name: host.compile_regex(rx, opt=opt) for name, rx in v["sources"].items()}
        for t in v["vectors"]:
            data = bytes.fromhex(t["in_hex"])
            for pf in (False, True):
                if t["code_hex"] is None:
                    with pytest.raises(oracle.OracleMatchError):
                        oracle.run(blobs[t["source"]], data, path_form=pf)
                else:
                    assert oracle.run(blobs[t["source"]], data, path_form=pf) == bytes.fromhex(t["code_hex"]), (t, opt, pf)


@pytest.mark.gpu
def test_reference_regex_sources_on_the_engine():
    """…and the engine writes the hand-derived codes (tests/golden/coder_vectors.json)."""
    v = _coder_vectors()
    for name, rx in v["sources"].items():
        p = host.Program(host.compile_regex(rx))
        try:
            for t in v["vectors"]:
                if t["source"] != name:
                    continue
                data = bytes.fromhex(t["in_hex"])
                if t["code_hex"] is None:
                    with pytest.raises(host.MatchError):
                        p.run_host(data)
                else:
                    assert p.run_host(data) == bytes.fromhex(t["code_hex"]), t
        finally:
            p.close()
