"""One-off soak of the regex flavour on the engine: random regexes (the generator of tests/test_regex_coder.py) under an outer
star x strings of their own language x segment sizes, with and without --la — engine (table atoms, inline-constant layout,
per-entry table ids) against the CPU oracle, every byte."""
import os, sys, random, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from test_regex_coder import _random_regex, greedy_code
from kleenexlang_amd import host
from oracle import oracle

rnd = random.Random(int(os.environ.get("SOAK_SEED", 9001)))
t0 = time.time(); runs = bad = progs = 0; layouts = {"inl": 0, "plain": 0}
for it in range(int(os.environ.get("SOAK_N", 400))):
    inner = _random_regex(rnd)
    regex = "((" + inner + ")x)*"
    try:
        blob = host.compile_flags(regex, opt=rnd.choice([0, 3]), la=rnd.random() < 0.3, regex=True)
    except Exception:
        continue
    words = []
    for _ in range(300):
        w = bytes(rnd.choice(b"abc") for _ in range(rnd.randrange(0, 7)))
        if greedy_code(inner, w) is not None:
            words.append(w)
    if not words:
        continue
    progs += 1
    for env in ({}, {"KX_INL": "0"}, {"KX_FORCE_TBLMODE": "1"}):
        for k, v in env.items(): os.environ[k] = v
        try:
            prog = host.Program(blob, segment_bytes=rnd.choice([0, 64, 4096, 16384]))
        except Exception:
            for k in env: del os.environ[k]
            continue
        for size in (0, 1, 60, 5000, 60000):
            data = b"".join(rnd.choice(words) + b"x" for _ in range(size))
            want = oracle.run(blob, data)
            got = prog.run_host(data)
            runs += 1
            if got != want:
                bad += 1
                print("MISMATCH", regex, env, size, len(want), len(got), flush=True)
        prog.close()
        for k in env: del os.environ[k]
print("coder programs", progs, "runs", runs, "mismatches", bad, "time %d" % (time.time() - t0))
