"""One-off soak (round 5): generated substitution programs with MANY byte classes (33-90: the delayed form's class table holds indices and
the sequences shift) and output/input ratios from 1 to 12 (beyond ~2.4 a lane takes HALF a piece), long accepted inputs, three segment
sizes, the engine against the oracle.  SOAK_LO / SOAK_HI: seed range."""
import os, sys, random, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from conftest import blob_of
from kleenexlang_amd import MatchError, Program, CompileError
from oracle import oracle

ALPHA = "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789!#$%&'()*+,-.:;<=>?@[]^_`{|}~ "
t0 = time.time(); n = bad = 0; dfs = [0, 0, 0]; outb = 0
for seed in range(int(os.environ.get("SOAK_LO", 0)), int(os.environ.get("SOAK_HI", 200))):
    r = random.Random(seed)
    k = r.randint(33, 90)
    letters = r.sample(ALPHA.replace("/", "").replace("\\", "").replace('"', ""), k)
    esc = lambda c: "\\" + c if c in "()[]{}|*+?.^$-/" else c
    alts = []
    for c in letters[:-3]:
        w = r.choice([0, 1, 1, 2, 3, 5, 8, 12])
        body = "".join(r.choice("xyzXYZ01") for _ in range(w))
        alts.append(('~/%s/ "%s"' if r.random() < 0.7 else '/%s/ "%s"') % (esc(c), body))
    rest = "".join(esc(c) for c in letters[-3:])
    tail = r.choice(["", ' "!" ', ' ";\\n" '])
    src = "main := ((%s | /[%s]/)%s)*\n" % (" | ".join(alts), rest, tail)
    try:
        blob = blob_of(src, 3)
    except CompileError as e:
        continue
    cands = [bytes(ord(r.choice(letters)) for _ in range(m)) for m in (1, 63, 64, 65, 5000, 70001, 400000)]
    cands.append(cands[-1][:100000] + b"\x00" + cands[-1][100001:200000])   # rejected in the middle
    for seg in (64, 4096, 0):
        p = Program(blob, segment_bytes=seg)
        for data in cands:
            try:
                want = oracle.run(blob, data)
            except oracle.OracleMatchError as e:
                want = ("fail", e.pos)
            try:
                got = p.run_host(data)
            except MatchError as e:
                got = ("fail", e.pos)
            n += 1; dfs[p.stage_delayed_form(0)] += 1
            if not isinstance(want, tuple): outb += len(want)
            if got != want:
                bad += 1
                print("MISMATCH seed", seed, "seg", seg, "len", len(data), "classes", k, flush=True)
                print(src)
        p.close()
    if time.time() - t0 > 600:
        break
print("output bytes", outb, "runs on the general engine / the delayed form / after a fall-back", dfs)
print("runs", n, "mismatches", bad, "seeds up to", seed, "time", round(time.time() - t0))
