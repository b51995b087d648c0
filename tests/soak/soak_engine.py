"""One-off soak: random programs x long accepted inputs x segment sizes on the engine vs the oracle."""
import os, sys, random, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import randprog
from conftest import blob_of
from kleenexlang_amd import MatchError, Program, CompileError
from oracle import oracle

STAR = False
t0 = time.time(); n = bad = 0; acc = 0; outb = 0; dfs = [0, 0, 0, 0]   # runs by the stage's state afterwards: no delayed form / on it / backing off after a fall-back / given up
for seed in range(int(os.environ.get("SOAK_LO", 300)), int(os.environ.get("SOAK_HI", 700))):
    src = randprog.program(seed)
    if STAR:   # make the whole program repeatable so that long accepted inputs exist: main := (old main "|")* with old main renamed
        src = src.replace("main :=", "kxbody :=", 1) + "\nmain := (kxbody /;/)*\n"
    try:
        if oracle.info(blob_of(src, 0))["nstates"] > 1500:
            continue
        blob = blob_of(src, 3)
    except CompileError:
        continue
    # long inputs: concatenate short random inputs, keep those the oracle accepts
    rng = random.Random(seed)
    pool = randprog.inputs(seed, 40, 60)
    lines = set()
    for x in pool:
        for ln in x.split(b"\n"):
            lines.add(ln + b"\n")
    for _ in range(200):
        lines.add(bytes(rng.choice(b"abc") for _ in range(rng.randint(0, 12))) + b"\n")
    good = []
    for ln in sorted(lines):
        try:
            oracle.run(blob, ln); good.append(ln)
        except oracle.OracleMatchError:
            pass
    pool = good or [b""]
    cands = []
    for k in (50, 400, 3000):
        cands.append(b"".join(rng.choice(pool) for _ in range(k)))
    for seg in (64, 1024, 4096):
        try:
            p = Program(blob, segment_bytes=seg)
        except Exception:
            break
        for data in cands + pool[:4]:
            try:
                want = oracle.run(blob, data)
            except oracle.OracleMatchError as e:
                want = ("fail", e.pos)
            try:
                got = p.run_host(data)
            except MatchError as e:
                got = ("fail", e.pos)
            n += 1
            dfs[p.stage_delayed_form(0)] += 1
            if not isinstance(want, tuple): acc += 1; outb += len(want)
            if got != want:
                bad += 1
                print("MISMATCH seed", seed, "seg", seg, "len", len(data), "want", (want if isinstance(want, tuple) else len(want)), "got", (got if isinstance(got, tuple) else len(got)), flush=True)
                print(src)
        p.close()
    if time.time() - t0 > 500:
        break
print("accepted", acc, "output bytes", outb, "runs on the general engine / the delayed form / after a fall-back", dfs); print("runs", n, "mismatches", bad, "seeds up to", seed, "time", round(time.time() - t0))
