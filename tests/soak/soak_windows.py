"""One-off soak of the windowed stdin->stdout path: random programs, ~100 KB accepted inputs, tiny windows."""
import os, sys, random, subprocess, tempfile, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import randprog
from conftest import blob_of
from kleenexlang_amd import build, CompileError
from oracle import oracle
kexc = os.path.join(build.OUT, "kexc")
tmp = tempfile.mkdtemp()
t0 = time.time(); n = bad = 0
for seed in range(1000, 1400):
    src = randprog.program(seed)
    try:
        if oracle.info(blob_of(src, 0))["nstates"] > 800:
            continue
        blob = blob_of(src, 3)
    except CompileError:
        continue
    rng = random.Random(seed)
    lines = set()
    for x in randprog.inputs(seed, 40, 60):
        for ln in x.split(b"\n"):
            lines.add(ln + b"\n")
    good = []
    for ln in sorted(lines):
        try:
            oracle.run(blob, ln); good.append(ln)
        except oracle.OracleMatchError:
            pass
    if len(good) < 2:
        continue
    data = b"".join(rng.choice(good) for _ in range(20000))
    try:
        want = oracle.run(blob, data)
    except oracle.OracleMatchError:
        continue
    kex = os.path.join(tmp, "p.kex"); exe = os.path.join(tmp, "p.bin")
    open(kex, "w").write(src)
    if subprocess.run([kexc, "compile", "--quiet", kex, "--out", exe]).returncode != 0:
        continue
    for window in (4096, 50000):
        r = subprocess.run([exe], input=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, KX_WINDOW_BYTES=str(window)))
        n += 1
        if r.returncode != 0 or r.stdout != want:
            bad += 1
            print("MISMATCH seed", seed, "window", window, "rc", r.returncode, len(r.stdout), len(want), r.stderr[-100:], flush=True)
            print(src)
    if time.time() - t0 > 400:
        break
print("windowed runs", n, "mismatches", bad, "seeds up to", seed, "time", round(time.time() - t0))
