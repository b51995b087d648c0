"""One-off soak of the exact slow path (round 6): apache_log inputs with escaped quotes at random densities, through the engine with random
segment sizes, through produced binaries with random windows, and sharded over 2-4 ranks on one device — against the oracle."""
import os, sys, random, subprocess, tempfile, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from conftest import blob_of
from kleenexlang_amd import MatchError, Program, build, program_path, workloads
from oracle import oracle

PATS = [b'\\"   5x HTTP/', b'\\" 7a\\" 33b HTTP/', b'\\" HTTP/', b'\\\\\\" HTTP/', b'\\"\\"\\" 200 1x HTTP/', b'x\\" 404 9 \\"y HTTP/']
blob = blob_of("apache_log")
tmp = tempfile.mkdtemp()
exe = os.path.join(tmp, "apache_log")
assert subprocess.run([os.path.join(build.OUT, "kexc"), "compile", "--quiet", program_path("apache_log"), "--out", exe]).returncode == 0
t0 = time.time(); runs = bad = 0; states = [0, 0, 0, 0]
seed = int(os.environ.get("SOAK_LO", 1))
while time.time() - t0 < float(os.environ.get("SOAK_SECONDS", 600)):
    r = random.Random(seed); seed += 1
    n = r.choice([20000, 70000, 300000, 1 << 20, 3 << 20])
    base = workloads.generate("apache_log", n, seed)
    lines = base.split(b"\n")
    every = r.choice([2, 5, 30, 300, 3000])
    for k in range(len(lines) - 1):
        if r.randrange(every) == 0 and b' HTTP/' in lines[k]:
            lines[k] = lines[k].replace(b' HTTP/', r.choice(PATS), 1)
    if r.random() < 0.3 and len(lines) > 2:
        lines[-2] = lines[-2].replace(b' HTTP/', r.choice(PATS), 1)      # the last line
    data = b"\n".join(lines)
    if r.random() < 0.15:
        i = r.randrange(len(data)); data = data[:i] + bytes([r.choice(b'\x00"\\\n x')]) + data[i + 1:]   # a damaged byte
    if r.random() < 0.1:
        data = data[:r.randrange(len(data))]                                                              # a truncated input
    try:
        want = oracle.run(blob, data)
    except oracle.OracleMatchError as e:
        want = ("fail", e.pos)
    for seg in r.sample([0, 64, 256, 1024, 4096, 16384], 2):
        p = Program(blob, segment_bytes=seg)
        try:
            for _ in range(2):      # (the second run is on the armed instance)
                try:
                    got = p.run_host(data)
                except MatchError as e:
                    got = ("fail", e.pos)
                runs += 1
                if got != want:
                    bad += 1
                    print("MISMATCH engine seed", seed - 1, "seg", seg, "len", len(data), flush=True)
            states[p.stage_delayed_form(0)] += 1
        finally:
            p.close()
    if not isinstance(want, tuple):
        win = r.choice([4096, 20000, 65536, 300000])
        q = subprocess.run([exe], input=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, KX_WINDOW_BYTES=str(win)))
        runs += 1
        if q.returncode != 0 or q.stdout != want:
            bad += 1; print("MISMATCH windows seed", seed - 1, "window", win, "len", len(data), q.returncode, flush=True)
        src = os.path.join(tmp, "in"); open(src, "wb").write(data)
        g = r.choice([2, 3, 4])
        with open(src, "rb") as fin:
            q = subprocess.run([exe, "--gpus", str(g)], stdin=fin, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, KX_SHARD_SAME_DEVICE="1"))
        runs += 1
        if q.returncode != 0 or q.stdout != want:
            bad += 1; print("MISMATCH shards seed", seed - 1, "ranks", g, "len", len(data), q.returncode, q.stderr[-200:], flush=True)
print("runs", runs, "mismatches", bad, "seeds up to", seed - 1, "states after (none / on the form / backing off / given up)", states, "time", round(time.time() - t0))
