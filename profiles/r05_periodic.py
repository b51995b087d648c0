"""Round 5 experiment: what conflict-free table reads would be worth.  An apache log whose every line is the SAME 128 bytes: all lanes of a
wave walk the same states in step (every class / transition read is a broadcast), against the seeded synthetic log.  Output unchecked.
  python profiles/r05_periodic.py"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from kleenexlang_amd import Program, compile_file, workloads
from oracle import oracle
blob = compile_file("apache_log")
line = b'61.66.189.242 - - [21/Oct/2010:19:00:58 -0400] "POST /8+8Yt/t%Xbi HTTP/1.0" 200 4064 "-" "curl/7.35.0"\n'
pad = 128 - len(line)
line = line.replace(b"/8+8Yt/", b"/8+8Yt" + b"x" * pad + b"/")
assert len(line) == 128
oracle.run(blob, line * 3)
res = {}
for name, base in (("periodic-128", line * (1 << 18)), ("synthetic", workloads.generate("apache_log", 32 << 20, seed=1))):
    reps = (10 << 30) // len(base)
    t = torch.frombuffer(bytearray(base), dtype=torch.uint8).to("cuda:0").repeat(reps)
    p = Program(blob, collect_timing=True)
    out = torch.empty(p.out_capacity(t.numel()), dtype=torch.uint8, device="cuda:0")
    best = None
    for i in range(5):
        p.run_tensor(t, out)
        k = p.last_stats.as_dict()["kernel_ms"]
        if i and (best is None or sum(k.values()) < sum(best.values())): best = k
    res[name] = {a: round(b, 3) for a, b in best.items()}
    del t, out
print(json.dumps(res))
