"""Round 5: kernel times of one program on tiled input WITHOUT checking the output (ablation switches in KX_DEBUG_FLAGS give wrong
bytes on purpose).  python profiles/r05_ablate.py PROGRAM GIB [label]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from kleenexlang_amd import Program, compile_file, workloads
prog = sys.argv[1]; gib = float(sys.argv[2]); label = sys.argv[3] if len(sys.argv) > 3 else ""
base = workloads.generate(workloads.PROGRAM_INPUT[prog], 32 << 20, seed=1)
reps = max(1, int(gib * (1 << 30)) // len(base))
t = torch.frombuffer(bytearray(base), dtype=torch.uint8).to("cuda:0").repeat(reps)
p = Program(compile_file(prog), collect_timing=True)
out = torch.empty(p.out_capacity(t.numel()), dtype=torch.uint8, device="cuda:0")
best = None
for i in range(5):
    p.run_tensor(t, out)
    k = p.last_stats.as_dict()["kernel_ms"]
    tot = sum(k.values())
    if i and (best is None or tot < best[0]): best = (tot, k)
print(json.dumps({"label": label, "flags": os.environ.get("KX_DEBUG_FLAGS", "0"), "program": prog, "gib": gib, "total_ms": round(best[0], 3),
                  "kernel_ms": {a: round(b, 3) for a, b in best[1].items()}, "GBps": round(t.numel() / best[0] / 1e6, 1)}))
