#!/usr/bin/env python3
"""Regex flavour on the device: the coder of a CSV row regex (shape of the reference's bench/regex_src/csv_project3.rx under
an outer star) over GiBs resident in HBM — kernel times from the engine's HIP events, every output byte checked on the
device against the CPU oracle's code of one base chunk.  usage: python profiles/coder_bench.py [GiB]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kleenexlang_amd import host, workloads  # noqa: E402
from oracle import oracle  # noqa: E402

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
regex = "(([^,\\n]*),([^,\\n]*),([^,\\n]*),([^,\\n]*),([^,\\n]*),([^,\\n]*)\\n)*"
blob = host.compile_regex(regex)
info = oracle.info(blob)
base = workloads.generate("csv", 32 << 20, seed=0x4B4C4558)
want = oracle.run(blob, base)
assert want[-1] == 1
unit = torch.frombuffer(bytearray(want[:-1]), dtype=torch.uint8).cuda()
k = max(1, int(gib * (1 << 30)) // len(base))
t = torch.frombuffer(bytearray(base), dtype=torch.uint8).cuda().repeat(k)
out = torch.empty(int(t.numel() * 2.1) + (1 << 20), dtype=torch.uint8, device="cuda")
prog = host.Program(blob, collect_timing=True)
stream = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    olen = prog.run_device(t.data_ptr(), t.numel(), out.data_ptr(), out.numel(), stream)
torch.cuda.synchronize()
steps, kern = 5, {}
t0 = time.perf_counter()
for _ in range(steps):
    olen = prog.run_device(t.data_ptr(), t.numel(), out.data_ptr(), out.numel(), stream)
    for name, ms in prog.last_stats.as_dict()["kernel_ms"].items():
        kern[name] = kern.get(name, 0.0) + ms / steps
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
m = unit.numel()
ok = olen == k * m + 1 and bool((out[:k * m].view(k, m) == unit).all().item()) and int(out[k * m].item()) == 1
print(json.dumps({"regex": regex, "tables": info, "input_bytes": t.numel(), "output_bytes": olen, "ms_per_step": round(dt * 1e3, 3),
                  "input_GBps": round(t.numel() / dt / 1e9, 1), "kernels_ms": {a: round(b, 3) for a, b in kern.items()},
                  "output_checked_bit_exact": ok, "output_bytes_checked": olen}))
