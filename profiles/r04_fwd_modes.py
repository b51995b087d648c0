#!/usr/bin/env python3
"""k_forward takes 2.75 or 3.03 ms per 10 GiB from one process to the next (every engine, every box): what decides?  One fresh process
per line: addresses of the input and output tensors and the per-kernel times."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from kleenexlang_amd import Program, compile_file, workloads
pad = int(sys.argv[1]) if len(sys.argv) > 1 else 0
dev = torch.device("cuda", 0)
if pad:
    _hold = torch.empty(pad, dtype=torch.uint8, device=dev)      # shifts where the next allocations land
prog = Program(compile_file("apache_log"), collect_timing=True)
base = workloads.generate("apache_log", 32 << 20)
tb = torch.frombuffer(bytearray(base), dtype=torch.uint8).to(dev)
t = tb.repeat((10 << 30) // len(base))
out = torch.empty(int(t.numel() * 1.3) + (1 << 20), dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream(dev).cuda_stream
ks = []
for i in range(4):
    prog.run_device(t.data_ptr(), t.numel(), out.data_ptr(), out.numel(), st)
    ks.append(prog.last_stats.as_dict()["kernel_ms"])
print(json.dumps({"pad": pad, "in_ptr": hex(t.data_ptr()), "out_ptr": hex(out.data_ptr()), "in_mod_2M": t.data_ptr() % (2 << 20), "forward_ms": [round(k["forward"], 3) for k in ks],
                  "backlen_ms": [round(k["backlen"], 3) for k in ks], "emit_ms": [round(k["emit"], 3) for k in ks]}))
