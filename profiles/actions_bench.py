#!/usr/bin/env python3
"""The action post-pass (k_actions) on the device: token-dense and literal-heavy register-action programs, input resident in
HBM, wall time of kx_run_device; output compared with the CPU oracle.  usage: python profiles/actions_bench.py [LIB.so]"""
import json
import os
import random
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kleenexlang_amd import host  # noqa: E402
from oracle import oracle  # noqa: E402

if len(sys.argv) > 1:   # A/B: another build of the engine
    alt = os.path.abspath(sys.argv[1])
    host._lib_path = lambda name, _p=host._lib_path: alt if name == "libkxhip.so" else _p(name)

PROGRAMS = {
    # every line: two redirects, two writes (a token every few bytes)
    "swap_fields": ('main := (a@/[a-z]*/ ~/,/ b@/[0-9]*/ !b "," !a /\\n/)*\n', lambda r: b"%s,%d\n" % (bytes(r.choice(b"abcdefgh") for _ in range(r.randrange(1, 9))), r.randrange(10 ** 6))),
    # long literal runs between the tokens
    "long_lines": ('main := (l@/[^\\n]*/ ~/\\n/ "<" !l ">\\n")*\n', lambda r: bytes(r.choice(b"abcdefghijklmnopqrstuvwxyz ") for _ in range(r.randrange(150, 400))) + b"\n"),
}
res = {}
for name, (src, line) in PROGRAMS.items():
    rnd = random.Random(5)
    base = b"".join(line(rnd) for _ in range(20000))
    data = base * max(1, (int(os.environ.get("KX_BENCH_MIB", "16")) << 20) // len(base))
    blob = host.compile_source(src)
    want = oracle.run(blob, data)
    prog = host.Program(blob, collect_timing=True)
    t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    out = torch.empty(len(want) + (1 << 20), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    olen = prog.run_device(t.data_ptr(), t.numel(), out.data_ptr(), out.numel(), stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    olen = prog.run_device(t.data_ptr(), t.numel(), out.data_ptr(), out.numel(), stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ok = olen == len(want) and bytes(out[:olen].cpu().numpy().tobytes()) == want
    km = prog.last_stats.as_dict()
    res[name] = {"transducer_kernels_ms": round(sum(km["kernel_ms"].values()), 2), "unsynced_segments": km.get("unsynced_segments"), "input_bytes": len(data), "output_bytes": olen, "seconds": round(dt, 4), "input_MBps": round(len(data) / dt / 1e6, 1), "bit_exact": ok}
    prog.close()
print(json.dumps(res))
