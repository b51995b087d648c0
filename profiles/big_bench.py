#!/usr/bin/env python3
"""BIG programs (table image beyond 16-bit LDS addressing: read from global memory through the L2), timed with the
input resident in HBM: (1) a generated dictionary rewriter of make_danish's size, (2) apache_log forced through the same
instances (KX_FORCE_BIG=1) next to its LDS-resident figure.  Every output byte is compared on the device with the oracle's
output of one base chunk, tiled.  usage: python profiles/big_bench.py [GIB]"""
import json
import os
import random
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import dictionary_program  # noqa: E402
from kleenexlang_amd import host  # noqa: E402
from oracle import oracle  # noqa: E402

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
src, words = dictionary_program()
blob = host.compile_source(src, opt=0)
rnd = random.Random(11)
alpha = "abcdefghijklmnopqrstuvwxyz"
base = " ".join(rnd.choice(words) if rnd.random() < 0.3 else "".join(rnd.choice(alpha) for _ in range(rnd.randint(2, 12))) for _ in range(400000)).encode() + b"\n"
want = oracle.run(blob, base)
k = max(1, int(gib * (1 << 30)) // len(base))
prog = host.Program(blob, collect_timing=True)
tb = torch.frombuffer(bytearray(base), dtype=torch.uint8).cuda()
t = tb.repeat(k)
out = torch.empty(len(want) * k + (1 << 20), dtype=torch.uint8, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
olen = prog.run_device(t.data_ptr(), t.numel(), out.data_ptr(), out.numel(), stream)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    olen = prog.run_device(t.data_ptr(), t.numel(), out.data_ptr(), out.numel(), stream)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
exp = torch.frombuffer(bytearray(want), dtype=torch.uint8).cuda()
ok = olen == len(want) * k and bool(torch.equal(out[:olen].view(k, len(want)), exp.expand(k, len(want))))
res = {"dictionary_rewriter": {"input_bytes": t.numel(), "output_bytes": olen, "ms": round(dt * 1e3, 2), "input_GBps": round(t.numel() / dt / 1e9, 1),
                               "kernels_ms": {k2: round(v, 3) for k2, v in prog.last_stats.as_dict()["kernel_ms"].items()}, "bit_exact": ok}}
prog.close()
del t, out
torch.cuda.empty_cache()
for name, env in (("apache_log_in_lds", {}), ("apache_log_forced_big", {"KX_FORCE_BIG": "1"})):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gib", str(gib), "--steps", "3", "--warmup", "1", "--no-cpu"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env))
    d = json.loads([x for x in r.stdout.decode().splitlines() if x.startswith('{"metric"')][-1])
    res[name] = {"input_GBps": d["value"], "ms_per_step": d["ms_per_step"], "kernels_ms": d["kernels_ms"], "bit_exact": d["output_checked_bit_exact"]}
print(json.dumps(res))
