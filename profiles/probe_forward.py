"""Timing probe (experiments, not the bench): one apache_log shard resident in HBM; prints kernel_ms of a run, or the
wall time of a run that ends in a match error.  PERIODIC=1 makes every 32 KiB segment hold the same bytes (all lanes of a wave
then read the same table rows: LDS reads without bank conflicts)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kleenexlang_amd import Program, MatchError, compile_file, workloads
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
base = workloads.generate("apache_log", 64 << 20, 7)
seg = 0
if os.environ.get("PERIODIC"):     # every 32 KiB segment holds the same whole lines (the last one padded in its user-agent field)
    b = workloads.generate("apache_log", 32768 - 400, 7)
    last = b[:-1].rfind(b"\n") + 1
    line = b[last:]
    q = line.rfind(b'"')
    b = b[:last] + line[:q] + b"x" * (32768 - len(b)) + line[q:]
    assert len(b) == 32768
    base, seg = b, 32768
tb = torch.frombuffer(bytearray(base), dtype=torch.uint8).to(dev)
n = int(gib * (1 << 30)) // len(base) * len(base)
t = tb.repeat(n // len(base)).clone()
out = torch.empty(int(n * 1.3) + (1 << 20), dtype=torch.uint8, device=dev)
p = Program(compile_file("apache_log"), collect_timing=True, segment_bytes=seg)
st = torch.cuda.current_stream(dev).cuda_stream
for it in range(4):
    torch.cuda.synchronize(); t0 = time.time()
    try:
        p.run_device(t.data_ptr(), n, out.data_ptr(), out.numel(), st)
        torch.cuda.synchronize()
        print("ok  %.3f ms" % ((time.time() - t0) * 1e3), {k: round(v, 3) for k, v in p.last_stats.as_dict()["kernel_ms"].items() if v})
    except MatchError as e:
        torch.cuda.synchronize()
        print("fail@%d %.3f ms wall" % (e.pos, (time.time() - t0) * 1e3), {k: round(v, 3) for k, v in p.last_stats.as_dict()["kernel_ms"].items() if v})
