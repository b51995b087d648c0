#!/usr/bin/env python3
"""Ceiling experiments for the table-read bank conflicts (round 4): the same engine on inputs whose 64-byte pieces are
all alike, so that the 64 lanes of a wave read the SAME table words at every step (LDS broadcast, no conflicts).

  python profiles/ceiling.py [--gib G] [--kind normal|p128|p192|p64x] [--steps K]

kind p128: every log line is exactly 128 bytes (two alternating piece contents), p192: 192 bytes (three), normal: the
synthetic log of bench.py.  Output is compared with the oracle on one base chunk (tiled), so the numbers are those of
a correct run.  Prints one JSON line with the per-kernel milliseconds.
"""
import argparse, json, os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def fixed_line(nbytes, r):
    head = '%d.%d.%d.%d - - [%02d/%s/%d:%02d:%02d:%02d +0100] "GET /' % (r.randint(100, 255), r.randint(100, 255), r.randint(100, 255), r.randint(100, 255),
                                                                    r.randint(10, 28), "Oct", 2014, r.randint(10, 23), r.randint(10, 59), r.randint(10, 59))
    tail = ' HTTP/1.1" 200 %d "-" "curl/7.35.0"\n' % r.randint(10000, 99999)
    fill = nbytes - len(head) - len(tail)
    assert fill > 0
    return head + "".join(r.choice("abcdefghijklmnopqrstuvwxyz") for _ in range(fill)) + tail


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gib", type=float, default=2.0)
    ap.add_argument("--kind", default="p128")
    ap.add_argument("--steps", type=int, default=3)
    a = ap.parse_args()
    import torch
    from kleenexlang_amd import Program, compile_file, workloads
    from oracle import oracle
    dev = torch.device("cuda", 0)
    blob = compile_file("apache_log")
    prog = Program(blob, collect_timing=True)
    if a.kind == "normal":
        base = workloads.generate("apache_log", 32 << 20)
    else:
        per = int(a.kind[1:])
        r = random.Random(7)
        one = fixed_line(per, r)
        base = (one * ((4 << 20) // per)).encode()      # the SAME line over and over: every lane of a wave walks the same states
    n = int(a.gib * (1 << 30))
    k = max(1, n // len(base))
    tb = torch.frombuffer(bytearray(base), dtype=torch.uint8).to(dev)
    t = tb.repeat(k)
    out = torch.empty(int(t.numel() * 1.45) + (1 << 20), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    kern = {}
    for i in range(a.steps + 1):
        olen = prog.run_device(t.data_ptr(), t.numel(), out.data_ptr(), out.numel(), stream)
        if i:
            for kn, ms in prog.last_stats.as_dict()["kernel_ms"].items():
                kern[kn] = kern.get(kn, 0.0) + ms / a.steps
    torch.cuda.synchronize()
    want = oracle.run(blob, base)
    ok = workloads.check_tiled_on_device(out[:olen], 0, workloads.tiled_parts("apache_log", want, k))
    gb = t.numel() / 1e9
    print(json.dumps({"kind": a.kind, "input_bytes": t.numel(), "out_over_in": olen / t.numel(), "bit_exact": bool(ok),
                      "kernels_ms": {k_: round(v, 4) for k_, v in kern.items()},
                      "ms_per_GB": {k_: round(v / gb, 4) for k_, v in kern.items()}}), flush=True)


if __name__ == "__main__":
    main()
