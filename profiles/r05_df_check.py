"""Round 5: first check of the delayed form on the GPU — the four workloads at a few sizes against the oracle, delayed form on
(default) and off (KX_DF=0 in a child process), then the per-kernel times at 1 GiB.  python profiles/r05_df_check.py [quick]"""
import os, sys, json, subprocess, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def child(progname, nbytes, seed):
    import torch
    from kleenexlang_amd import Program, compile_file, workloads
    from oracle import oracle
    blob = compile_file(progname)
    data = workloads.generate(workloads.PROGRAM_INPUT[progname], nbytes, seed=seed)
    want = oracle.run(blob, data)
    prog = Program(blob, collect_timing=True)
    t = torch.frombuffer(bytearray(data), dtype=torch.uint8).to("cuda:0")
    got = prog.run_tensor(t)
    torch.cuda.synchronize()
    got = bytes(got.cpu().numpy().tobytes())
    ok = got == want
    first = next((i for i in range(min(len(got), len(want))) if got[i] != want[i]), min(len(got), len(want)))
    print(json.dumps({"program": progname, "n": len(data), "df": os.environ.get("KX_DF", "1"), "ok": ok, "out": len(got), "want": len(want),
                      "first_diff": None if ok else first, "kernel_ms": getattr(prog, "last_stats", None) and prog.last_stats.as_dict()["kernel_ms"]}))
    if not ok:
        print("   got :", got[max(0, first - 60):first + 60])
        print("   want:", want[max(0, first - 60):first + 60])

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4])); sys.exit(0)
    sizes = [1000, 70000, 1 << 20, (1 << 23) + 12345]
    for prog in ("apache_log", "csv2json", "iso_datetime_to_json", "thousand_sep"):
        for n in sizes:
            for df in ("1", "0"):
                env = dict(os.environ, KX_DF=df, KX_DEBUG="1" if n == sizes[-1] else "")
                if not env["KX_DEBUG"]: env.pop("KX_DEBUG")
                r = subprocess.run([sys.executable, __file__, "child", prog, str(n), "3"], env=env, capture_output=True, text=True, timeout=600)
                sys.stdout.write(r.stdout); sys.stdout.write("".join(l for l in r.stderr.splitlines(True) if "[kx]" in l or "rror" in l)); sys.stdout.flush()
