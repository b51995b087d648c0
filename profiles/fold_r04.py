#!/usr/bin/env python3
"""Folds gpurun_out/<TAG>/ (written by profiles/run_final_r04.sh on the GPU box) into the tracked profiles/<TAG>_* files and a summary."""
import csv, json, os, re, shutil, sys
TAG = sys.argv[1] if len(sys.argv) > 1 else "r04z"
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(R, "gpurun_out", TAG), os.path.join(R, "profiles")
for f in ("gpu_pytest.txt", "kernel_stats.csv", "pmc_raw.json", "traffic.json", "soak_engine.txt", "soak_windows.txt"):
    if os.path.exists(os.path.join(G, f)): shutil.copy(os.path.join(G, f), os.path.join(P, "%s_%s" % (TAG, f)))
for p in ("apache_log", "csv2json", "iso_datetime_to_json"):
    shutil.copy(os.path.join(G, "bench_%s.json" % p), os.path.join(P, "%s_bench_%s.json" % (TAG, p)))
shutil.copy(os.path.join(G, "bench.json"), os.path.join(P, "%s_bench_under_rocprof.json" % TAG))
for f, t in (("bench_thousand_sep.json", "%s_bench_thousand_sep.json" % TAG), ("coder_bench_csv_rows_4gib.json", "r04z_coder_bench_csv_rows_4gib.json"),
             ("actions_16m.json", "%s_actions_16m.json" % TAG), ("actions_1g.json", "%s_actions_1g.json" % TAG)):
    if os.path.exists(os.path.join(G, f)) and os.path.getsize(os.path.join(G, f)): shutil.copy(os.path.join(G, f), os.path.join(P, t))
sq = os.path.join(R, "gpurun_out", TAG + "_sq", "sq_counters.json")
if os.path.exists(sq): shutil.copy(sq, os.path.join(P, "%s_sq_counters.json" % TAG))
load = lambda n: json.loads(open(os.path.join(G, n)).read())
b = {p: load("bench_%s.json" % p) for p in ("apache_log", "csv2json", "iso_datetime_to_json")}
ur, tr = load("bench.json"), load("traffic.json")
rows = {}
for r in csv.DictReader(open(os.path.join(G, "kernel_stats.csv"))):
    m = re.search(r"::(k_\w+)", r["Name"])
    if m: rows[m.group(1)] = (int(r["Calls"]), float(r["AverageNs"]) / 1e6)
ev = {"k_emit": "emit", "k_backlen": "backlen", "k_forward": "forward", "k_sync": "sync"}
out = ["# Round 4 final (%s): rocprofv3 summaries of the default engine (apache_log, 10 GiB resident in HBM, 1×MI355X)\n\n" % TAG,
       "Collected by `profiles/run_final_r04.sh %s` → `profiles/collect.sh`: `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 3 --warmup 1`\n" % TAG,
       "(`%s_kernel_stats.csv`; the bench line of that same run is `%s_bench_under_rocprof.json`), then `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, each in its own run with\n" % (TAG, TAG),
       "`--kernel-trace` only (`%s_pmc_raw.json`; `%s_traffic.json` = per launch, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16-byte-per-lane loads,\n" % (TAG, TAG),
       "stamped with engine_sha %s = the hash of the engine sources this was measured with).\n\n" % tr["engine_sha"],
       "| kernel | calls | avg ms (rocprof) | HIP events in bench.py | HBM fetch (corrected) | HBM write |\n|---|---|---|---|---|---|\n"]
tot = 0
for k, (calls, ms) in sorted(rows.items(), key=lambda x: -x[1][1]):
    t = tr["per_launch"].get(k, {"fetch_bytes_corrected": 0, "write_bytes": 0})
    tot += t["fetch_bytes_corrected"] + t["write_bytes"]
    out.append("| %s | %d | %.3f | %s | %.2f GB | %.2f GB |\n" % (k, calls, ms, ("%.3f" % ur["kernels_ms"][ev[k]]) if k in ev else "(in resolve)", t["fetch_bytes_corrected"] / 1e9, t["write_bytes"] / 1e9))
a = b["apache_log"]
out.append("\nSummed HBM traffic per step: %.1f GB for %.2f GB of input and %.2f GB of output (algorithmic minimum of this three-pass design: 3 + r = 4.25 B per input byte = 45.6 GB).\n" % (tot / 1e9, tr["input_bytes"] / 1e9, tr["output_bytes"] / 1e9))
out.append("Under rocprof: %.1f GB/s input, %.3f ms per step; plain runs `%s_bench_apache_log.json` (%.0f GB/s, %.1f ms, cpu_baseline %.3f GB/s on %d core), `%s_bench_csv2json.json` (%.0f GB/s), `%s_bench_iso_datetime_to_json.json` (%.0f GB/s): every output byte verified on the device.\n"
           % (ur["value"], ur["ms_per_step"], TAG, a["value"], a["ms_per_step"], a["cpu_baseline"]["value"], a["cpu_baseline"]["cores"], TAG, b["csv2json"]["value"], TAG, b["iso_datetime_to_json"]["value"]))
out.append("`roofline.frac` (SURVEY §8d: 1 B × input bytes ÷ dominant kernel %s ÷ 8 TB/s) = %.3f; whole path %.0f GB/s = %.1f %% of 8 TB/s.\n" % (a["roofline"]["kernel"], a["roofline"]["frac"], a["value"], a["value"] / 80))
pt = open(os.path.join(G, "gpu_pytest.txt")).read().strip().splitlines()[-1]
se = open(os.path.join(G, "soak_engine.txt")).read().strip().splitlines()[-1] if os.path.exists(os.path.join(G, "soak_engine.txt")) else "-"
sw = open(os.path.join(G, "soak_windows.txt")).read().strip().splitlines()[-1] if os.path.exists(os.path.join(G, "soak_windows.txt")) else "-"
out.append("GPU test suite of the same tree: `%s_gpu_pytest.txt` (%s); soak of this engine build: `%s_soak_engine.txt` (%s), `%s_soak_windows.txt` (%s).\n" % (TAG, pt, TAG, se, TAG, sw))
out.append("SQ counters of this engine (2 GiB, one counter per pass): `%s_sq_counters.json`.\n" % TAG)
out.append("Round-4 experiments and their evidence: `r04_experiments.md` (run by run), `r04_experiments.json` (every A/B line: `fold_r04_experiments.py`), scripts `r04_ceiling.sh`, `ceiling.py`, `r04_ab*.sh`.\n")
for f in ("timeline.txt", "gpu_pytest_jl1.txt"):
    if os.path.exists(os.path.join(G, f)): shutil.copy(os.path.join(G, f), os.path.join(P, "%s_%s" % (TAG, f)))
if os.path.exists(os.path.join(G, "timeline.txt")): out.append("Timeline of a `k_emit` wave in this engine (`KX_DEBUG_FLAGS=64`, 2 GiB): " + open(os.path.join(G, "timeline.txt")).read().strip() + "\n")
jl = {p_: json.loads(open(os.path.join(G, "bench_jl1_%s.json" % p_)).read()) for p_ in ("apache_log", "csv2json", "iso_datetime_to_json") if os.path.exists(os.path.join(G, "bench_jl1_%s.json" % p_))}
for p_, d in jl.items(): shutil.copy(os.path.join(G, "bench_jl1_%s.json" % p_), os.path.join(P, "%s_bench_jl1_%s.json" % (TAG, p_)))
if jl: out.append("Job-stride layout, opt-in (`KX_JL=1`, same box): " + ", ".join("%s %.0f GB/s (k_emit %.2f ms)" % (p_, d["value"], d["kernels_ms"]["emit"]) for p_, d in jl.items()) + "; its GPU tests `%s_gpu_pytest_jl1.txt`.\n" % TAG)
ex = lambda f: json.loads(open(os.path.join(G, f)).read()) if os.path.exists(os.path.join(G, f)) and os.path.getsize(os.path.join(G, f)) else None
ts, cb, a1 = ex("bench_thousand_sep.json"), ex("coder_bench_csv_rows_4gib.json"), ex("actions_1g.json")
if ts: out.append("Inline-constant layout: thousand_sep 10 GiB %.0f GB/s (`%s_bench_thousand_sep.json`, k_emit %.2f ms).\n" % (ts["value"], TAG, ts["kernels_ms"]["emit"]))
if cb: out.append("Table atoms + inline constants: CSV-row coder over 4 GiB %.0f GB/s, k_emit %.2f ms, %d output bytes checked (`r04z_coder_bench_csv_rows_4gib.json`).\n" % (cb["input_GBps"], cb["kernels_ms"]["emit"], cb["output_bytes_checked"]))
if a1: out.append("Action post-pass at 1 GiB: swap_fields %.1f GB/s, long_lines %.1f GB/s (`%s_actions_1g.json`).\n" % (a1["swap_fields"]["input_MBps"] / 1e3, a1["long_lines"]["input_MBps"] / 1e3, TAG))

open(os.path.join(P, "%s_summary.md" % TAG), "w").write("".join(out))
print("".join(out))
