#!/usr/bin/env python3
"""Folds the A/B runs of round 4 (gpurun_out/r04a … r04i, r04s, r04ab, r04ab2, written by profiles/r04_*.sh on GPU boxes) into one tracked table:
profiles/r04_experiments.json (every run: box tag, variant, GB/s, per-kernel ms, bit-exact flag, k_emit plan line, timeline line)."""
import glob, json, os, re
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = []
for d in sorted(x for x in glob.glob(os.path.join(R, "gpurun_out", "r04*")) if not os.path.basename(x).startswith("r04z")):
    tag = os.path.basename(d)
    for f in sorted(glob.glob(os.path.join(d, "*.json"))):
        name = os.path.basename(f)[:-5]
        try:
            j = json.loads(open(f).read().strip().splitlines()[-1])
        except Exception:
            continue
        err = f[:-5] + ".err"
        plan = tl = None
        if os.path.exists(err):
            txt = open(err, errors="replace").read()
            m = re.findall(r"\[kx\] emit: (.*)", txt); plan = m[-1] if m else None
            m = re.findall(r"\[kx\] emit timeline \(share of wave time\): (.*)", txt); tl = m[-1] if m else None
        if not isinstance(j, dict):
            continue
        if "kernels_ms" in j and "value" in j:      # a bench.py line
            rows.append({"run": tag, "variant": name, "input_GBps": j["value"], "ms_per_step": j["ms_per_step"], "kernels_ms": j["kernels_ms"],
                         "bit_exact": j["output_checked_bit_exact"], "emit_plan": plan, "timeline": tl})
        elif "kernels_ms" in j:                     # a profiles/ceiling.py line (2 GiB)
            rows.append({"run": tag, "variant": name, "input_bytes": j["input_bytes"], "kernels_ms": j["kernels_ms"], "bit_exact": j["bit_exact"],
                         "emit_plan": plan, "timeline": tl})
    for f in sorted(glob.glob(os.path.join(d, "timelines.txt"))):
        rows.append({"run": tag, "variant": "timelines", "lines": open(f).read().splitlines()})
    for f in sorted(glob.glob(os.path.join(d, "sq_*.txt"))):
        rows.append({"run": tag, "variant": os.path.basename(f)[:-4], "sq_counters": [l.split() for l in open(f).read().splitlines() if l.strip()]})
    for f in sorted(glob.glob(os.path.join(d, "pytest*.txt"))):
        t = open(f).read().strip().splitlines()
        rows.append({"run": tag, "variant": os.path.basename(f)[:-4], "pytest_tail": t[-2:]})
json.dump({"_comment": "Round 4 A/B runs; one GPU box per `run` (variants inside a run were alternated on the same box). See profiles/r04_experiments.md.",
           "rows": rows}, open(os.path.join(R, "profiles", "r04_experiments.json"), "w"), indent=1)
print(len(rows), "rows")
