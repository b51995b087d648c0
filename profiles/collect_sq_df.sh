#!/bin/bash
# SQ counters of the delayed-form kernels (one counter per pass, --kernel-trace only), 2 GiB.
# usage (on the GPU box): profiles/collect_sq_df.sh TAG   → gpurun_out/TAG/sq_counters.json
TAG=${1:-sq}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export GRAFT_REPO_ROOT=$R
cd /tmp && export TMPDIR=/tmp
rm -f /tmp/pmc_*.csv
for c in ${SQC:-GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR}; do
  rm -rf /tmp/pmc_$c
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python $R/bench.py --program ${PROG:-apache_log} --steps 2 --warmup 1 --no-cpu --gib 2 > /tmp/pmc_$c.log 2>&1
  cp $(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1) /tmp/pmc_$c.csv
done
python3 - "$OUT" <<'PY'
import csv, json, re, sys, collections, glob, os
out = sys.argv[1]
res = collections.defaultdict(dict)
for f in glob.glob("/tmp/pmc_*.csv"):
    c = os.path.basename(f)[4:-4]
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        m = re.search(r"k_\w+", r["Kernel_Name"])
        if m and r["Counter_Name"] == c:
            acc[m.group(0)][0] += 1; acc[m.group(0)][1] += float(r["Counter_Value"])
    for k, (n, v) in acc.items():
        res[k][c] = v / n
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
try:
    from kleenexlang_amd import build as kbuild
    sha = kbuild.engine_sha()
except Exception:
    sha = None
json.dump({"workload": "%s 2 GiB, per launch (device totals)" % os.environ.get("PROG", "apache_log"), "program": os.environ.get("PROG", "apache_log"),
           "input_bytes": 2 * 2**30 // (32 << 20) * (32 << 20), "engine_sha": sha, "kernels": res}, open(out + "/sq_counters.json", "w"), indent=1)
for k in [x for x in ("k_demit", "k_dforward", "k_emit", "k_backlen", "k_forward") if x in res]:
    d = res[k]
    cu = d["GRBM_GUI_ACTIVE"] / 8
    print(k, "CU-cycles per 4 KiB = %.0f" % (cu * 256 / (2 * 2**30 / 4096)), "LDS active / CU-cycles = %.2f" % (d["SQ_LDS_IDX_ACTIVE"] / 256 / cu),
          "conflict share = %.2f" % (d["SQ_LDS_BANK_CONFLICT"] / d["SQ_LDS_IDX_ACTIVE"]),
          "VALU/LDS/SALU instr per 4 KiB = %.0f / %.0f / %.0f" % tuple(d[x] / (2 * 2**30 / 4096) for x in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU")),
          "busy(any)/wave-cycles = %.2f" % (d["SQ_ACTIVE_INST_ANY"] / d["SQ_WAVE_CYCLES"]))
PY
