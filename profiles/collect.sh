#!/bin/bash
# Runs on the GPU box (gpurun): rocprofv3 kernel-trace stats of the default bench, then the PMC passes for HBM
# traffic (each counter in its own run with --kernel-trace only), and folds them into profiles-style files
# under gpurun_out/$TAG.  usage: profiles/collect.sh TAG
TAG=${1:-prof}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o p -- python $R/bench.py --steps 3 --warmup 1 > $OUT/bench_under_rocprof.log 2>&1
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
grep "^{\"metric\"" $OUT/bench_under_rocprof.log | tail -1 > $OUT/bench.json
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu > $OUT/pmc_$c.log 2>&1
  cp $(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1) /tmp/pmc_$c.csv
done
python3 - "$OUT" <<'PY'
import csv, json, os, re, sys, collections
out = sys.argv[1]
raw = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open("/tmp/pmc_%s.csv" % c)):
        m = re.search(r"k_\w+", r["Kernel_Name"])
        if m and r["Counter_Name"] == c:
            acc[m.group(0)][0] += 1; acc[m.group(0)][1] += float(r["Counter_Value"])
    raw[c] = {k: {"launches": n, "sum_KB": v} for k, (n, v) in acc.items()}
json.dump(raw, open(out + "/pmc_raw.json", "w"), indent=1)
bench = json.loads(open(out + "/bench.json").read())
per = {}
for k in raw["FETCH_SIZE"]:
    f = raw["FETCH_SIZE"][k]; w = raw["WRITE_SIZE"].get(k, {"launches": 1, "sum_KB": 0})
    fb = int(2 * 1024 * f["sum_KB"] / f["launches"]); wb = int(1024 * w["sum_KB"] / max(1, w["launches"]))
    per[k] = {"fetch_bytes_corrected": fb, "write_bytes": wb, "total": fb + wb}
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from kleenexlang_amd import build as kbuild
json.dump({"engine_sha": kbuild.engine_sha(), "program": "apache_log", "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only), python bench.py --steps 1 "
           "--warmup 1 --no-cpu, apache_log 10 GiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (16-byte-per-lane loads report half); KB->bytes",
           "input_bytes": bench["config"]["input_bytes_per_gpu"], "output_bytes": bench["config"]["output_bytes_rank0"],
           "per_launch": per}, open(out + "/traffic.json", "w"), indent=1)
print(json.dumps(per, indent=1)[:1500])
PY
