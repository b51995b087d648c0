#!/bin/bash
# L2 / memory-side counters of the delayed-form kernels (one counter per pass, --kernel-trace only), 2 GiB, for two builds of the
# forward pass's record flush: cooperative (default) and per lane (KX_DEBUG_FLAGS=256) — VERDICT r5 item 2 ("counters, not adjectives").
# usage (GPU box): profiles/collect_tcc.sh TAG  → gpurun_out/TAG/tcc_counters.json
TAG=${1:-tcc}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export GRAFT_REPO_ROOT=$R
cd /tmp && export TMPDIR=/tmp
rm -f /tmp/tcc_*.csv
for mode in coop lane; do
  for c in ${TCCC:-TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum}; do
    rm -rf /tmp/tcc_${mode}_$c
    KX_DEBUG_FLAGS=$([ $mode = lane ] && echo 256 || echo 0) timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/tcc_${mode}_$c -o p -- python $R/bench.py --program ${PROG:-apache_log} --steps 2 --warmup 1 --no-cpu --gib 2 > /tmp/tcc_${mode}_$c.log 2>&1
    f=$(find /tmp/tcc_${mode}_$c -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && cp $f /tmp/tcc_${mode}_$c.csv
  done
done
python3 - "$OUT" <<'PY'
import csv, json, re, sys, collections, glob, os
out = sys.argv[1]
res = {"coop": collections.defaultdict(dict), "lane": collections.defaultdict(dict)}
for f in glob.glob("/tmp/tcc_*_*.csv"):
    m = re.match(r"tcc_(coop|lane)_(.*)\.csv", os.path.basename(f))
    mode, c = m.group(1), m.group(2)
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = re.search(r"k_\w+", r["Kernel_Name"])
        if k and r["Counter_Name"] == c:
            acc[k.group(0)][0] += 1; acc[k.group(0)][1] += float(r["Counter_Value"])
    for k, (n, v) in acc.items():
        res[mode][k][c] = v / n
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
try:
    from kleenexlang_amd import build as kbuild
    sha = kbuild.engine_sha()
except Exception:
    sha = None
json.dump({"workload": "apache_log 2 GiB, per launch (device totals)", "engine_sha": sha, "record_flush": res}, open(out + "/tcc_counters.json", "w"), indent=1)
for mode in ("coop", "lane"):
    d = res[mode].get("k_dforward", {})
    print(mode, "k_dforward", {k: int(v) for k, v in sorted(d.items())})
PY
