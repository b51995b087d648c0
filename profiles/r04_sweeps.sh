#!/bin/bash
# round 4, final engine: segment sizes and waves per CU once more (the automatic choices were made for the round-2 engine)
ulimit -c 0; export HSA_COREDUMP_PATTERN=/dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04s; mkdir -p $O; cd $R
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read()); print(sys.argv[2], d["value"], d["ms_per_step"], d["kernels_ms"], d["output_checked_bit_exact"])
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
}
for P in apache_log csv2json iso_datetime_to_json; do
  for seg in 0 24576 32768 49152 65536; do
    timeout 600 python bench.py --program $P --steps 5 --warmup 1 --no-cpu --segment $seg > $O/seg_${P}_$seg.json 2>/dev/null; line $O/seg_${P}_$seg.json "$P seg=$seg"
  done
  for w in 8 12; do
    KX_EMIT_WAVES=$w timeout 600 python bench.py --program $P --steps 5 --warmup 1 --no-cpu > $O/waves_${P}_$w.json 2>/dev/null; line $O/waves_${P}_$w.json "$P waves=$w"
  done
done
# SQ counters of k_emit for the two constant-heavy configurations (final engine)
cd /tmp && export TMPDIR=/tmp
for P in csv2json iso_datetime_to_json; do for c in SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU; do
  rm -rf /tmp/pmc_$c
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python $R/bench.py --program $P --steps 1 --warmup 1 --no-cpu --gib 2 > /tmp/pmc_$c.log 2>&1
  python3 - $c $P $(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1) <<'PY' | tee -a $O/sq_$P.txt
import csv, re, sys, collections
c, kind, f = sys.argv[1:4]
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    m = re.search(r"k_\w+", r["Kernel_Name"])
    if m and r["Counter_Name"] == c:
        acc[m.group(0)][0] += 1; acc[m.group(0)][1] += float(r["Counter_Value"])
for k, (n, v) in sorted(acc.items()):
    if k in ("k_emit", "k_backlen", "k_forward"): print(kind, c, k, v / n, "per 4KiB", v / n / (2 * 2**30 / 4096))
PY
done; done
for P in csv2json iso_datetime_to_json; do
  KX_DEBUG=1 KX_DEBUG_FLAGS=64 timeout 300 python $R/bench.py --program $P --steps 2 --warmup 1 --no-cpu --gib 2 > $O/tl_$P.json 2> $O/tl_$P.err; grep "emit timeline" $O/tl_$P.err | tail -1 | sed "s/^/$P /" | tee -a $O/timelines.txt
done
