#!/bin/bash
# round 4, experiment 1: where does a k_emit wave spend its time, and what would conflict-free table reads / stores give?
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r04a}; mkdir -p $O; cd $R
ulimit -c 0
for kind in normal p192 p256 p128; do
  timeout 300 python profiles/ceiling.py --kind $kind --gib 2 > $O/plain_$kind.json 2> $O/plain_$kind.err; tail -1 $O/plain_$kind.json
done
for kind in normal p192 p256; do
  KX_DEBUG=1 KX_DEBUG_FLAGS=64 timeout 300 python profiles/ceiling.py --kind $kind --gib 2 > $O/tl_$kind.json 2> $O/tl_$kind.err; tail -1 $O/tl_$kind.json; grep "emit timeline" $O/tl_$kind.err | tail -1
done
for f in 32 96; do
  KX_DEBUG=1 KX_DEBUG_FLAGS=$f timeout 300 python profiles/ceiling.py --kind normal --gib 2 > $O/pitch_$f.json 2> $O/pitch_$f.err; tail -1 $O/pitch_$f.json; grep "emit timeline" $O/pitch_$f.err | tail -1
done
cd /tmp && export TMPDIR=/tmp
for kind in normal p256; do for c in SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU; do
  rm -rf /tmp/pmc_$c
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python $R/profiles/ceiling.py --kind $kind --gib 2 --steps 1 > /tmp/pmc_$c.log 2>&1
  python3 - $c $kind $(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1) <<'PY' | tee -a $O/sq_$kind.txt
import csv, re, sys, collections
c, kind, f = sys.argv[1:4]
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    m = re.search(r"k_\w+", r["Kernel_Name"])
    if m and r["Counter_Name"] == c:
        acc[m.group(0)][0] += 1; acc[m.group(0)][1] += float(r["Counter_Value"])
for k, (n, v) in sorted(acc.items()):
    if k in ("k_emit", "k_backlen", "k_forward"): print(kind, c, k, v / n)
PY
done; done
