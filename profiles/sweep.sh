#!/bin/bash
# usage: sweep.sh TAG PROGRAM "ENV=.. ENV=.." "ENV=.." ...   → one bench line per environment setting
ulimit -c 0; export HSA_COREDUMP_PATTERN=/dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; P=$2; shift; shift
O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
i=0
for e in "$@"; do
  i=$((i+1))
  env $e timeout 300 python bench.py --program $P --steps 6 --warmup 2 --no-cpu > $O/s$i.json 2> $O/s$i.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/s$i.json").read()); print("[$e]", d["value"], d["kernels_ms"], d["output_checked_bit_exact"])
except Exception as ex: print("[$e] FAILED", ex); print(open("$O/s$i.err").read()[-800:])
PY
done
