#!/bin/bash
# timing ablations of k_emit (KX_DEBUG_FLAGS: 1 = no constant copies, 2 = no flush; outputs wrong by construction)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-abl_emit}; mkdir -p $O; cd $R
for p in ${@:2}; do for f in 0 1 2 3; do
  KX_DEBUG=1 KX_DEBUG_FLAGS=$f timeout 300 python bench.py --program $p --steps 3 --warmup 1 --no-cpu > $O/${p}_f$f.json 2> $O/${p}_f$f.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/${p}_f$f.json").read()); print("$p flags", $f, d["kernels_ms"], d["output_checked_bit_exact"])
except Exception as e: print("$p flags", $f, "FAILED", e)
PY
  grep "\[kx\] emit" $O/${p}_f$f.err | tail -1
done; done
