#!/bin/bash
# timing ablations of k_backlen / k_place (KX_DEBUG_FLAGS bits; outputs are wrong by construction, only kernel times matter)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-abl}; mkdir -p $O; cd $R
for f in 0 2 4 8 14 16 32 48; do
  KX_DEBUG_FLAGS=$f timeout 300 python bench.py --program ${2:-apache_log} --steps 5 --warmup 1 --no-cpu > $O/f$f.json 2> $O/f$f.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/f$f.json").read()); print("flags", $f, d["kernels_ms"], d["output_checked_bit_exact"])
except Exception as e: print("flags", $f, "FAILED", e)
PY
done
KX_EMIT_OLD=1 timeout 300 python bench.py --program ${2:-apache_log} --steps 5 --warmup 1 --no-cpu > $O/old.json 2> $O/old.err
python -c "
import json; d=json.loads(open('$O/old.json').read()); print('old', d['kernels_ms'], d['output_checked_bit_exact'])"
