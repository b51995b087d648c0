#!/bin/bash
ulimit -c 0; export HSA_COREDUMP_PATTERN=/dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04j; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
run() { local name=$1; shift
  env "$@" KX_DEBUG=1 timeout 600 python bench.py --program $P --steps 10 --warmup 2 --no-cpu > $O/bench_${P}_$name.json 2> $O/bench_${P}_$name.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_${P}_$name.json").read()); print("$P $name", d["value"], d["ms_per_step"], d["kernels_ms"], d["output_checked_bit_exact"])
except Exception as e: print("$P $name", "FAILED", e); print(open("$O/bench_${P}_$name.err").read()[-800:])
PY
  grep -o "layout=[a-z-]* next=[a-z()-]*" $O/bench_${P}_$name.err | sort | uniq -c | tr '\n' ';'; echo
}
for P in apache_log csv2json iso_datetime_to_json thousand_sep; do
  run auto X=1; run fixed KX_JL_AUTO_OFF=1; run auto2 X=1; run fixed2 KX_JL_AUTO_OFF=1
done
