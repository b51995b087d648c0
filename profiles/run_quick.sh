#!/bin/bash
# quick GPU check: engine parity tests (fast subset unless FULL=1) + bench lines; usage: run_quick.sh TAG [programs...]
ulimit -c 0; export HSA_COREDUMP_PATTERN=/dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-q}; shift
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
if [ -n "$FULL" ]; then
  timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt
else
  timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -x -q -k "not 10gib and not rccl and not binary" > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt
fi
for p in ${@:-apache_log}; do
  KX_DEBUG=${KXDBG:-} timeout 600 python bench.py --program $p --steps 10 --warmup 2 --no-cpu > $O/bench_$p.json 2> $O/bench_$p.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$p.json").read()); print("$p", d["value"], d["ms_per_step"], d["kernels_ms"], d["output_checked_bit_exact"])
except Exception as e: print("$p", "FAILED", e); print(open("$O/bench_$p.err").read()[-1500:])
PY
done
tail -4 $O/pytest.txt
