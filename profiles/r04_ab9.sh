#!/bin/bash
ulimit -c 0; export HSA_COREDUMP_PATTERN=/dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04l; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_engine_gpu.py tests/test_pipeline_seam.py tests/test_regex_coder.py -m gpu -x -q -k "not 10gib and not rccl" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
run() { local name=$1; shift
  env "$@" timeout 600 python bench.py --program $P --steps 10 --warmup 2 --no-cpu > $O/bench_${P}_$name.json 2> $O/bench_${P}_$name.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_${P}_$name.json").read()); print("$P $name", d["value"], d["ms_per_step"], d["kernels_ms"], d["output_checked_bit_exact"])
except Exception as e: print("$P $name", "FAILED", e); print(open("$O/bench_${P}_$name.err").read()[-800:])
PY
}
P=csv2json
run c32 X=1; run nest KX_NO_CMPX32=1; run c32b X=1; run nestb KX_NO_CMPX32=1; run c32c X=1; run nestc KX_NO_CMPX32=1
